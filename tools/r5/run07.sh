#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1200 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "bn_ or apply or bnapply or head or abn" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_zf_unet_gpu.py -x -q -m gpu -k "reproducible or golden or bs32" 2>&1 | tail -4
python tools/coresidency_probe.py 2>/dev/null | tail -3
run() { python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-box "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
echo "lean apply:   $(run)"
echo "wide apply:   $(SEGNB_BN_LEAN=0 run)"
done
for m in linknet34 fcdensenet103; do
echo "$m lean: $(run --model $m)  wide: $(SEGNB_BN_LEAN=0 run --model $m)"
done

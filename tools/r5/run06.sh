#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "fused_bn_reduce" 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_replay_gpu.py tests/test_replay_guard_gpu.py -x -q -m gpu -k "fcdense or tiramisu or dense" 2>&1 | tail -6
run() { python bench.py --model fcdensenet103 --steps 60 --warmup 8 --no-box 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
echo "fused reduce in the dense data gradient: $(run)"
echo "separate reduce pass:                    $(SEGNB_BNREDUCE_FUSED=0 run)"
done

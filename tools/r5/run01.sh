#!/bin/bash
# split-K at 7x7: parity tests, per-layer A/B, step A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "split_k or tall_7x7 or (full_size and 32x7x)" 2>&1 | tail -15
for ks in 0 1; do
  echo "== layer_bench ksplit=$ks"
  python tools/layer_bench.py --only enc5.l1,enc5.l2 --what fprop,dgrad --ksplit $ks --reps 50
done
run() { python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-box 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
echo "ksplit auto: $(run)"
echo "ksplit off:  $(SEGNB_FPROP_KSPLIT=0 run)"
done

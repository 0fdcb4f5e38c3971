#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "split_k or tall_7x7 or (full_size and 32x7x)" 2>&1 | tail -5
for nm in 0 1; do for ks in 0 1; do
  echo "== layer_bench ntmajor=$nm ksplit=$ks"
  SEGNB_FPROP_NTMAJOR=$nm python tools/layer_bench.py --only enc5.l1,enc5.l2 --what fprop,dgrad --ksplit $ks --reps 50 2>/dev/null
done; done
run() { python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-box 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
echo "ntmajor+ksplit:   $(run)"
echo "ntmajor only:     $(SEGNB_FPROP_KSPLIT=0 run)"
echo "neither:          $(SEGNB_FPROP_KSPLIT=0 SEGNB_FPROP_NTMAJOR=0 run)"
done

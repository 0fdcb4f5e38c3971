#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5
python tools/prologue_gap.py 2>/dev/null | tee gpurun_out/r5/prologue_gap.txt
echo "== fuse-optimizer at ws=1"
run() { python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-box "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do
echo "plain:            $(run)"
echo "fuse-optimizer:   $(run --fuse-optimizer)"
done
echo "== sustained 2000 steps"
python bench.py --steps 2000 --warmup 10 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r5/bench_2000steps.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
echo "== cpu sweep"
python tools/cpu_sweep.py 2>/dev/null | tee gpurun_out/r5/cpu_sweep.txt | tail -3

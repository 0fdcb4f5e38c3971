#!/bin/bash
cd "$(dirname "$0")/../.."
run() { python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-box "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3 4; do
echo "default:            $(run)"
echo "consumer fusion:    $(SEGNB_CONSUMER_FUSION=1 run)"
done

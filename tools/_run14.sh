cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "upcat" 2>&1 | tail -8 > gpurun_out/r3n/upcat.log
python tools/upcat_bench.py > gpurun_out/r3n/upcat_bench.txt 2>&1
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r3n/bench_auto_$i.json 2>> gpurun_out/r3n/bench.err
SEGNB_SUBPIXEL=0 python bench.py --no-cpu-baseline > gpurun_out/r3n/bench_plain_$i.json 2>> gpurun_out/r3n/bench.err
done
tail -4 gpurun_out/r3n/upcat.log; cat gpurun_out/r3n/upcat_bench.txt | cut -c1-200
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3n/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "upcat" 2>&1 | tail -5 > gpurun_out/r3l/upcat.log
for i in 1 2 3; do
python bench.py --no-cpu-baseline > gpurun_out/r3l/bench_thin_$i.json 2>> gpurun_out/r3l/bench.err
SEGNB_SUBPIXEL_THIN=0 python bench.py --no-cpu-baseline > gpurun_out/r3l/bench_nothin_$i.json 2>> gpurun_out/r3l/bench.err
SEGNB_SUBPIXEL=force python bench.py --no-cpu-baseline > gpurun_out/r3l/bench_force_$i.json 2>> gpurun_out/r3l/bench.err
done
tail -3 gpurun_out/r3l/upcat.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3l/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
for i in 1 2; do
for np in 0 1 2 3; do
SEGNB_WGRAD_POSTPONE=$np python bench.py --no-cpu-baseline > gpurun_out/r3m/bench_post${np}_$i.json 2>> gpurun_out/r3m/bench.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY

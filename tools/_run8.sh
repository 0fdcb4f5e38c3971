cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "upcat" 2>&1 | tail -5 > gpurun_out/r3h/upcat.log
for i in 1 2 3; do
SEGNB_SUBPIXEL=0 python bench.py --no-cpu-baseline > gpurun_out/r3h/bench_plain_$i.json 2>> gpurun_out/r3h/bench.err
python bench.py --no-cpu-baseline > gpurun_out/r3h/bench_auto_$i.json 2>> gpurun_out/r3h/bench.err
SEGNB_SUBPIXEL=force python bench.py --no-cpu-baseline > gpurun_out/r3h/bench_force_$i.json 2>> gpurun_out/r3h/bench.err
done
tail -3 gpurun_out/r3h/upcat.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/bench_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
PY

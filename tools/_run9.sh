cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
python -m pytest tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3i/hip_ops.log
python -m pytest tests/test_zf_unet_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3i/models.log
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r3i/bench_auto_$i.json 2>> gpurun_out/r3i/bench.err
done
tail -3 gpurun_out/r3i/hip_ops.log; tail -3 gpurun_out/r3i/models.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/bench_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
PY

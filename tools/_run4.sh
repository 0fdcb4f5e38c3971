cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "upcat" 2>&1 | tail -30 > gpurun_out/r3d/upcat.log
python -m pytest tests/test_zf_unet_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r3d/zf_tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r3d/bench_subpixel_$i.json 2>> gpurun_out/r3d/bench.err
SEGNB_SUBPIXEL=0 python bench.py --no-cpu-baseline > gpurun_out/r3d/bench_plain_$i.json 2>> gpurun_out/r3d/bench.err
done
tail -4 gpurun_out/r3d/upcat.log; tail -4 gpurun_out/r3d/zf_tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/bench_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
PY

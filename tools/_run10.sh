cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "bn_apply or bnapply" 2>&1 | tail -12 > gpurun_out/r3j/bnapply.log
python -m pytest tests/test_zf_unet_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3j/zf.log
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r3j/bench_auto_$i.json 2>> gpurun_out/r3j/bench.err
SEGNB_WGRAD_BNAPPLY=0 python bench.py --no-cpu-baseline > gpurun_out/r3j/bench_nobna_$i.json 2>> gpurun_out/r3j/bench.err
done
tail -6 gpurun_out/r3j/bnapply.log; tail -3 gpurun_out/r3j/zf.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j/bench_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
PY

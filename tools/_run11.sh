cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "upcat" 2>&1 | tail -12 > gpurun_out/r3k/upcat.log
python tools/upcat_bench.py > gpurun_out/r3k/upcat_bench.txt 2>&1
python -m pytest tests/test_zf_unet_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3k/zf.log
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r3k/bench_auto_$i.json 2>> gpurun_out/r3k/bench.err
SEGNB_SUBPIXEL_FWD=0 python bench.py --no-cpu-baseline > gpurun_out/r3k/bench_nofwd_$i.json 2>> gpurun_out/r3k/bench.err
done
tail -6 gpurun_out/r3k/upcat.log; cat gpurun_out/r3k/upcat_bench.txt; tail -3 gpurun_out/r3k/zf.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3k/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 gpurun_out/r3k/bench.err

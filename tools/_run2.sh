cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "upcat" 2>&1 | tail -30 > gpurun_out/r3b/upcat.log
python -m pytest tests/test_zf_unet_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r3b/zf_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r3b/bench_subpixel.json 2> gpurun_out/r3b/bench.err
SEGNB_SUBPIXEL=0 python bench.py --no-cpu-baseline > gpurun_out/r3b/bench_plain.json 2>> gpurun_out/r3b/bench.err
tail -5 gpurun_out/r3b/upcat.log; tail -5 gpurun_out/r3b/zf_tests.log; cut -c1-400 gpurun_out/r3b/bench_subpixel.json; cut -c1-400 gpurun_out/r3b/bench_plain.json

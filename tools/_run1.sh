cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python -m pytest tests/test_zf_unet_gpu.py -x -q -m gpu -s 2>&1 | tail -60 > gpurun_out/r3a/zf_tests.log
python -m pytest tests/test_models_gpu.py tests/test_tiles_gpu.py -x -q -m gpu -k "abn or split or gather" 2>&1 | tail -15 > gpurun_out/r3a/other_tests.log
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "head" 2>&1 | tail -15 >> gpurun_out/r3a/other_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
python bench.py --no-cpu-baseline > gpurun_out/r3a/bench2.json 2>> gpurun_out/r3a/bench.err
tail -5 gpurun_out/r3a/zf_tests.log; cat gpurun_out/r3a/bench.json | cut -c1-600

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "upcat" 2>&1 | tail -30 > gpurun_out/r3c/upcat.log
python tools/upcat_bench.py > gpurun_out/r3c/upcat_bench.txt 2>&1
SEGNB_FPROP_UPD=0 python tools/upcat_bench.py > gpurun_out/r3c/upcat_bench_general.txt 2>&1
tail -4 gpurun_out/r3c/upcat.log; cat gpurun_out/r3c/upcat_bench.txt; cat gpurun_out/r3c/upcat_bench_general.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
python tools/host_ahead.py > gpurun_out/r3f/host_ahead.txt 2>&1
cat gpurun_out/r3f/host_ahead.txt

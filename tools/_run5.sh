cd $GRAFT_REPO_ROOT
R=$(pwd)
mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3e/trace -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/r3e/trace.log 2>&1
cd $R
python tools/trace_step.py gpurun_out/r3e/trace 3 list > gpurun_out/r3e/step.txt 2>&1
head -70 gpurun_out/r3e/step.txt
rm -rf gpurun_out/r3e/trace

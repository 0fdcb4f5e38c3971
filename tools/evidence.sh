#!/bin/bash
# The measurement set behind profiles/rNN_*: run on the GPU box from the repository root as
#     bash tools/evidence.sh r03
# Writes under gpurun_out/<tag>_*; the summaries that are judged are then copied into profiles/ by hand.
# rocprofv3: the python program directly after "--" (no env / bash -c hop), counters in passes of their own.
set -u
TAG=${1:-r03}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
python3 $R/bench.py --size 512 --batch 16 --no-cpu-baseline > $O/${TAG}_bench_512.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stats -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-box > $O/${TAG}_prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --timed-only > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --timed-only > $O/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/${TAG}_pmc_sq1 -- python3 $R/bench.py --steps 4 --warmup 2 --timed-only > $O/${TAG}_pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/${TAG}_pmc_sq2 -- python3 $R/bench.py --steps 4 --warmup 2 --timed-only > $O/${TAG}_pmc_sq2.log 2>&1
for m in linknet34 fcdensenet103 unet16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$m -- python3 $R/bench.py --model $m --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-box > $O/${TAG}_prof_$m.log 2>&1
  find $O/${TAG}_prof_$m -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/${TAG}_${m}_kernel_stats.csv
  rm -rf $O/${TAG}_prof_$m
done
# HBM traffic of the other model rows: the same two PMC passes per model
for m in linknet34 fcdensenet103 unet16; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmcf_$m -- python3 $R/bench.py --model $m --steps 4 --warmup 2 --timed-only > $O/${TAG}_pmcf_$m.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmcw_$m -- python3 $R/bench.py --model $m --steps 4 --warmup 2 --timed-only > $O/${TAG}_pmcw_$m.log 2>&1
  (cd $R && python3 tools/pmc_traffic.py $O/${TAG}_pmcf_$m $O/${TAG}_pmcw_$m $O/${TAG}_pmc_traffic_$m.json > $O/${TAG}_pmc_traffic_$m.log 2>&1)
  rm -rf $O/${TAG}_pmcf_$m $O/${TAG}_pmcw_$m
done
cd $R
python3 tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.log 2>&1
python3 tools/pmc_summary.py $O/${TAG}_pmc_sq1 conv_ > $O/${TAG}_pmc_conv_issue_wait.txt 2>&1
python3 tools/pmc_summary.py $O/${TAG}_pmc_sq2 conv_ > $O/${TAG}_pmc_conv_lds.txt 2>&1
find $O/${TAG}_prof_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/${TAG}_bench_bs32_kernel_stats.csv
python3 tools/layer_bench.py > $O/${TAG}_layers.txt 2>&1
python3 tools/prologue_gap.py > $O/${TAG}_prologue_gap.txt 2>&1
python3 tools/coresidency_probe.py > $O/${TAG}_coresidency.txt 2>&1
python3 bench.py --model unet16 --tiled > $O/${TAG}_bench_tiled.json 2> $O/${TAG}_bench_tiled.log
python3 tools/bn_bench.py > $O/${TAG}_bn_passes.txt 2>&1
python3 tools/step_timeline.py > $O/${TAG}_timeline.txt 2>&1
python3 tools/upcat_bench.py > $O/${TAG}_upcat_layers.txt 2>&1
python3 tools/c8_bench.py > $O/${TAG}_c8_bench.txt 2>&1
python3 tools/pack_bench.py > $O/${TAG}_pack_bench.txt 2>&1
python3 tools/head_bench.py > $O/${TAG}_head_bench.txt 2>&1
python3 tools/dense_dgrad_bench.py > $O/${TAG}_dense_dgrad_bench.txt 2>&1
python3 tools/step_ramp.py > $O/${TAG}_step_ramp.txt 2>&1
python3 tools/mask_bench.py > $O/${TAG}_mask_bench.txt 2>&1
python3 tools/determinism_probe.py > $O/${TAG}_determinism.txt 2>&1
# cross-stream fork: cost of the marker packet vs an event on the kernel's own dispatch, and the ordering check
mkdir -p tools/_bin
[ -x tools/_bin/fork_cost ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/_bin/fork_cost tools/fork_cost.hip > /dev/null 2>&1
[ -x tools/_bin/fork_check ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/_bin/fork_check tools/fork_check.hip > /dev/null 2>&1
(./tools/_bin/fork_cost; ./tools/_bin/fork_check) > $O/${TAG}_fork_cost.txt 2>&1
# (--timed-only: every step of the traced run is a warm-up or a TIMED step -- without it the steps counted from the end are the logged /
# host-probe steps that follow the timed region, which start from an idle GPU and carry the logging's reductions)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --steps 12 --warmup 3 --timed-only > $O/${TAG}_trace.log 2>&1)
python3 tools/trace_step.py $O/${TAG}_trace 4 list > $O/${TAG}_step_trace.txt 2>&1
rm -rf $O/${TAG}_trace
# one traced step of each executor model (kernel families, queue gaps)
for m in linknet34 fcdensenet103 unet16; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_$m -- python3 $R/bench.py --model $m --steps 8 --warmup 3 --timed-only > $O/${TAG}_trace_$m.log 2>&1)
  python3 tools/trace_step.py $O/${TAG}_trace_$m 3 > $O/${TAG}_step_trace_$m.txt 2>&1
  rm -rf $O/${TAG}_trace_$m
done
# the same step replayed from ONE HIP graph (where does the two-branch overlap go?)
python3 bench.py --graph on --no-cpu-baseline --no-kernel-timer > $O/${TAG}_bench_graph.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_graph -- python3 $R/bench.py --graph on --steps 12 --warmup 3 --timed-only > $O/${TAG}_trace_graph.log 2>&1)
python3 tools/trace_step.py $O/${TAG}_trace_graph 4 list > $O/${TAG}_step_trace_graph.txt 2>&1
rm -rf $O/${TAG}_trace_graph
python3 tools/insitu.py > $O/${TAG}_insitu.txt 2>&1
python3 tools/host_profile.py --plan-profile > $O/${TAG}_host_profile.txt 2>&1
python3 tools/model_bench.py --steps 10 > $O/${TAG}_model_bench.txt 2>&1
python3 tools/conv_sites.py --model linknet34 --top 30 > $O/${TAG}_conv_sites_linknet34.txt 2>&1
python3 tools/conv_sites.py --model fcdensenet103 --top 30 > $O/${TAG}_conv_sites_fcdensenet103.txt 2>&1
python3 tools/conv_sites.py --model unet16 --size 1024 --batch 4 --top 30 > $O/${TAG}_conv_sites_unet16.txt 2>&1
for m in linknet34 fcdensenet103 unet16; do python3 bench.py --model $m --no-cpu-baseline >> $O/${TAG}_bench_models.json 2>/dev/null; done
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 $O/${TAG}_prof_stats
echo evidence done

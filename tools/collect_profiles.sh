#!/bin/bash
# Collect the rocprofv3 evidence on the GPU box for one of bench.py's workloads:
#   1. --kernel-trace --stats            -> gpurun_out/<tag>_kernel_stats.csv
#   2. separate --pmc passes             -> gpurun_out/<tag>_pmc.json (per-kernel means per launch)
# Counters are collected in their own runs (never with sys/hip/hsa traces).  Only small summaries are written under
# gpurun_out/ (the raw CSVs stay in /tmp on the box).  Only the workload's own kernels run in the profiled process: no
# CPU baseline, no builds (python3 is started directly, never through env / bash -c).
# usage (on the box, from the repo root):  bash tools/collect_profiles.sh <tag> <track|frontend|legs> [bench args...]
#   track     the headline: svo_track_batch_dev over 1000 frames (no legs)
#   frontend  the batched front end, 128 pairs per launch
#   legs      a short headline run WITH its legs (multi_sequence, sharded, semantic_elas, elas, msa): kernel stats and the
#             HBM traffic counters (FETCH_SIZE / WRITE_SIZE passes) of the ELAS / MSA / gating kernels
set -u
TAG=${1:-r03}; shift || true
WL=${1:-track}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
FULL=1
if [ "$WL" = "track" ]; then
  ARGS="--workload track --frames 1000 --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-legs $*"
elif [ "$WL" = "legs" ]; then
  ARGS="--workload track --frames 600 --steps 2 --warmup 1 --no-cpu-baseline --no-profile $*"
  FULL=0
else
  ARGS="--workload frontend --frames 1536 --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-legs $*"
fi
# counters only for the library's own kernels (k_...): the renderer's hundreds of thousands of elementwise launches would
# each be serialised and read out otherwise (minutes per pass)
KRE='(^|[^a-zA-Z0-9_])k_[a-z]'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py $ARGS > /tmp/prof_stats.log 2>&1
python3 $R/tools/kstats.py /tmp/prof_stats $OUT/${TAG}_kernel_stats.csv
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
DIRS=""
if [ $FULL = 1 ]; then
  timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc $P1 --output-format csv -d /tmp/prof_pmc1 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc $P2 --output-format csv -d /tmp/prof_pmc2 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc2.log 2>&1
  DIRS="/tmp/prof_pmc1 /tmp/prof_pmc2"
fi
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc FETCH_SIZE --output-format csv -d /tmp/prof_pmc3 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc WRITE_SIZE --output-format csv -d /tmp/prof_pmc4 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc4.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc.json $DIRS /tmp/prof_pmc3 /tmp/prof_pmc4 > $OUT/${TAG}_pmc.txt
tail -12 $OUT/${TAG}_pmc.txt | cut -c1-300
tail -3 /tmp/prof_stats.log | cut -c1-300

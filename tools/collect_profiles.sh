#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's default command on the GPU box:
#   1. --kernel-trace --stats            -> profiles/<tag>_kernel_stats.csv
#   2. four separate --pmc passes        -> profiles/<tag>_pmc.json (per-kernel means per launch)
# Counters are collected in their own runs (never with sys/hip/hsa traces).  Only small summaries
# are written under gpurun_out/ (the raw CSVs stay in /tmp on the box).
# usage (on the box, from the repo root):  bash tools/collect_profiles.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
# front-end kernels only in the profiled process: no ELAS / MSA legs (their host thread pools, oracle loading and
# `make` children do not belong under rocprofv3), no CPU baseline, no tracking leg
ARGS="--workload frontend --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-track-leg --no-elas-leg $*"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py $ARGS > /tmp/prof_stats.log 2>&1
python3 $R/tools/kstats.py /tmp/prof_stats $OUT/${TAG}_kernel_stats.csv
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
timeout 400 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d /tmp/prof_pmc1 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d /tmp/prof_pmc2 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_pmc3 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_pmc4 -- python3 $R/bench.py $ARGS > /tmp/prof_pmc4.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc.json /tmp/prof_pmc1 /tmp/prof_pmc2 /tmp/prof_pmc3 /tmp/prof_pmc4 > $OUT/${TAG}_pmc.txt
tail -8 $OUT/${TAG}_pmc.txt | cut -c1-400

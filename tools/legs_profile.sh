#!/bin/bash
# rocprofv3 evidence for the dense-stereo rows (f-1 MSA, f-2 ELAS), on small dedicated runs (the legs of bench.py under a
# profiler take tens of minutes: hundreds of thousands of small launches).  Kernel stats + FETCH_SIZE / WRITE_SIZE passes.
# usage (on the box, from the repo root): bash tools/legs_profile.sh <tag>
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
# counters only for the library's own kernels (k_...): the renderer's hundreds of thousands of elementwise launches would
# each be serialised and read out otherwise (minutes per pass)
KRE='(^|[^a-zA-Z0-9_])k_[a-z]'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lp_*
E="$R/tools/elas_batch_bench.py --batch 64 --iters 2"   # (the urban1 pair replicated: no renderer kernels in the profile)
M="$R/tools/msa_profile_run.py"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_es -- python3 $E > /tmp/lp_es.log 2>&1
python3 $R/tools/kstats.py /tmp/lp_es $OUT/${TAG}_elas_kernel_stats.csv | head -20
timeout 150 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc FETCH_SIZE --output-format csv -d /tmp/lp_ef -- python3 $E > /tmp/lp_ef.log 2>&1
timeout 150 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc WRITE_SIZE --output-format csv -d /tmp/lp_ew -- python3 $E > /tmp/lp_ew.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/${TAG}_elas_pmc.json /tmp/lp_ef /tmp/lp_ew > $OUT/${TAG}_elas_pmc.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_ms -- python3 $M > /tmp/lp_ms.log 2>&1
python3 $R/tools/kstats.py /tmp/lp_ms $OUT/${TAG}_msa_kernel_stats.csv | head -20
timeout 150 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc FETCH_SIZE --output-format csv -d /tmp/lp_mf -- python3 $M > /tmp/lp_mf.log 2>&1
timeout 150 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc WRITE_SIZE --output-format csv -d /tmp/lp_mw -- python3 $M > /tmp/lp_mw.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/${TAG}_msa_pmc.json /tmp/lp_mf /tmp/lp_mw > $OUT/${TAG}_msa_pmc.txt
tail -2 /tmp/lp_es.log /tmp/lp_ms.log | cut -c1-200

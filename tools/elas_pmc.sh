#!/bin/bash
# SQ counters of the dense ELAS kernels (tools/elas_batch_bench.py --batch 64): what bounds each of them
# usage (on the box, from the repo root): bash tools/elas_pmc.sh <tag>
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
KRE='(^|[^a-zA-Z0-9_])k_[a-z]'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ep_*
E="$R/tools/elas_batch_bench.py --batch 64 --iters 2"
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ep_s -- python3 $E > /tmp/ep_s.log 2>&1
python3 $R/tools/kstats.py /tmp/ep_s $OUT/${TAG}_elas_kernel_stats.csv | head -24
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc $P1 --output-format csv -d /tmp/ep_1 -- python3 $E > /tmp/ep_1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc $P2 --output-format csv -d /tmp/ep_2 -- python3 $E > /tmp/ep_2.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/${TAG}_elas_sq_pmc.json /tmp/ep_1 /tmp/ep_2 > $OUT/${TAG}_elas_sq_pmc.txt
tail -2 /tmp/ep_s.log | cut -c1-200
python3 - <<PY
import json
d = json.load(open("$OUT/${TAG}_elas_sq_pmc.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if "SQ_WAVES" not in v: continue
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print("%-22s waves %8.0f  valu/wave %7.0f  all/wave %7.0f  ipqc %.3f  valu_busy %.3f  wait_any %.2f  lds_conf %.2f  vmem/wave %.0f  cycles %.0f" % (
        k, v["SQ_WAVES"], v["SQ_INSTS_VALU"] / v["SQ_WAVES"], (v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"]) / v["SQ_WAVES"],
        (v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"]) / max(v["SQ_WAVE_CYCLES"], 1),
        v["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * max(cyc, 1)), v.get("SQ_WAIT_ANY", 0) / max(v["SQ_WAVE_CYCLES"], 1),
        v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), v.get("SQ_INSTS_VMEM", 0) / v["SQ_WAVES"], cyc))
PY

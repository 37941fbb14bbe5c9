#!/bin/bash
# Round 5: which hardware counter separates a fast placement of pass B's record buffer from a slow one?  tools/probes/placement_probe (stand-alone,
# no Python) times pass B's write pattern — partition-major as the engine lays it out, workgroup-major, sequential — and a TLB-reach pointer chase on
# twelve candidate allocations held together; then the same command under rocprofv3 --kernel-trace --pmc, one counter set per run (no other tracing).
# usage: tools/gpu_placement_r5.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
P=$R/tools/probes/placement_probe
cd /tmp; export TMPDIR=/tmp
timeout 120 $P > $O/probe_plain.log 2>&1
cat $O/probe_plain.log
# the effect present?  (slowest partition-major time >= 4 % above the fastest)
python3 - "$O/probe_plain.log" <<'PY' || { echo "no placement effect on this box: counter passes skipped"; exit 0; }
import sys
ms = [float(l.split()[2]) for l in open(sys.argv[1]) if l[:1].isdigit()]
print("pm spread: min %.4f max %.4f" % (min(ms), max(ms)))
sys.exit(0 if ms and max(ms) >= 1.04 * min(ms) else 1)
PY
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_UTCL2_BUSY" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o x -- $P 100000000 782 12 240000000 1 0 > $O/probe_pmc$i.log 2>&1
  grep -E "^[0-9]+ +0x" $O/probe_pmc$i.log | head -12
done
cd $R
python3 tools/placement_counters.py $O > $O/placement_counters.txt 2>&1
cat $O/placement_counters.txt
rm -rf $O/p[0-9]*

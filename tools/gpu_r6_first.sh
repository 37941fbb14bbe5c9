#!/bin/bash
# Round-6 first call: what decides the round's deletions.  (1) one-synchronisation form vs the plain form in ONE engine (placement off: same
# buffers for both), C2 and C4; (2) the first job of an engine and jobs on fresh column buffers, with and without the placement search;
# (3) the HIP API time of the first job (hipMalloc, module load) from rocprofv3 --hip-trace.    usage: tools/gpu_r6_first.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 300 python tools/ab_one_engine.py --config c2 --base placement=never > $O/ab1_c2_one_sync.log 2>&1
timeout 300 python tools/ab_one_engine.py --config c4 --base placement=never > $O/ab1_c4_one_sync.log 2>&1
timeout 300 python tools/cold_probe.py --config c2 --plan placement=never > $O/cold_c2_noplace.log 2>&1
timeout 300 python tools/cold_probe.py --config c2 > $O/cold_c2_auto.log 2>&1
timeout 300 python tools/cold_probe.py --config c4 --plan placement=never > $O/cold_c4_noplace.log 2>&1
timeout 300 python tools/cold_probe.py --config c4 > $O/cold_c4_auto.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/ht -o cold -- python $R/tools/cold_probe.py --config c2 --plan placement=never --tables 1 --jobs 2 > $O/ht.log 2>&1
for f in $(find $O/ht -name "*stats.csv"); do cp $f $O/cold_$(basename $f); done
rm -rf $O/ht
cd $R
cat $O/ab1_*.log $O/cold_c*.log
head -25 $O/cold_*hip_api_stats.csv 2>/dev/null

#!/bin/bash
# kernel-trace stats of one bench config for the shipped library and for variant builds.   usage: tools/gpu_r5_kstats_variants.sh <tag> <config> <grep pattern> [variant ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; CFG=$2; PAT=$3
shift; shift; shift
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs"
kt() {  # name, command...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- "$@" > $O/kt_$n.log 2>&1
  f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/kt_$n
}
kt ${CFG}_shipped $B --config $CFG --steps 5 --warmup 1
for v in "$@"; do TAD_LIBRARY_PATH=$R/theia_amd/lib/variants/libtad_$v.so kt ${CFG}_$v $B --config $CFG --steps 5 --warmup 1; done
grep -H "$PAT" $O/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200

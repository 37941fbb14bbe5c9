R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4p1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in shipped nocompare notable; do
  if [ $v = shipped ]; then unset TAD_LIBRARY_PATH; else export TAD_LIBRARY_PATH=$R/theia_amd/lib/variants/libtad_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o x -- python $R/tools/strings_bench.py --shapes low --steps 5 --host-rows 100000 --arrow-rows 100000 > $O/$v.log 2>&1
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${v}_kernel_stats.csv; rm -rf $O/kt
  echo == $v; tail -1 $O/$v.log | cut -c1-200; head -4 $O/${v}_kernel_stats.csv | cut -c1-160
done

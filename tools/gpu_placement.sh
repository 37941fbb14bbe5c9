#!/bin/bash
# does keeping the best of four placements of the record buffer (tad_capi.cpp:place_recs) move pass B?  tools/ab_engines.py, eight engine instances
# per process, with the calibration (cal) and without (nocal); measurement builds with -DTAD_TRACE_ALLOC print the candidates' probe times
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for v in nocal cal; do
  for c in c2 c4; do
    TAD_LIBRARY_PATH=$R/theia_amd/lib/variants/libtad_$v.so timeout 300 python tools/ab_engines.py --config $c --engines 8 > $O/${v}_$c.log 2>&1
    echo == $v $c; grep "pass B per engine" $O/${v}_$c.log; grep "tad placement" $O/${v}_$c.log | awk '{print $4, $6, $8, $9, $10, $11, $12}' | head -40
  done
done

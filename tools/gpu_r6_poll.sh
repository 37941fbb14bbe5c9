#!/bin/bash
# ARIMA fit: what the per-cycle poll of the pause word costs at C3 (same process, alternating engines, one library build per variant)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
V=theia_amd/lib/variants
timeout 900 python tools/ab_plans.py --config c3 --rounds 4 --steps 2 --variants "every=;poll4=lib:$V/libtad_poll4.so;poll16=lib:$V/libtad_poll16.so;never=lib:$V/libtad_nopoll.so" > $O/ab_c3_poll.log 2>&1
cat $O/ab_c3_poll.log

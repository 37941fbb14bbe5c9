#!/bin/bash
# which boxes show the placement classes?  device state (clocks, temperatures, power, memory partition) next to the probe's table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
{
  rocm-smi --showclocks --showtemp --showpower --showmemuse --showperflevel --showmemorypartition --showcomputepartition 2>&1 | grep -v "^$" | head -60
  (amd-smi metric -g 0 --clock --temperature --power --mem-usage 2>/dev/null || true) | head -80
  cd /tmp; timeout 120 $R/tools/probes/placement_probe
  rocm-smi --showclocks --showtemp 2>&1 | grep -v "^$" | head -30
} > $O/box.log 2>&1
cat $O/box.log

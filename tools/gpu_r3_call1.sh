#!/bin/bash
# round 3, first GPU call: default parity after the ARIMA sign change, then every queued switch (parity + same-box A/B)
cd /root/repo
mkdir -p gpurun_out/r3c1
O=gpurun_out/r3c1
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest_default.log 2>&1
for s in c3 c2 c4; do
  ( timeout 1500 bash tools/gpu_queued_ab.sh $s ) > $O/ab_$s.log 2>&1
done
tail -n 40 $O/*.log

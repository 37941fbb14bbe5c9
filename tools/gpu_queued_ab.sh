#!/bin/bash
# First GPU calls for the work that was written without GPU time (switches, default off; parity checked on the host emulator,
# tools/hipemu/check_experiments.py; the default path's kernels are ISA-identical to the last measured commit, tools/isa_diff.py).
# Per section: the parity tests with the switch on, then same-box A/B of the bench lines.  usage: tools/gpu_queued_ab.sh c2|c3|c4|sparse|all
#   c2      TAD_META_PREFETCH=1        software-pipelined sampled histogram in pass A
#           TAD_EWMA_FUSED=1           sigma + detector + compaction + emit in one kernel (decoupled look-back); every wait under `timeout`
#   c3      TAD_ARIMA_FILTER=collapsed collapsed Kalman recursion, four chains jointly with a batched inversion (the tests switch the
#                                      oracle with the same variable); TAD_ARIMA_WAVES=2|3|4 wavefronts per SIMD
#   c4      TAD_DBSCAN_TILESTATS=1|2   per-round key statistics from pass C; 2: key rounds + settled keys' grid columns not written
#           TAD_DBSCAN_WAVELIST=1      exact pair tests with one wavefront per listed key;  TAD_TWO_LEVEL=1 (measured plan) for comparison
#   sparse  the length-class tests (default path for skewed sparse tables; tests/test_gpu_sparse.py)
cd /root/repo
what=${1:-all}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('$1', round(d['ms_per_step'],3), 'ms; meta', round(p['ms_meta'],3), 'stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'detect', round(p['ms_detect_and_emit'],3), 'anomalies', d['result']['anomalies'])"; }
c2() { env $1 timeout 90 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | line "C2 [$1]"; }
c4() { env $1 timeout 90 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | line "C4 [$1]"; }
c3() { env $1 timeout 180 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line "C3 [$1]"; }
parity() { echo "== parity with $1: $2"; env $1 timeout ${3:-300} python -m pytest $2 -m gpu -x -q ${4:+-k "$4"} 2>&1 | tail -2; }

if [ "$what" = c2 ] || [ "$what" = all ]; then
  parity TAD_META_PREFETCH=1 "tests/test_gpu_parity.py tests/test_gpu_random.py" 300 "not arima"
  parity TAD_EWMA_FUSED=1 "tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_job.py" 300 "not arima"
  parity "TAD_EWMA_FUSED=1 TAD_EMIT_CAP=300" "tests/test_gpu_random.py" 200
  for r in 1 2; do c2 TAD_EWMA_FUSED=0; c2 TAD_META_PREFETCH=1; c2 TAD_EWMA_FUSED=1; c2 "TAD_EWMA_FUSED=1 TAD_META_PREFETCH=1"; done
fi
if [ "$what" = c3 ] || [ "$what" = all ]; then
  parity TAD_ARIMA_FILTER=collapsed "tests/test_gpu_arima.py tests/test_gpu_job.py" 600 "arima or e2e"
  parity TAD_ARIMA_FILTER=collapsed "tests/test_gpu_fullsize.py" 900 "c3"
  c3 TAD_ARIMA_FILTER=general
  for w in 2 3 4; do c3 "TAD_ARIMA_FILTER=collapsed TAD_ARIMA_WAVES=$w"; done
fi
if [ "$what" = c4 ] || [ "$what" = all ]; then
  for e in TAD_DBSCAN_TILESTATS=1 TAD_DBSCAN_WAVELIST=1 "TAD_DBSCAN_TILESTATS=2 TAD_DEBUG_PLAN=1" "TAD_TWO_LEVEL=1 TAD_DBSCAN_TILESTATS=2"; do
    parity "$e" "tests/test_gpu_parity.py tests/test_gpu_random.py" 300 "not arima"
  done
  parity "TAD_DBSCAN_TILESTATS=2 TAD_DBSCAN_WAVELIST=1" "tests/test_gpu_fullsize.py" 600 "c4"
  for r in 1 2; do
    c4 TAD_DBSCAN_TILESTATS=0; c4 TAD_DBSCAN_TILESTATS=1; c4 TAD_DBSCAN_WAVELIST=1; c4 TAD_DBSCAN_TILESTATS=2
    c4 "TAD_DBSCAN_TILESTATS=2 TAD_DBSCAN_WAVELIST=1"; c4 TAD_TWO_LEVEL=1; c4 "TAD_TWO_LEVEL=1 TAD_DBSCAN_TILESTATS=2 TAD_DBSCAN_WAVELIST=1"
  done
fi
if [ "$what" = sparse ] || [ "$what" = all ]; then
  parity TAD_NONE=1 "tests/test_gpu_sparse.py" 600
fi

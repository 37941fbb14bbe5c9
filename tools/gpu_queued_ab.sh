#!/bin/bash
# First GPU call for the experiments that were written without GPU time (env switches, default off; parity checked on the
# host emulator, tools/hipemu): the parity tests with each switch on, then same-box A/B of the bench lines.
#   TAD_META_PREFETCH=1      C2: software-pipelined sampled histogram in pass A
#   TAD_DBSCAN_TILESTATS=1   C4: DBSCAN scan from pass C's per-round key statistics
#   TAD_DBSCAN_TILESTATS=2   C4: pass C's rounds split the partition by key sub-range (whole series per tile) and do not write the columns of settled keys
#   TAD_DBSCAN_WAVELIST=1    C4: exact pair tests with one wavefront per listed key
#   TAD_ARIMA_FILTER=collapsed   C3: ARIMA likelihood by the collapsed recursion (2.3x fewer instructions per Kalman step);
#                                TAD_ARIMA_WAVES=2|3|4 wavefronts per SIMD (the tests switch the oracle with the same variable)
#   TAD_EWMA_FUSED=1         C2: sigma + detector + compaction + emit in one kernel (decoupled look-back, no count pass, one host sync
#                            fewer); run under `timeout`: the look-back has only ever run on the host emulator
cd /root/repo
for e in TAD_META_PREFETCH=1 TAD_DBSCAN_TILESTATS=1 TAD_DBSCAN_WAVELIST=1; do
  echo "== parity with $e"
  env $e timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not arima and not c3" 2>&1 | tail -2
done
c2() { env $1 timeout 60 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('C2 [$1]', round(d['ms_per_step'],3), 'ms; meta', round(p['ms_meta'],3), 'stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'detect', round(p['ms_detect_and_emit'],3))"; }
c4() { env $1 timeout 60 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('C4 [$1]', round(d['ms_per_step'],3), 'ms; stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'detect', round(p['ms_detect_and_emit'],3), d['result']['anomalies'])"; }
for r in 1 2; do c2 TAD_META_PREFETCH=0; c2 TAD_META_PREFETCH=1; done
echo "== parity with TAD_EWMA_FUSED=1 (every EWMA job after the first of an engine takes the fused kernel)"
env TAD_EWMA_FUSED=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_job.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not arima and not c3 and not c4" 2>&1 | tail -2
for r in 1 2; do c2 TAD_EWMA_FUSED=0; c2 TAD_EWMA_FUSED=1; c2 "TAD_EWMA_FUSED=1 TAD_META_PREFETCH=1"; done
echo "== parity with TAD_DBSCAN_TILESTATS=2 (full-size C4: every point)"
env TAD_DBSCAN_TILESTATS=2 TAD_DEBUG_PLAN=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not arima and not c3 and not c2" 2>&1 | tail -2
echo "== parity with TAD_TWO_LEVEL=1 TAD_DBSCAN_TILESTATS=2"
env TAD_TWO_LEVEL=1 TAD_DBSCAN_TILESTATS=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not arima and not c3 and not c2" 2>&1 | tail -2
for r in 1 2; do c4 TAD_DBSCAN_TILESTATS=0; c4 TAD_DBSCAN_TILESTATS=1; c4 TAD_DBSCAN_WAVELIST=1; c4 "TAD_DBSCAN_TILESTATS=1 TAD_DBSCAN_WAVELIST=1"; c4 TAD_DBSCAN_TILESTATS=2; c4 "TAD_DBSCAN_TILESTATS=2 TAD_DBSCAN_WAVELIST=1"; c4 TAD_TWO_LEVEL=1; c4 "TAD_TWO_LEVEL=1 TAD_DBSCAN_TILESTATS=2"; c4 "TAD_TWO_LEVEL=1 TAD_DBSCAN_TILESTATS=2 TAD_DBSCAN_WAVELIST=1"; done
echo "== ARIMA parity with the collapsed filter"
env TAD_ARIMA_FILTER=collapsed timeout 900 python -m pytest tests/test_gpu_arima.py tests/test_gpu_fullsize.py tests/test_gpu_job.py -m gpu -x -q -k "arima or c3 or e2e" 2>&1 | tail -2
c3() { env $1 timeout 120 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 [$1]', round(d['ms_per_step'],1), 'ms  kalman steps', d['result'].get('kalman_steps'))"; }
c3 TAD_ARIMA_FILTER=general
for w in 2 3 4; do c3 "TAD_ARIMA_FILTER=collapsed TAD_ARIMA_WAVES=$w"; done

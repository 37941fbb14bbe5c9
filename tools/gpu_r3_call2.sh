#!/bin/bash
# round 3, call 2: the cleaned-up library (ABI 7, one ARIMA contract, wave list) — parity, bench lines, ARIMA time split,
# rocprofv3 kernel stats, C2 pass-B A/B exact vs sampled histogram
cd /root/repo
O=gpurun_out/r3c2; mkdir -p $O
V=$PWD/theia_amd/lib/variants
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest.log 2>&1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('$1', round(d['ms_per_step'],3), 'ms; meta', round(p['ms_meta'],3), 'stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'partB', round(d['roofline'].get('avg_kernel_ms',0),3), 'detect', round(p['ms_detect_and_emit'],3), 'anomalies', d['result']['anomalies'])"; }
{
for r in 1 2; do
  timeout 120 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line "C3 static"
  TAD_LIBRARY_PATH=$V/libtad_dynamic.so timeout 120 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line "C3 dynamic"
done
TAD_LIBRARY_PATH=$V/libtad_prof_static.so timeout 120 python bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "arima prof"
TAD_LIBRARY_PATH=$V/libtad_prof_dynamic.so timeout 120 python bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "arima prof"
for r in 1 2; do
  timeout 120 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | line "C2 default"
  timeout 120 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --plan histogram=exact 2>/dev/null | line "C2 exact-hist"
  timeout 120 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | line "C4 default"
  timeout 120 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline --plan partition_pass=wc 2>/dev/null | line "C4 wc-forced"
done
} > $O/ab.log 2>&1
cd /tmp; export TMPDIR=/tmp
prof() { # name, args...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -o $n -- python /root/repo/bench.py "$@" --no-cpu-baseline --no-other-configs > $O/prof_$n.log 2>&1
  f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
}
cd /root/repo
prof c2 --config c2 --steps 5 --warmup 1
prof c2_exact --config c2 --steps 5 --warmup 1 --plan histogram=exact
prof c4 --config c4 --steps 5 --warmup 1
prof c3 --config c3 --steps 1 --warmup 0
rm -rf $O/prof_*/
tail -n 30 $O/pytest.log $O/ab.log; head -12 $O/*_kernel_stats.csv | cut -c1-150

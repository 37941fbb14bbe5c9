#!/bin/bash
# adaptive queue depths in pass B (k_partition_wc): the library as built against the variant without them (tools/build_variants.py noadapt:TAD_ADAPTIVE_QUEUES=0)
out=gpurun_out/${1:-r6_s6}; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_factorize.py tests/test_gpu_job.py -x -q 2>&1 | tail -5 > $out/pytest_subset.log
cat $out/pytest_subset.log
for c in ${2:-c2}; do
  python tools/ab_plans.py --config $c --variants "adapt=;noadapt=lib:theia_amd/lib/variants/libtad_noadapt.so" --rounds 8 --steps 20 > $out/ab_${c}_adaptive.log 2>&1
  python tools/order_bench.py --config $c > $out/order_${c}_adapt.log 2>&1
  python tools/order_bench.py --config $c --library theia_amd/lib/variants/libtad_noadapt.so > $out/order_${c}_noadapt.log 2>&1
done
python tools/skew_check.py 100000000 0.0 0.1 0.5 > $out/skew_adapt.log 2>&1
TAD_LIBRARY_PATH=theia_amd/lib/variants/libtad_noadapt.so python tools/skew_check.py 100000000 0.0 0.1 0.5 > $out/skew_noadapt.log 2>&1
cat $out/ab_*.log $out/order_*.log $out/skew*.log | grep -v amdgpu.ids

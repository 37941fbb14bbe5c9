#!/bin/bash
# the row orders / key lifetimes / time windows a caller can bring (tools/order_bench.py); optional: a variant library to compare with
out=gpurun_out/${1:-r6_s11}; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_factorize.py tests/test_gpu_job.py tests/test_gpu_sparse.py -x -q 2>&1 | tail -5 > $out/pytest_subset.log
cat $out/pytest_subset.log
for c in ${2:-c2 c4}; do
  python tools/order_bench.py --config $c > $out/order_${c}.log 2>&1
  if [ -n "$3" ]; then python tools/order_bench.py --config $c --library $3 > $out/order_${c}_variant.log 2>&1; fi
done
cat $out/order_*.log | grep -v amdgpu.ids

#!/bin/bash
# the row orders / key lifetimes / time windows a caller can bring (tools/order_bench.py)
out=gpurun_out/${1:-r6_s5}; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "orders_a_caller or hot_key or synthetic_tables or sampled_histogram" 2>&1 | tail -5 > $out/pytest_subset.log
for c in c2 c4; do
  python tools/order_bench.py --config $c > $out/order_${c}.log 2>&1
done
cat $out/pytest_subset.log $out/order_*.log | grep -v amdgpu.ids

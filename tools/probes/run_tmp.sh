timeout 1500 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "triplet" 2>&1 | tail -3
for i in 1 2; do timeout 200 python tools/kernel_bench.py --only "tricol" 2>/dev/null | tail -1; done

timeout 1200 python -m pytest tests/test_sparse_contraction.py tests/test_fullsize_and_host_rules.py tests/test_device_group.py -m gpu -q -x 2>&1 | tail -4
MIK_FUZZ_CASES=600 timeout 900 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -3

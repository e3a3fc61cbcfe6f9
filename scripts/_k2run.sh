timeout 900 python -m pytest tests/test_sparse_contraction.py tests/test_hip_parity.py -m gpu -q -x -k "geograph or geo_ok2d or sparse_path_on_the_reference" 2>&1 | tail -5

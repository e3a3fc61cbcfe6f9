timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "lean_exp or mw or moving" 2>&1 | tail -4

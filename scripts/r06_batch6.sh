#!/bin/bash
REPO=$PWD; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_device_group.py tests/test_exchange_deadline.py -m gpu -q --tb=short ) > $OUT/pytest_subset.txt 2>&1
grep -E "passed|failed|^real|^FAILED|^E  " $OUT/pytest_subset.txt | cut -c1-400 | head -12
timeout 900 python scripts/r06_diag_mw10.py > $OUT/diag_mw10.txt 2>&1; cat $OUT/diag_mw10.txt | cut -c1-300

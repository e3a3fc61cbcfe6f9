#!/bin/bash
REPO=$PWD; export TMPDIR=/tmp
timeout 600 python scripts/r06_diag_group.py > $OUT/diag_group.txt 2>&1; cat $OUT/diag_group.txt | cut -c1-330
timeout 900 python scripts/r06_diag_mw10.py > $OUT/diag_mw10.txt 2>&1; cat $OUT/diag_mw10.txt | cut -c1-400

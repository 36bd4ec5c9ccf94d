#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 600 python -m pytest tests/test_dropin.py -q -m gpu --timeout 300 2>&1 | tail -15
./tests/dropin/_bin/ref_test_kcore --file=tests/golden/kcore_test.mtx 2>&1 | tail -5

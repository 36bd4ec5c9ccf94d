#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x 2>&1 | tail -4

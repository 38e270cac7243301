#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "timed\|per-step"; done

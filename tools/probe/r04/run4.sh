#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/probe
for d in 0 1 2 3; do for v in pre_sweep_d$d pre_sweep_ar_d$d; do echo "== $v"; timeout -k 5 120 ./$v 21 1024 2>&1 | grep -v mismatch | tail -2; done; done

#!/bin/bash
# E-phase ablations of the prefiltered pass: tools/probe/ab/e_ab.sh <variant>...  ("base" = product build, run first)
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset ECOZ2VQ_LIB; else export ECOZ2VQ_LIB=$R/tools/probe/ab/$v/libecoz2vq.so; fi
  timeout -k 10 200 python3 $R/tools/probe/e_phase.py 2>&1 | grep "kernel ms" || exit 1
done

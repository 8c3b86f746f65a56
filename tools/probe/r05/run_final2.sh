#!/bin/bash
# round 5, final sources: validation (whole GPU suite, smoke, fuzz) and then the measurement set of run_final.sh
cd $GRAFT_REPO_ROOT
bash tools/probe/r05/run_i.sh || exit 1
bash tools/probe/r05/run_final.sh

#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== simple loop, M = 1024"; ECOZ2_VQ_PRE_SIMPLE_LOOP=1 timeout -k 10 120 python tools/probe/r04/dbg1.py 1024 2>&1 | tail -5
echo "== rotating loop, M = 1024"; AMD_LOG_LEVEL=1 timeout -k 10 120 python tools/probe/r04/dbg1.py 1024 2>&1 | tail -15
echo "== rotating loop, M = 128 (MT = 4)"; timeout -k 10 120 python tools/probe/r04/dbg1.py 128 2>&1 | tail -5
dmesg 2>/dev/null | tail -5

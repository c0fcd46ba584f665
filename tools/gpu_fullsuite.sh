#!/bin/bash
# the whole GPU suite as the driver runs it, then smoke()
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/suite
mkdir -p $out
cd $R
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $out/pytest.txt 2>&1
echo "pytest rc $?" >> $out/pytest.txt
tail -6 $out/pytest.txt
timeout -k 10 300 python __graft_entry__.py smoke > $out/smoke.txt 2>&1
echo "smoke rc $?"; tail -3 $out/smoke.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04l
mkdir -p $out
cd $R
MPCASM_LTI=1 MPCASM_JIT=1 timeout -k 10 300 python tools/stamp_resident.py 4096 > $out/stamps.txt 2>&1
grep -E "set-up" $out/stamps.txt
MPCASM_LTI=0 MPCASM_JIT=1 timeout -k 10 300 python tools/stamp_resident.py 4096 > $out/stamps0.txt 2>&1
echo "-- S, U read from memory (no tables generated on chip)"
grep -E "set-up" $out/stamps0.txt

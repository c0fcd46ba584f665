#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04j
mkdir -p $out
cd $R
for percu in 2 1; do
  echo "== per CU $percu" >> $out/stamps.txt
  MPCASM_PER_CU=$percu MPCASM_LTI=1 MPCASM_JIT=1 timeout -k 10 300 python tools/stamp_resident.py 4096 >> $out/stamps.txt 2>&1
done
grep -E "==|set-up|grid" $out/stamps.txt

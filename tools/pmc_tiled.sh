#!/bin/bash
# counters of the tiled kernel on C4 (gpurun -- 'bash tools/pmc_tiled.sh [batch]'): several passes
B=${1:-1024}
WHAT=${2:-all}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_tiled
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $out/$name -o p --output-format csv -- python3 $R/tools/run_tiled_only.py $B 2 1 $WHAT > $out/$name.log 2>&1 || return 1
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU || exit 1
pass b SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS || exit 1
pass c SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_SMEM || exit 1
pass d SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES || true
cd $R && python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_tiled")
for name in "abcd":
    for f in glob.glob(out + "/%s/**/p_counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            for key in ("toeplitz_assemble", "tiled_assemble", "lti_tables", "compose_d"):
                if key in k:
                    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: "%.4g" % (sorted(v)[len(v) // 2]) for c, v in cs.items()})
PY

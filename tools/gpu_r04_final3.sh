#!/bin/bash
# end of round 4: the whole GPU suite + smoke(), then the default bench line
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu_fullsuite.sh
mkdir -p gpurun_out/prof_r04
python3 bench.py > gpurun_out/prof_r04/bench_plain.json 2> gpurun_out/prof_r04/bench_plain.log; echo "bench rc $?"; wc -c gpurun_out/prof_r04/bench_plain.json

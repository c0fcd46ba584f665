#!/bin/bash
# The round's profile set, run on the GPU box (gpurun -- 'bash tools/profile_round.sh r01'):
# rocprofv3 kernel stats of the bench command, the HBM traffic counters in passes of their
# own, the bench lines, the other configurations.  tools/summarize_profile.py condenses the
# directory into profiles/.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}   # (as the package sets it; under rocprofv3 the runtime starts before python)
rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 > $out/bench_stats.json 2> $out/stats.log || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --steps 20 > $out/bench_fetch.json 2> $out/fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --streams 1 --steps 20 > $out/bench_write.json 2> $out/write.log || exit 1
cd $R
python3 bench.py > $out/bench_plain.json 2> $out/bench_plain.log || exit 1
python3 bench.py --no-cpu-baseline --no-extras --streams 1 > $out/bench_one_stream.json 2>> $out/bench_plain.log || exit 1
python3 bench.py --no-cpu-baseline --no-extras --two-kernels > $out/bench_two_kernels.json 2>> $out/bench_plain.log || exit 1
python3 bench.py --no-cpu-baseline --no-extras --batch 65536 --steps 200 --warmup 20 --rotate 1 > $out/bench_B65536.json 2>> $out/bench_plain.log || exit 1
python3 tools/bench_configs.py > $out/configs.txt 2>&1 || exit 1
python3 tools/bench_next_rows.py > $out/next_rows.txt 2>&1 || exit 1

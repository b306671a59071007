#!/bin/bash
# round 5: the default bench, the round's profiles (kernel trace + PMC of the bench and of the association sweep), kernel stats of the SlideMatch legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { echo "bench failed"; tail -5 gpurun_out/r5_bench.err; exit 1; }
echo "bench done"
BENCH_ARGS="--no-place-leg" timeout -k 10 1000 bash tools/profile_round.sh r05 || { echo "profile failed"; exit 1; }
echo "profiles done"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05/ptrace -- python3 tools/place_prof.py > gpurun_out/prof_r05/ptrace.log 2>&1 || { echo "place trace failed"; exit 1; }
f=$(find gpurun_out/prof_r05/ptrace -name "*kernel_stats.csv" | head -1)
cp "$f" profiles/r05_place_kernel_stats.csv && cp "$f" gpurun_out/profiles_r05/r05_place_kernel_stats.csv
rm -rf gpurun_out/prof_r05/ptrace
echo "place trace done"

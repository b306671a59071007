#!/bin/bash
# round 5: the default bench, the round's profiles (kernel trace + PMC of the bench and of the association sweep), kernel stats of the SlideMatch legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { echo "bench failed"; tail -5 gpurun_out/r5_bench.err; exit 1; }
echo "bench done"
BENCH_ARGS="--no-place-leg" timeout -k 10 1000 bash tools/profile_round.sh r05 || { echo "profile failed"; exit 1; }
echo "profiles done"
bash tools/gpu_place_traces.sh
echo "place traces done"

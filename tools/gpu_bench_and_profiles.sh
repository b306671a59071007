#!/bin/bash
# the default bench and the round's profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r18_bench.json 2> gpurun_out/r18_bench.err || { echo "bench failed"; tail -5 gpurun_out/r18_bench.err; exit 1; }
echo "bench done"
timeout -k 10 900 bash tools/profile_round.sh r04 || { echo "profile failed"; exit 1; }
echo "profiles done"

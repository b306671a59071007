#!/bin/bash
# cooperative CLIPPER tests, then the default bench and the round's profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_place.py -x -q -m gpu -s -k "clique" > gpurun_out/r13_place.log 2>&1
rc=$?
tail -5 gpurun_out/r13_place.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "place tests timed out"; exit 1; fi
timeout -k 10 600 python bench.py > gpurun_out/r13_bench.json 2> gpurun_out/r13_bench.err || { echo "bench failed"; tail -5 gpurun_out/r13_bench.err; exit 1; }
echo "bench done"
timeout -k 10 900 bash tools/profile_round.sh r04 || { echo "profile failed"; exit 1; }
echo "profiles done"

#!/bin/bash
# cooperative CLIPPER tests, then the whole GPU suite, the default bench and the round's profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_place.py -x -q -m gpu -s > gpurun_out/r11_place.log 2>&1
rc=$?
tail -5 gpurun_out/r11_place.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "place tests timed out"; exit 1; fi
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_place.py > gpurun_out/r11_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/r11_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "gpu tests timed out"; exit 1; fi

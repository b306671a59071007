#!/bin/bash
# the whole GPU suite and the smoke entry
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r21_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/r21_gpu.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"

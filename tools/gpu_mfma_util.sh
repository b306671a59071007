#!/bin/bash
# MFMA utilisation (PMC) of the Schur / factorisation kernels on the headline configuration: its own rocprofv3 pass (counters only)
set -o pipefail
mkdir -p gpurun_out/prof_r05 profiles gpurun_out/profiles_r05
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/prof_r05/pmc_mfma -- python3 bench.py --steps 20 --warmup 3 --probe 0 --no-cpu --no-parity --no-dense-leg --no-dense-relmeas --ingest-only --no-place-leg > gpurun_out/prof_r05/pmc_mfma.log 2>&1 || { tail -5 gpurun_out/prof_r05/pmc_mfma.log; exit 1; }
python3 tools/pmc_mfma_summary.py gpurun_out/prof_r05/pmc_mfma r05
cp profiles/r05_pmc_mfma_util.* gpurun_out/profiles_r05/
rm -rf gpurun_out/prof_r05/pmc_mfma

#!/bin/bash
# Round profiles (run on the MI355X box from the repo root):  bash tools/profile_round.sh r02
#   kernel-trace + stats of the default bench, then HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes: they do not fit one)
#   of the default bench (k_chol_step_batched, PCG kernels) and of the association sweep.  Summaries -> profiles/<round>_*.
set -o pipefail
R=${1:-r03}
OUT=gpurun_out/prof_$R
mkdir -p $OUT profiles
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 20 --warmup 3 --probe 0 --no-cpu --no-parity --no-dense-leg --no-dense-relmeas --ingest-only ${BENCH_ARGS}"      # BENCH_ARGS=--dense-profile: the dense-profile configuration
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -- $BENCH > $OUT/pmc_f.log 2>&1 || exit 1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -- $BENCH > $OUT/pmc_w.log 2>&1 || exit 1
echo "write done"
if [ -z "$SKIP_ASSOC" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/atrace -- python3 tools/assoc_sweep_prof.py > $OUT/atrace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/apmc_f -- python3 tools/assoc_sweep_prof.py > $OUT/apmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/apmc_w -- python3 tools/assoc_sweep_prof.py > $OUT/apmc_w.log 2>&1 || exit 1
echo "assoc done"
fi
python3 tools/profile_summary.py $OUT $R
python3 tools/trace_batched.py $OUT/trace 15 > profiles/${R}_bench_chol_step_batched_by_k.txt 2>&1
python3 tools/trace_exact.py $OUT/trace 15 > profiles/${R}_exact_pass_kernels.txt 2>&1
# the raw traces are hundreds of MB: keep the summaries only (profiles/ is what is committed; a copy goes back through gpurun_out/)
rm -rf $OUT/trace $OUT/pmc_f $OUT/pmc_w $OUT/atrace $OUT/apmc_f $OUT/apmc_w
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/

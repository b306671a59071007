set -o pipefail
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r4_run3_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run3_summary.txt; exit 1; fi; return 0; }
rm -f gpurun_out/r4_run3_summary.txt
step r4_trace 200 python -u tools/ll_trace.py 1000 3776
step r4_threads8b 500 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "eight_ranks"
export SLIDE_CHOL_LL=1
step r4_ll_bench2 400 python bench.py --steps 50 --warmup 10 --no-cpu --no-dense-relmeas
cat gpurun_out/r4_run3_summary.txt; tail -80 gpurun_out/r4_trace.log

set -o pipefail
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r4_run7_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run7_summary.txt; exit 1; fi; return 0; }
rm -f gpurun_out/r4_run7_summary.txt
step r4_wf 300 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "wildfire or incremental"
step r4_full_gpu2 1000 python -m pytest tests/ -q -m gpu
step r4_bench_b 400 python bench.py --steps 50 --warmup 10 --no-cpu
cat gpurun_out/r4_run7_summary.txt; tail -8 gpurun_out/r4_wf.log; tail -12 gpurun_out/r4_full_gpu2.log

set -o pipefail
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r4_run6_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run6_summary.txt; exit 1; fi; return 0; }
rm -f gpurun_out/r4_run6_summary.txt
step r4_full_gpu 1000 python -m pytest tests/ -x -q -m gpu
step r4_bench_a 400 python bench.py --steps 50 --warmup 10
cat gpurun_out/r4_run6_summary.txt; tail -5 gpurun_out/r4_full_gpu.log

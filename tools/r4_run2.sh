set -o pipefail
mkdir -p gpurun_out
step() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/r4_run2_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run2_summary.txt; exit 1; fi
  return 0
}
rm -f gpurun_out/r4_run2_summary.txt
step r4_ll_unit 420 python -u -m pytest -v tests/test_gpu_kernels.py -x -q -m gpu -k "left_looking or dense_spd"
export SLIDE_CHOL_LL=1
step r4_ll_tiny 600 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "matches_oracle_shards_and_the_joint_replica or with_relative_pose_factors"
step r4_ll_c4 600 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "c4_exact_joint_step_matches_oracle_shards_at_size or segmented_bands"
step r4_ll_bench 400 python bench.py --steps 50 --warmup 10 --no-cpu
unset SLIDE_CHOL_LL
step r4_threads8 600 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "eight_ranks"
cat gpurun_out/r4_run2_summary.txt

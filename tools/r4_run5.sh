set -o pipefail
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r4_run5_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run5_summary.txt; exit 1; fi; return 0; }
rm -f gpurun_out/r4_run5_summary.txt
step r4_threads8d 500 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "eight_ranks"
export SLIDE_LL_MODE=3
for m in 1 3; do
export SLIDE_CHOL_LL=$m
step r4_ll3_bench_m$m 300 python bench.py --steps 50 --warmup 10 --no-cpu --no-dense-relmeas --no-parity
done
export SLIDE_CHOL_LL=1
step r4_ll3_c4par 400 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "c4_exact_joint_step_matches_oracle_shards_at_size and cayley or segmented_bands"
unset SLIDE_CHOL_LL
python - <<'PY'
import json
for m in (1,3):
    try:
        z=json.loads(open(f'gpurun_out/r4_ll3_bench_m{m}.log').read().strip().splitlines()[-1])
        print(m, z["ms_per_step"], z["roofline"]["exact_joint_pass"]["stages_ms"])
    except Exception as e: print(m, "failed", e)
PY
cat gpurun_out/r4_run5_summary.txt; grep -E "Error|error" gpurun_out/r4_threads8d.log | head -5

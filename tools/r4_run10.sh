set -o pipefail
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r4_run10_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run10_summary.txt; exit 1; fi; return 0; }
rm -f gpurun_out/r4_run10_summary.txt
step r4_bits3 1000 python -m pytest tests/test_bench_config.py -q -m gpu -k "two_ranks or four_ranks or eight_ranks or dense_relative or c4_exact_joint_step_matches_oracle_shards_at_size and cayley"
step r4_bench_d 400 python bench.py --steps 50 --warmup 10 --no-cpu --no-dense-relmeas

cat gpurun_out/r4_run10_summary.txt; tail -5 gpurun_out/r4_bits3.log
python - <<'PY'
import json
for f in ("r4_bench_d","r4_bench_d0"):
    try:
        z=json.loads(open(f'gpurun_out/{f}.log').read().strip().splitlines()[-1])
        print(f, z["ms_per_step"], z["roofline"].get("exact_joint_pass",{}).get("stages_ms"))
    except Exception as e: print(f, "failed", e)
PY

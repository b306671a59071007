set -o pipefail
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r4_run8_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run8_summary.txt; exit 1; fi; return 0; }
rm -f gpurun_out/r4_run8_summary.txt
step r4_wf2 300 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "wildfire or incremental"
step r4_bits 1000 python -m pytest tests/test_bench_config.py -q -m gpu -k "two_ranks or four_ranks or eight_ranks or c4_exact_joint_step_matches_oracle_shards_at_size or matches_oracle_shards_and_the_joint_replica or with_relative_pose or dense_relative"
step r4_bench_c 400 python bench.py --steps 50 --warmup 10 --no-cpu
SLIDE_BENCH_BACKEND=gloo step r4_bench_g2 400 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-dense-relmeas
SLIDE_BENCH_BACKEND=gloo step r4_bench_g4 400 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu --no-dense-relmeas
cat gpurun_out/r4_run8_summary.txt; tail -12 gpurun_out/r4_bits.log
python - <<'PY'
import json
for f in ("r4_bench_c","r4_bench_g2","r4_bench_g4"):
    try:
        z=json.loads(open(f'gpurun_out/{f}.log').read().strip().splitlines()[-1])
        print(f, z["ms_per_step"], z.get("parity",{}).get("vs_n1_max_rel"), z["roofline"].get("exact_joint_pass",{}).get("stages_ms"))
    except Exception as e: print(f, "failed", e)
PY

set -o pipefail
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2; timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a gpurun_out/r4_run9_summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/r4_run9_summary.txt; exit 1; fi; return 0; }
rm -f gpurun_out/r4_run9_summary.txt
step r4_bits2 1000 python -m pytest tests/test_bench_config.py -q -m gpu -k "two_ranks or four_ranks or eight_ranks"
SLIDE_BENCH_BACKEND=gloo step r4_bench_g2 400 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-dense-relmeas
SLIDE_BENCH_BACKEND=gloo step r4_bench_g4 400 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu --no-dense-relmeas
cat gpurun_out/r4_run9_summary.txt; tail -12 gpurun_out/r4_bits2.log
python - <<'PY'
import json
for f in ("r4_bench_g2","r4_bench_g4"):
    try:
        z=json.loads(open(f'gpurun_out/{f}.log').read().strip().splitlines()[-1])
        print(f, z["ms_per_step"], z.get("parity",{}).get("vs_n1_max_rel"), z.get("cut_pass_ms"))
    except Exception as e: print(f, "failed", e)
PY

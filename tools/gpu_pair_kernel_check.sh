#!/bin/bash
# round 5: the pair kernel (two block columns per launch): unit tests (dense, bordered), step kernels vs pair kernel inside a C4 pass
# (SLIDE_PAIR_VERIFY), the exact-pass parity tests with it on, and the bench with / without
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_chol_bordered.py -x -q -k "dense_spd or bordered" > gpurun_out/r5_pair_unit.log 2>&1
rc=$?
tail -5 gpurun_out/r5_pair_unit.log
if [ $rc -ne 0 ]; then exit 1; fi
# step kernels vs pair kernel on the systems of a C4 exact pass, tile by tile (SLIDE_PAIR_VERIFY=1, one un-captured pass)
SLIDE_CHOL_PAIR=15 SLIDE_PAIR_VERIFY=1 timeout -k 10 200 python tests/gpu_scenarios.py pair_verify gpurun_out/r5_vfy_15.json C4 1 > gpurun_out/r5_vfy_15.log 2>&1 || { tail -3 gpurun_out/r5_vfy_15.log; exit 1; }
grep "pair verify" gpurun_out/r5_vfy_15.log | head -20
export SLIDE_CHOL_PAIR=${PAIR_MASK:-15}
timeout -k 10 700 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "exact_joint_step and not rccl and not eight_ranks and not four_ranks and not two_ranks" > gpurun_out/r5_pair_exact.log 2>&1
rc=$?
tail -15 gpurun_out/r5_pair_exact.log
if [ $rc -ne 0 ]; then exit 1; fi
python -c "import slide_slam_amd as s; print('pair timeouts', s.pair_timeouts())"
B="python bench.py --steps 100 --warmup 10 --no-cpu --no-parity --no-dense-leg --probe 0 --no-dense-relmeas"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["roofline"]["exact_joint_pass"]; print(sys.argv[1], round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in e["stages_ms"].items()})'
for v in SLIDE_CHOL_PAIR=0 SLIDE_CHOL_PAIR=15 SLIDE_CHOL_PAIR=1 SLIDE_CHOL_PAIR=2 SLIDE_CHOL_PAIR=4 SLIDE_CHOL_PAIR=8; do
  env $v timeout -k 10 300 $B 2> gpurun_out/r5_pair_bench.err | python -c "$pick" "$v" || exit 1
done

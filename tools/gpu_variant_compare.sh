#!/bin/bash
# library variants compared on the default bench: bash tools/gpu_variant_compare.sh "" brd8 brd16
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --steps 100 --warmup 10 --no-cpu --no-parity --no-dense-leg --probe 0 --no-dense-relmeas"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["roofline"]["exact_joint_pass"]; print(sys.argv[1] or "default", d["ms_per_step"], {k: round(v, 4) for k, v in e["stages_ms"].items()})'
for v in "$@"; do
  SLIDE_LIB_VARIANT=$v timeout -k 10 300 $B 2> gpurun_out/var_x.err | python -c "$pick" "$v" || exit 1
done

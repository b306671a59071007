#!/bin/bash
# border product: job-table orders / prefetch depth / resident workgroups compared on the default bench (stage times from the JSON line)
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --steps 100 --warmup 10 --no-cpu --no-parity --no-dense-leg --probe 0 --no-dense-relmeas"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["roofline"]["exact_joint_pass"]; print(sys.argv[1], round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in e["stages_ms"].items()})'
for v in "$@"; do
  env $v timeout -k 10 300 $B 2> gpurun_out/syrk_x.err | python -c "$pick" "$v" || exit 1
done

#!/bin/bash
# GPU vs oracle per pass under both charts, three sizes
set -o pipefail
mkdir -p gpurun_out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["parity"]["per_pass"])'
for p in C4tiny C3 C4; do
  for c in cayley expmap; do
    timeout -k 10 400 python bench.py --preset $p --chart $c --steps 5 --warmup 1 --no-dense-leg --probe 0 --no-dense-relmeas 2> gpurun_out/chart_x.err | python -c "$pick" "$p/$c" || { tail -3 gpurun_out/chart_x.err; exit 1; }
  done
done

#!/bin/bash
# round 5: wildfire vs oracle, association test on the shared generator, the bench with the SlideMatch / SlideGraph / CLIPPER legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -k "wildfire or assoc_sweep_batch_matches" > gpurun_out/r5_misc_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r5_misc_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --probe 0 --no-dense-leg > gpurun_out/r5_bench_legs.json 2> gpurun_out/r5_bench_legs.err
rc=$?
tail -3 gpurun_out/r5_bench_legs.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5_bench_legs.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'])
r = d['roofline']
for k in ('assoc', 'place', 'triangles', 'affinity', 'clipper'):
    v = r.get(k, {})
    print(k, json.dumps({kk: vv for kk, vv in v.items() if kk not in ('note', 'kernel', 'config')})[:1500])
PY
exit $rc

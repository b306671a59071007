#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "frame_by_frame_association or matches_oracle_shards_and_the_joint_replica" > gpurun_out/r5_assoc_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r5_assoc_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --probe 0 --no-dense-leg --no-place-leg --no-cpu > gpurun_out/r5_bench_assoc.json 2> gpurun_out/r5_bench_assoc.err
rc=$?
tail -3 gpurun_out/r5_bench_assoc.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5_bench_assoc.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'])
print(json.dumps(d.get('association'))[:1500])
PY
exit $rc

#!/bin/bash
# kernel stats of the SlideMatch sweep and of the SlideGraph / CLIPPER legs under rocprofv3, one process each (round 5: the three legs in
# one process ended in a SIGSEGV inside the exit handlers under the profiler, after the kernels had run)
mkdir -p gpurun_out/prof_r05 gpurun_out/profiles_r05 profiles
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for leg in place graph; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05/ptrace_$leg -- python3 tools/place_prof.py $leg > gpurun_out/prof_r05/ptrace_$leg.log 2>&1
  echo "$leg rc=$?"
  f=$(find gpurun_out/prof_r05/ptrace_$leg -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" profiles/r05_${leg}_kernel_stats.csv; cp "$f" gpurun_out/profiles_r05/r05_${leg}_kernel_stats.csv; head -5 "$f" | cut -c1-200; fi
  rm -rf gpurun_out/prof_r05/ptrace_$leg
done

R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for nap in ${NAPS:-8}; do
for n in 1280 3776; do
  export SLIDE_BWD_NAP=$nap
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bw$n -o bw -- python3 $R/tools/chol_big.py $n > $R/gpurun_out/bw$n.log 2>&1 || exit 1
  echo nap $nap n $n; grep "k_chol_bwd_chain" $R/gpurun_out/bw$n/bw_kernel_stats.csv | sed "s/([^)]*)//g"
  rm -rf $R/gpurun_out/bw$n
done; done

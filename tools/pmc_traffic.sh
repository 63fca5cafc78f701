# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as the guide prescribes) of the bench command's kernels.
# usage (GPU box): bash tools/pmc_traffic.sh r02  -> gpurun_out/pmc_traffic_r02.txt
R=$PWD; tag=${1:-x}; cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/pmc_traffic_$tag.txt; : > $out
run() { n=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmct_$n -o p -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events > /tmp/pmct_$n.log 2>&1
  f=$(find /tmp/pmct_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)" >> $out; python $R/tools/pmc_summary.py $f render_ tile_sort scatter preprocess split_kernel wgrad_act_kernel head_bwd conv5_ >> $out 2>&1
}
run c FETCH_SIZE
run d WRITE_SIZE

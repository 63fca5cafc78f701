# PMC passes over tools/bench_raster.py (rasterizer only, one frame per launch): render kernels' issue / wait / cache counters.
# usage (GPU box): bash tools/pmc_raster.sh <tag>   -> gpurun_out/pmc_raster_<tag>.txt
R=$PWD; tag=${1:-x}; cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/pmc_raster_$tag.txt; : > $out
run() { n=$1; shift
  REPS=10 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcr_$n -o p -- python $R/tools/bench_raster.py > /tmp/pmcr_$n.log 2>&1
  f=$(find /tmp/pmcr_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)" >> $out; python $R/tools/pmc_summary.py $f render_ tile_sort scatter preprocess >> $out 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
run b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run e FETCH_SIZE
run f WRITE_SIZE
run g TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum
tail -3 /tmp/pmcr_a.log >> $out

# PMC passes over the rasterizer micro-benchmark's child run of ONE set (tools/bench_raster.py --child <set> <size>: 12 forward +
# backward calls, 2 frames per launch): issue / wait / LDS / cache counters of every rasterizer kernel.
# usage (GPU box): bash tools/pmc_raster.sh <tag> [set] [size]   -> gpurun_out/pmc_raster_<tag>.txt
R=$PWD; tag=${1:-x}; S=${2:-avatar_3mm}; Z=${3:-200k}; cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/pmc_raster_$tag.txt; echo "set $S $Z" > $out
run() { n=$1; shift
  rm -rf /tmp/pmcr_$n
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcr_$n -o p -- python $R/tools/bench_raster.py --child $S $Z > /tmp/pmcr_$n.log 2>&1
  f=$(find /tmp/pmcr_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)" >> $out; python $R/tools/pmc_summary.py $f render_ tile_sort tile_merge tile_scan scatter preprocess >> $out 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
run b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run h SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run g TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum

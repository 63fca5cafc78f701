# PMC passes over tools/microbench_upz.py (map-path kernels; GPU box).  usage: bash tools/pmc_upz.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r05pmc
run() { # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcu_$n -o p -- python $R/tools/microbench_upz.py > /tmp/pmcu_$n.log 2>&1
  f=$(find /tmp/pmcu_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)"; python $R/tools/pmc_summary.py $f upsample_z dz_upsample layer_fwd_spec rowgemm
}
{
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run b FETCH_SIZE
run c WRITE_SIZE
run d SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_SMEM
run e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run f TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
} 2>&1 | tee $R/gpurun_out/r05pmc/pmc_upz.txt

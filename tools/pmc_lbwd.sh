# PMC passes over tools/microbench_bwd_fused.py (dev tool, GPU box)
R=$PWD; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
run() { # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcl_$n -o p -- python $R/tools/microbench_bwd_fused.py > /tmp/pmcl_$n.log 2>&1
  f=$(find /tmp/pmcl_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)"; python $R/tools/pmc_summary.py $f layer_bwd mlp_bwd_split wgrad_split
}
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run b FETCH_SIZE
run c WRITE_SIZE
run d SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE

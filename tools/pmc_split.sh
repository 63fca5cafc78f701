# PMC passes over tools/split_check.py (dev tool, GPU box): usage tools/pmc_split.sh [ONLY filter]
R=$PWD; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
run() { # name counters...
  n=$1; shift
  ONLY="$FILT" timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcs_$n -o p -- python $R/tools/split_check.py > /tmp/pmcs_$n.log 2>&1
  f=$(find /tmp/pmcs_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)"; python $R/tools/pmc_summary.py $f split_kernel mlp_fwd_kernel mlp_bwd_kernel wgrad_act_kernel
}
FILT="$1"
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU > $R/gpurun_out/pmcs_a.txt 2>&1
run b FETCH_SIZE > $R/gpurun_out/pmcs_b.txt 2>&1
run c WRITE_SIZE > $R/gpurun_out/pmcs_c.txt 2>&1
run d SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > $R/gpurun_out/pmcs_d.txt 2>&1
tail -2 /tmp/pmcs_a.log

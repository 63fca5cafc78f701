# PMC passes over tools/microbench_ssim.py (dev tool, GPU box); usage: tools/pmc_ssim.sh [GA_DEV value]
R=$PWD; cd /tmp; export TMPDIR=/tmp
[ -n "$1" ] && export GA_DEV=$1
run() { # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcs_$n -o p -- python $R/tools/microbench_ssim.py > /tmp/pmcs_$n.log 2>&1
  f=$(find /tmp/pmcs_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)"; python $R/tools/pmc_summary.py $f ssim_fwd ssim_bwd
}
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
run d SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA

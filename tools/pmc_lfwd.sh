# PMC passes over tools/lfwd_ablate.py (hidden-layer forward: one launch and the three-branch launch; GPU box).
# usage: bash tools/pmc_lfwd.sh  (needs build_v_lfwd/lfwd_a16: bash tools/lfwd_ablate.sh build)
R=$PWD; cd /tmp; export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  GA_DEV=lib_dir=$R/build_v_lfwd/lfwd_a16 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcf_$n -o p -- python $R/tools/lfwd_ablate.py > /tmp/pmcf_$n.log 2>&1
  f=$(find /tmp/pmcf_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)"; python $R/tools/pmc_summary.py $f layer_fwd_spec
}
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run b FETCH_SIZE
run c WRITE_SIZE
run d SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_MFMA

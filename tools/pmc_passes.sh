R=$PWD; cd /tmp; export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$n -o p -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events > /tmp/pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n ($*)"; python $R/tools/pmc_summary.py $f mlp_ wgrad_act_kernel head_bwd render_ tile_sort ssim_
}
run a GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU > $R/gpurun_out/pmc_a.txt 2>&1
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU > $R/gpurun_out/pmc_b.txt 2>&1
run c FETCH_SIZE > $R/gpurun_out/pmc_c.txt 2>&1
run d WRITE_SIZE > $R/gpurun_out/pmc_d.txt 2>&1
tail -3 /tmp/pmc_b.log

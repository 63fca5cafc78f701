# The round's judged artefacts in one GPU call -> gpurun_out/<tag>/ (copy what is kept into profiles/):
#   pytest_gpu.txt, smoke.txt, bench_driver_cmd.json   tools/run_gpu_checks.sh: full -m gpu suite, smoke(), the driver's bench command
#   kernel_stats.txt            rocprofv3 --kernel-trace --stats of the bench command (tools/kstats.sh)
#   gap_stats.txt               GPU busy fraction / gaps of that trace
#   pmc_traffic.txt             FETCH_SIZE / WRITE_SIZE, separate passes
#   sustained/                  2000-step run with a per-100-step series + rocm-smi samples
# usage: bash tools/collect_r03.sh r03_final
R=$PWD; tag=${1:-r03_final}; O=$R/gpurun_out/$tag; mkdir -p $O
bash tools/run_gpu_checks.sh $tag
bash tools/kstats.sh > $O/kernel_stats.txt 2>&1; tail -1 $O/kernel_stats.txt
f=$(find /tmp/prof_ks -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/gap_stats.py $f > $O/gap_stats.txt 2>&1; head -1 $O/gap_stats.txt
( cd /tmp; export TMPDIR=/tmp; : > $O/pmc_traffic.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmct_$c
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmct_$c -o p -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events --no-fixed-batch > /tmp/pmct_$c.log 2>&1
    f=$(find /tmp/pmct_$c -name "*counter_collection.csv" | head -1)
    echo "== pass $c" >> $O/pmc_traffic.txt
    python $R/tools/pmc_summary.py $f render_ tile_sort scatter preprocess layer_bwd_spec layer_fwd_spec split_kernel head_bwd conv5_ ssim_ skin_dmats >> $O/pmc_traffic.txt 2>&1
  done )
grep -c mean $O/pmc_traffic.txt
bash tools/sustained.sh $O/sustained 2000 | head -2

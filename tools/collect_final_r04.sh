#!/bin/bash
# The round's judged artefacts in one GPU call -> gpurun_out/<tag>/ (copy what is kept into profiles/):
#   pytest_gpu.txt, smoke.txt, bench_driver_cmd.json   tools/run_gpu_checks.sh: full -m gpu suite, smoke(), the driver's bench command
#   kernel_stats.txt + gap_stats.txt                   rocprofv3 --kernel-trace --stats of the bench command (tools/kstats.sh)
#   pmc_traffic.txt, traffic.json                      FETCH_SIZE / WRITE_SIZE, separate passes
#   config2 / stage2 / config5 .json + _kernel_stats.txt   tools/collect_r04.sh
#   reference_loop.json                                the reference's train.py verbatim (tools/reference_loop.py)
#   sustained/                                         2000-step run with a per-100-step series
# usage: bash tools/collect_final_r04.sh r04_final
R=$PWD; tag=${1:-r04_final}; O=$R/gpurun_out/$tag; mkdir -p $O
bash tools/run_gpu_checks.sh $tag
bash tools/kstats.sh > $O/kernel_stats.txt 2>&1; tail -1 $O/kernel_stats.txt
f=$(find /tmp/prof_ks -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/gap_stats.py $f > $O/gap_stats.txt 2>&1; head -1 $O/gap_stats.txt
CMD="python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events --no-fixed-batch --no-secondary"
( cd /tmp; export TMPDIR=/tmp; : > $O/pmc_traffic.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmct_$c
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmct_$c -o p -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events --no-fixed-batch --no-secondary > /tmp/pmct_$c.log 2>&1
    f=$(find /tmp/pmct_$c -name "*counter_collection.csv" | head -1)
    echo "== pass $c" >> $O/pmc_traffic.txt
    python $R/tools/pmc_summary.py $f render_ tile_sort scatter preprocess layer_bwd_spec layer_fwd_spec split_kernel head_bwd conv5_ ssim_ skin_dmats >> $O/pmc_traffic.txt 2>&1
  done )
python tools/traffic_json.py $O/pmc_traffic.txt $O/traffic.json "$CMD" | cut -c1-200
bash tools/collect_r04.sh $tag config2 stage2 config5
timeout 500 python tools/reference_loop.py --iters 320 --out $O/reference_loop.json > $O/reference_loop.log 2>&1; python -c "import json;d=json.load(open('$O/reference_loop.json'));print('reference loop', round(d['iters_per_s_mean'],1),'it/s median ms',round(d['ms_per_iter_median'],2))"
bash tools/sustained.sh $O/sustained 2000 | head -2

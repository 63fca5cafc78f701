# Collects the round's judged artefacts on the GPU box into gpurun_out/ (copy what is kept into profiles/):
#   <tag>_bench.json                     python bench.py (headline config, default flags)
#   <tag>_config2.json, _config5_1f.json secondary BASELINE configs (1 GPU)
#   <tag>_kernel_stats.txt               rocprofv3 --kernel-trace --stats of the same bench command
#   <tag>_dsweep.txt                     D sensitivity: --iteration 5 / 7 / 14
# usage: bash tools/collect_profiles.sh r02
R=$PWD; tag=${1:-r02}; out=$R/gpurun_out
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --config 2 --no-cpu-baseline > $out/${tag}_config2.json 2>/dev/null
python bench.py --config 5 --global-batch 1 --no-cpu-baseline --steps 100 > $out/${tag}_config5_1frame.json 2>/dev/null
python bench.py --stage 2 --no-cpu-baseline --steps 100 > $out/${tag}_stage2_smpl.json 2>/dev/null
for it in 5 7 14; do python bench.py --iteration $it --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels']['per_kernel']
print('iteration', d['config']['iteration'], 'D/frame', round(d['config']['mean_tile_pairs_per_frame']), 'it/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3),
      ' '.join('%s=%.1f' % (n, k[n]['us_per_iter']) for n in ('preprocess','tile_scan','scatter','tile_sort','render_fwd','render_bwd','preprocess_bwd')))
"; done > $out/${tag}_dsweep.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" > $out/${tag}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python bench.py --steps 25 --warmup 10 --no-cpu-baseline --no-kernel-events  (35 iterations incl. warm-up)")
print("%-78s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:45]:
    print("%-78s %8s %12.1f %10.2f %7s" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -2 /tmp/prof_$tag.log

#!/bin/bash
# The rasterizer against Gaussian size inside the training iteration (profiles/rNN_dsweep.txt): bench.py --iteration N scales
# every Gaussian by N / 1000 (the reference's scale warm-up, model/avatar_model.py:315-316); per-kernel HIP events of the
# instrumented warm-up steps, us per 2-frame iteration.   usage: tools/dsweep.sh > gpurun_out/dsweep.txt
cd "$(dirname "$0")/.."
for it in ${ITERS:-7 60 150 300}; do
  python bench.py --iteration $it --steps 20 --warmup 8 --no-cpu-baseline --no-secondary --no-fixed-batch --no-measure-traffic 2>/dev/null |
    python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    k=d['kernels']['per_kernel']
    ras={n:round(k[n]['us_per_iter']) for n in ('preprocess','tile_scan','scatter','tile_sort','render_fwd','render_bwd','preprocess_bwd') if n in k}
    print('iteration $it it/s %.1f pairs/frame timed %.0f, in the instrumented steps %.0f: %s' % (d['value'], d['config'].get('mean_tile_pairs_per_frame'), d['kernels'].get('probe_pairs_per_frame') or 0, ras))
"
done

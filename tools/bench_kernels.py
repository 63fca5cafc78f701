#!/usr/bin/env python
"""Print value and the per-kernel table (us per iteration, launches) of bench.py JSON lines: tools/bench_kernels.py a.json b.json"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e); continue
    k = d.get("kernels", {}).get("per_kernel", {})
    print(f"{f}: {d['value']:.1f} it/s  {d['ms_per_step']:.3f} ms  pairs/frame {d['config'].get('mean_tile_pairs_per_frame', 0):.0f}")
    print("   " + "  ".join(f"{n}={v['us_per_iter']:.0f}" for n, v in k.items()))
    r = d.get("roofline_raster_bwd")
    if r: print(f"   render_bwd {r['avg_us']:.1f} us  frac {r['frac']:.3f}")

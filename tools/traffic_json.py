#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.txt (tools/collect_r04.sh: per-kernel means of the FETCH_SIZE / WRITE_SIZE passes) ->
profiles/rNN_traffic.json, the per-family HBM-side traffic per launch that bench.py cites under roofline.traffic.
usage: tools/traffic_json.py <pmc_traffic.txt> <out.json> <command the passes ran>"""
import json, re, sys
FAMILIES = {   # kernel-name substring -> family key (first match wins; order matters)
    "layer_bwd_spec_kernel<false, true, 128>": "layer_bwd",
    "layer_bwd_spec_kernel<true, true, 128>": "layer_bwd_accumulating",
    "layer_fwd_spec_kernel": "layer_fwd", "head_bwd_kernel<true, 3>": "head_bwd",
    "render_bwd_kernel": "render_bwd", "render_fwd_kernel": "render_fwd", "preprocess_bwd_kernel": "preprocess_bwd",
    "preprocess_kernel": "preprocess", "scatter_kernel": "scatter", "tile_sort_chunk_kernel": "tile_sort",
    "ssim_fwd_kernel": "ssim_fwd", "ssim_bwd_kernel": "ssim_bwd", "skin_dmats_kernel": "skin_dmats",
    "conv5_kernel": "conv5_apply", "conv5_wgrad_kernel": "conv5_wgrad",
}
txt = open(sys.argv[1]).read()
out = {"_comment": "HBM-side traffic per launch from rocprofv3 PMC passes of `%s` (tools/collect_r04.sh: separate --pmc FETCH_SIZE "
                   "and --pmc WRITE_SIZE passes; values as reported = bytes/1024; mean over the second half of the launches; 2 frames "
                   "per launch). FETCH_SIZE under-reports 16-B/lane reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section): bench.py "
                   "uses 2*FETCH + WRITE." % (sys.argv[3] if len(sys.argv) > 3 else "python bench.py")}
cur = None
for block in txt.split("== pass ")[1:]:
    counter = block.split()[0]
    key = "fetch_kb" if counter.startswith("FETCH") else "write_kb"
    lines = block.splitlines()[1:]
    for name, val in zip(lines[0::2], lines[1::2]):
        m = re.search(r"mean\s+([0-9.]+)", val)
        if not m:
            continue
        for sub, fam in FAMILIES.items():
            if sub in name:
                out.setdefault(fam, {}).setdefault(key, float(m.group(1)))
                break
if "layer_fwd" in out:
    out["mlp_fwd"] = dict(out["layer_fwd"])
json.dump(out, open(sys.argv[2], "w"), indent=1)
print({k: v for k, v in out.items() if k != "_comment"})

#!/usr/bin/env python
"""Per-wave residency of ssim_fwd_kernel (needs a -DGANET_SSIM_TRACE build, GA_DEV=lib_dir=<dir>): start / end
(s_memrealtime, 100 MHz) and the XCD / SE / CU / SIMD every wave ran on."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussianavatar_amd import _native, fused
lib = _native.ganet()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
a = torch.rand(B, 3, 1024, 1024, device="cuda"); b = torch.rand_like(a)
for _ in range(3):
    s, l = fused.ssim_l1_mean(a, b)
torch.cuda.synchronize()
buf = np.zeros((8192, 4), dtype=np.uint64)
lib.ganet_dev_ssim_trace.argtypes = [ctypes.c_void_p]
assert lib.ganet_dev_ssim_trace(buf.ctypes.data) == 0
t = buf[buf[:, 1] > 0].astype(np.int64)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # us
hw, xcc = t[:, 2], t[:, 3] & 0xf
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print("waves %d  kernel span %.1f us  start: median %.1f p90 %.1f max %.1f us  lifetime: median %.1f p10 %.1f p90 %.1f us" % (
    len(t), en.max(), np.median(st), np.percentile(st, 90), st.max(), np.median(en - st), np.percentile(en - st, 10), np.percentile(en - st, 90)))
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
cus, cnt = np.unique(key, return_counts=True)
print("distinct CUs %d  waves per CU: min %d median %d max %d" % (len(cus), cnt.min(), np.median(cnt), cnt.max()))
sk = key * 4 + simd
sim, c2 = np.unique(sk, return_counts=True)
print("distinct SIMDs %d  waves per SIMD histogram:" % len(sim), dict(zip(*np.unique(c2, return_counts=True))))
# concurrency on the busiest SIMDs
for s_ in sim[np.argsort(-c2)][:3]:
    m = sk == s_
    print("  SIMD %d:" % s_, sorted((round(float(x), 1), round(float(y), 1)) for x, y in zip(st[m], en[m])))
print("per XCD: waves, first start, last end:", [(int(x), int((xcc == x).sum()), round(float(st[xcc == x].min()), 1), round(float(en[xcc == x].max()), 1)) for x in np.unique(xcc)])

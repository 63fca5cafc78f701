"""GPU parity tests, second line (round-2 review, "harden parity where it is loose"):

* per-ELEMENT gradient checks against the oracle's analytic backward — the first line's bar
  (max |diff| <= 2e-3 of the tensor's max) lets entries far below the maximum be arbitrarily wrong;
* edge cases of the per-Gaussian forward: near-plane cull at t.z = 0.2 to the ulp,
  the +-1.3 tanfov clamp, det == 0, w -> -1e-7, a radius that ends exactly on a tile border — HIP vs oracle,
  integer outputs bit-exact, driven by hypothesis;
* n_contrib exactly equal wherever the decision is not a float coin toss (the float32 and float64 oracles agree).
"""
import numpy as np
import pytest

from tests.scenes import cam_kwargs, camera, random_scene
from tests.test_raster_gpu import assert_forward_parity, oracle_forward

pytestmark = pytest.mark.gpu

ELEM_FLOOR = 1e-4          # entries above this fraction of the tensor's max are checked one by one
ELEM_REL_TOL = 1e-2
COS_TOL = 1e-6


def assert_gradient_elements(name, got, ref32, ref64):
    """got: HIP (float32, atomics); ref32 / ref64: the oracle's analytic backward in float32 / float64.
    Every entry above 1e-4 of the tensor's max must be within 1e-2 (relative) of the float64 value — unless the
    SEQUENTIAL float32 evaluation itself misses the float64 value by more than a quarter of that (the sum cancels or
    a pixel flipped a 1/255 / 1e-4 decision between the precisions: the entry is not defined to 1e-2 in float32).
    Those entries — at most 1 % of the tensor — are held to the first line's absolute bar (2e-3 of the tensor's max).
    At most 1e-4 of the well-conditioned entries may miss the 1e-2 bar (decision flips, see below).
    Cosine of the whole tensor against float64 >= 1 - 1e-6."""
    a, b, c = (x.astype(np.float64).ravel() for x in (got, ref32, ref64))
    big = np.abs(c) > ELEM_FLOOR * np.abs(c).max()
    den = np.abs(c)[big]
    rel = np.abs(a - c)[big] / den
    rel32 = np.abs(b - c)[big] / den
    soft = rel32 > 0.25 * ELEM_REL_TOL
    assert soft.mean() <= 1e-2, (name, "ill-conditioned entries", float(soft.mean()))
    bad = (rel > ELEM_REL_TOL) & ~soft
    # A handful of entries per 100,000 may still be off: the HIP forward evaluates transmittance as a prefix-product
    # tree, so a pixel can take a 1/255 / 1e-4 decision differently from BOTH oracles and move one Gaussian's gradient
    # by about a percent (measured: 3 of 183,264 at the headline size, 4 of 134,133 at 1080p). Bounded in number
    # and, like every entry, by the first line's absolute bar.
    assert bad.mean() <= 1e-4, (name, "entries off:", int(bad.sum()), "of", int(big.sum()), "worst rel", float(rel[~soft].max()))
    assert (np.abs(a - c)[big][bad] <= 2e-3 * np.abs(c).max()).all(), name
    # the ill-conditioned ones: a different summation order realises a different rounding error, so their bar is the
    # first line's absolute one
    lim = 2e-3 * np.abs(c).max() + 8.0 * np.abs(b - c)[big][soft]       # ... or a few times the float32 oracle's own miss
    assert (np.abs(a - c)[big][soft] <= lim).all(), (name, "ill-conditioned entries beyond the absolute bar")
    cos = float(a @ c / (np.linalg.norm(a) * np.linalg.norm(c) + 1e-300))
    assert cos >= 1.0 - COS_TOL, (name, cos)


@pytest.mark.parametrize("kind", ["general", "avatar"])
@pytest.mark.parametrize("P,W,H,scale", [(300, 48, 32, 0.05), (3000, 128, 128, 0.03), (4000, 200, 120, 0.02)])
def test_backward_parity_per_element(raster_oracle, raster_oracle_f64, kind, P, W, H, scale):
    from tests.hip_helpers import hip_forward_backward
    sc = random_scene(P, W, H, seed=7 + P, kind=kind, scale_med=scale)
    g = np.random.default_rng(3).normal(0, 1, (3, H, W)).astype(np.float32)
    ref = oracle_forward(raster_oracle, sc)
    rb = raster_oracle.backward(ref, g)
    ref64 = oracle_forward(raster_oracle_f64, sc)
    rb64 = raster_oracle_f64.backward(ref64, g)
    _c, _r, grads = hip_forward_backward(sc, g)
    keys = ("dmeans3D", "dcolors", "dopacity", "dscales", "drots") if kind == "general" else ("dmeans3D", "dcolors", "dscales")
    for k in keys:
        assert_gradient_elements(k, grads[k], rb[k], rb64[k])
    assert_gradient_elements("dmeans2D", grads["dmeans2D"][:, :2], rb["dmeans2D"][:, :2], rb64["dmeans2D"][:, :2])


def test_backward_parity_per_element_headline_size(raster_oracle, raster_oracle_f64):
    """200k avatar-like Gaussians at 1024 x 1024 (BASELINE.json configs[2])."""
    from tests.hip_helpers import hip_forward_backward
    P, W, H = 200_000, 1024, 1024
    sc = random_scene(P, W, H, seed=1, kind="avatar", spread=0.45, scale_med=0.0035)
    g = np.random.default_rng(5).normal(0, 1, (3, H, W)).astype(np.float32)
    ref = oracle_forward(raster_oracle, sc)
    rb = raster_oracle.backward(ref, g)
    rb64 = raster_oracle_f64.backward(oracle_forward(raster_oracle_f64, sc), g)
    _c, _r, grads = hip_forward_backward(sc, g)
    for k in ("dmeans3D", "dcolors", "dscales"):
        assert_gradient_elements(k, grads[k], rb[k], rb64[k])


def test_backward_parity_per_element_1080p(raster_oracle, raster_oracle_f64):
    """300k general Gaussians at 1920 x 1080 (BASELINE.json configs[4]'s image size)."""
    from tests.hip_helpers import hip_forward_backward
    P, W, H = 300_000, 1920, 1080
    sc = random_scene(P, W, H, seed=21, kind="general", spread=1.2, scale_med=0.006)
    g = np.random.default_rng(8).normal(0, 1, (3, H, W)).astype(np.float32)
    ref = oracle_forward(raster_oracle, sc)
    rb = raster_oracle.backward(ref, g)
    rb64 = raster_oracle_f64.backward(oracle_forward(raster_oracle_f64, sc), g)
    _c, _r, grads = hip_forward_backward(sc, g)
    for k in ("dmeans3D", "dcolors", "dscales", "dopacity", "drots"):
        assert_gradient_elements(k, grads[k], rb[k], rb64[k])


@pytest.mark.parametrize("P,W,H,kind,scale", [(3000, 128, 128, "avatar", 0.03), (5000, 256, 256, "general", 0.02),
                                              (200_000, 1024, 1024, "avatar", 0.0035)])
def test_n_contrib_exact_where_decisions_are_stable(raster_oracle, raster_oracle_f64, P, W, H, kind, scale):
    """n_contrib (index of the last blended list entry) must be EXACTLY the oracle's on every pixel whose value does
    not hinge on a float coin toss — taken as: the float32 and the float64 oracle agree on it."""
    from tests.hip_helpers import hip_forward_state
    sc = random_scene(P, W, H, seed=1 if P == 200_000 else P, kind=kind, scale_med=scale,
                      **({"spread": 0.45} if P == 200_000 else {}))
    ref = oracle_forward(raster_oracle, sc)
    ref64 = oracle_forward(raster_oracle_f64, sc)
    got = hip_forward_state(sc)
    n32, n64, nh = (np.asarray(x["n_contrib"]).astype(np.int64).reshape(H, W) for x in (ref, ref64, got))
    stable = n32 == n64
    assert stable.mean() > 0.99
    assert np.array_equal(nh[stable], n32[stable]), int((nh[stable] != n32[stable]).sum())


# ---------------------------------------------------------------------------------------------- edge cases
def _edge_scene(W=96, H=64):
    """A small avatar-like scene to plant edge-case Gaussians into."""
    return random_scene(64, W, H, seed=17, kind="general", scale_med=0.05)


def _view_z(sc, p):
    V = np.asarray(sc["viewmatrix"], np.float32)              # row-vector convention: t = [p, 1] @ V
    return np.float32(np.float32(np.float32(p[0] * V[0, 2]) + np.float32(p[1] * V[1, 2])) + np.float32(p[2] * V[2, 2])) + V[3, 2]


def test_near_plane_cull_at_float_granularity(raster_oracle):
    """t.z <= 0.2 culls: Gaussians whose view depth straddles 0.2 in the smallest steps the
    float32 transform can take there (the sum's terms are ~2, so its results near 0.2 lie 16 ulps of 0.2 apart) must
    fall on the same side in the HIP preprocess as in the oracle (same operation order, no FMA contraction)."""
    sc = _edge_scene()
    V = np.asarray(sc["viewmatrix"], np.float64)
    # the world point on the optical axis at view depth 0.2, then its dominant world coordinate
    # stepped ulp by ulp: the float32 view depth crosses 0.2 somewhere inside the sweep
    Vinv = np.linalg.inv(V)
    p0 = (np.array([0.0, 0.0, 0.2, 1.0]) @ Vinv)[:3].astype(np.float32)        # on the optical axis
    ax = int(np.argmax(np.abs(V[:3, 2])))
    n = 61
    pts = np.repeat(p0[None], n, 0)
    for j, k in enumerate(range(-30, 31)):
        v = pts[j, ax]
        for _ in range(abs(k)):
            v = np.nextafter(v, np.float32(np.inf if k > 0 else -np.inf))
        pts[j, ax] = v
    sc = dict(sc, means3D=np.concatenate([pts, sc["means3D"]]), colors=np.concatenate([sc["colors"][:1].repeat(n, 0), sc["colors"]]),
              opacities=np.concatenate([np.full(n, 0.7, np.float32), sc["opacities"]]),
              scales=np.concatenate([np.full((n, 3), 0.002, np.float32), sc["scales"]]),
              rotations=np.concatenate([sc["rotations"][:1].repeat(n, 0), sc["rotations"]]), P=sc["P"] + n)
    ref, got = assert_forward_parity(raster_oracle, sc)
    # both sides of the plane are really present among the planted points
    assert (ref["radii"][:n] > 0).any() and (ref["radii"][:n] == 0).any(), ref["radii"][:n]


def test_frustum_clamp_and_singular_covariance(raster_oracle):
    """|t.x / t.z| beyond 1.3 tanfov (the Jacobian is evaluated at the clamped position, A.1 step 4), Gaussians far
    outside the image whose rectangle still reaches it, a zero scale (Sigma2D = 0.3 I: det = 0.09, the dilation keeps
    it regular) and a huge anisotropy (det cancels, step 5). (Infinite / NaN covariances are left out: their
    float -> int conversions are undefined behaviour in the reference as well.)"""
    from tests.hip_helpers import hip_forward_backward
    sc = _edge_scene()
    V = np.asarray(sc["viewmatrix"], np.float64)
    Vinv = np.linalg.inv(V)
    def world(tx, ty, tz):           # view-space point -> world (row-vector convention)
        return (np.array([tx, ty, tz, 1.0]) @ Vinv)[:3]
    tz = 2.0
    lim = 1.3 * sc["tanfovx"] * tz
    planted = [world(s * f * lim, 0.1, tz) for s in (-1, 1) for f in (0.98, 1.0, 1.02, 1.5, 3.0)]
    n = len(planted)
    sc["means3D"][:n] = np.asarray(planted, np.float32)
    sc["scales"][:n] = 0.4                                  # big enough to reach the image from outside
    sc["scales"][n] = 0.0                                   # degenerate: only the 0.3 dilation is left
    # a needle along the screen diagonal: Sigma2D ~ lambda [[1, 1], [1, 1]] / 2, a c - b^2 cancels in float32
    # (to 0 -> invisible by A.1 step 5, or to rounding noise — the same noise on both sides, same operation order)
    sc["scales"][n + 1] = (1e4, 1e-3, 1e-3)
    sc["rotations"][n + 1] = (np.cos(np.pi / 8), 0.0, 0.0, np.sin(np.pi / 8))
    sc["scales"][n + 2] = (3e3, 3e3, 1e-6)                  # a disc larger than the frustum
    ref, got = assert_forward_parity(raster_oracle, sc)
    g = np.random.default_rng(2).normal(0, 1, (3, sc["H"], sc["W"])).astype(np.float32)
    rb = raster_oracle.backward(ref, g)
    _c, radii, grads = hip_forward_backward(sc, g)
    np.testing.assert_array_equal(radii, ref["radii"])
    # the needle's conic is rounding noise of a cancelled determinant (values ~1e-4 from operands ~1e8): its own
    # gradient is compared for finiteness only, everybody else's (the needle composites over them) to the usual bar
    ok = np.ones(sc["P"], bool)
    ok[n + 1] = False
    for k in ("dmeans3D", "dscales", "dcolors"):
        assert np.isfinite(grads[k]).all(), k
        a, b = grads[k][ok], rb[k][ok]
        assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max() + 1e-12, k
    # the clamped ones: x-gradient of the Jacobian term is zeroed (A.5c) — covered by the comparison above; make
    # sure the clamp was active on the planted set
    t = np.c_[sc["means3D"][:n], np.ones(n)] @ V
    assert (np.abs(t[:, 0] / t[:, 2]) > 1.3 * sc["tanfovx"]).sum() >= 4


def test_w_near_zero_and_behind_camera_mix(raster_oracle):
    """h.w + 1e-7 in the perspective divide (A.1 step 2): points with clip-space w around -1e-7 .. 1e-7 sit at the
    camera plane and are culled by the near test; the divide must not produce a different rect for their visible
    neighbours. Mixed with points behind the camera."""
    sc = _edge_scene()
    V = np.asarray(sc["viewmatrix"], np.float64)
    Vinv = np.linalg.inv(V)
    pts = [(np.array([0.01 * i, 0.02, z, 1.0]) @ Vinv)[:3] for i, z in enumerate((-1e-7, -1e-8, 0.0, 1e-8, 1e-7, 0.1999999, 0.2000001, -3.0))]
    sc["means3D"][:len(pts)] = np.asarray(pts, np.float32)
    assert_forward_parity(raster_oracle, sc)


def test_radius_ending_on_a_tile_border(raster_oracle):
    """p.x + radius + 15 and p.x - radius landing exactly on multiples of 16 (A.1 step 8: C truncation decides
    whether the neighbouring tile column is touched)."""
    W, H = 128, 96
    sc = random_scene(32, W, H, seed=3, kind="avatar", scale_med=0.01)
    cam = camera(W, H)
    V = np.asarray(sc["viewmatrix"], np.float64)
    Vinv = np.linalg.inv(V)
    fx = W / (2 * sc["tanfovx"])
    tz = 2.0
    pts, scl = [], []
    for px in (15.5, 16.0, 31.5, 47.5, 48.0, 63.5, 64.0, 79.5):
        for r_target in (4.0, 8.0, 16.0):
            tx = (px - (W - 1) / 2.0) * tz / fx               # pixel centre px  <->  ndc  <->  view x
            pts.append((np.array([tx, 0.0, tz, 1.0]) @ Vinv)[:3])
            # ceil(3 sigma) = r_target for sigma just below r_target / 3 (sigma^2 = s^2 fx^2 / tz^2 + 0.3)
            sig2 = (r_target / 3.0) ** 2 * 0.999
            scl.append(np.sqrt(max(sig2 - 0.3, 1e-8)) * tz / fx)
    n = min(len(pts), 32)
    sc["means3D"][:n] = np.asarray(pts[:n], np.float32)
    sc["scales"][:n] = np.asarray(scl[:n], np.float32)[:, None]
    ref, got = assert_forward_parity(raster_oracle, sc)
    assert len(np.unique(ref["rect"][:n], axis=0)) > 6


try:
    from hypothesis import given, settings, strategies as st
    HAVE_HYPOTHESIS = True
except ImportError:                                          # pragma: no cover
    HAVE_HYPOTHESIS = False


@pytest.mark.skipif(not HAVE_HYPOTHESIS, reason="hypothesis not installed")
def test_random_edge_mixtures_hypothesis(raster_oracle):
    """Property test: for random small scenes laced with extreme values — depths at the near
    plane, positions at the clamp, tiny / huge / zero scales, un-normalised and zero quaternions, opacities at 0,
    1/255 and 1 — every integer output of the HIP forward equals the oracle's and the image stays within tolerance."""
    specials_z = [0.2, np.nextafter(np.float32(0.2), np.float32(1)), np.nextafter(np.float32(0.2), np.float32(0)), 0.0, -1.0, 50.0]
    specials_s = [0.0, 1e-12, 1e-4, 0.3, 5.0, 1e3]
    specials_o = [0.0, 1.0 / 255.0, np.nextafter(np.float32(1.0 / 255.0), np.float32(0)), 0.5, 1.0]

    @settings(max_examples=25, deadline=None, derandomize=True)
    @given(seed=st.integers(0, 10_000), nz=st.integers(0, 8), ns=st.integers(0, 8), no=st.integers(0, 8),
           zero_quat=st.booleans(), W=st.sampled_from([33, 48, 97]), H=st.sampled_from([17, 32, 64]))
    def run(seed, nz, ns, no, zero_quat, W, H):
        rng = np.random.default_rng(seed)
        sc = random_scene(48, W, H, seed=seed, kind="general", scale_med=0.05)
        V = np.asarray(sc["viewmatrix"], np.float64)
        Vinv = np.linalg.inv(V)
        for i in range(nz):
            z = float(rng.choice(specials_z))
            x = float(rng.choice([-1.5, -1.3, 0.0, 1.3, 1.5])) * sc["tanfovx"] * max(z, 0.2)
            sc["means3D"][i] = (np.array([x, 0.05 * i, z, 1.0]) @ Vinv)[:3].astype(np.float32)
        for i in range(ns):
            sc["scales"][8 + i, rng.integers(0, 3)] = np.float32(rng.choice(specials_s))
        for i in range(no):
            sc["opacities"][16 + i] = np.float32(rng.choice(specials_o))
        if zero_quat:
            sc["rotations"][24] = 0.0
            sc["rotations"][25] *= 50.0
        assert_forward_parity(raster_oracle, sc)

    run()


@pytest.mark.parametrize("P,W,H,kind,scale", [(3000, 128, 128, "avatar", 0.03), (20000, 256, 256, "general", 0.01),
                                              (200_000, 1024, 1024, "avatar", 0.0035)])
def test_debug_backward_is_bitwise_repeatable(raster_oracle, P, W, H, kind, scale):
    """settings.debug (the reference's debug knob, /root/reference/gaussian_renderer/__init__.py:33) selects the
    deterministic backward: one wave per tile in fixed segment order, per-pair records, per-Gaussian gather in tile
    order — no float atomics. Two runs must agree bit for bit (the default path's atomics do not), and the result
    must meet the same parity bar as the default path."""
    import torch
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    from tests.hip_helpers import scene_tensors, settings_from_scene
    sc = random_scene(P, W, H, seed=1 if P == 200_000 else P, kind=kind, scale_med=scale,
                      **({"spread": 0.45} if P == 200_000 else {}))
    g = torch.tensor(np.random.default_rng(4).normal(0, 1, (3, H, W)).astype(np.float32), device="cuda")

    def run(debug):
        rs = settings_from_scene(sc, debug=debug)
        t = scene_tensors(sc, requires_grad=True)
        color, radii = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                              colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
        color.backward(g)
        return color.detach(), [t[k].grad.clone() for k in ("means3D", "colors", "opacities", "scales", "rotations")]

    img_a, ga = run(True)
    img_b, gb = run(True)
    assert torch.equal(img_a, img_b)
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
    img_c, gc = run(False)
    assert torch.equal(img_a, img_c)                     # the forward pass is the same kernel either way
    ref = oracle_forward(raster_oracle, sc)
    rb = raster_oracle.backward(ref, g.cpu().numpy())
    for got, fast, key in zip(ga, gc, ("dmeans3D", "dcolors", "dopacity", "dscales", "drots")):
        want = rb[key].reshape(got.shape)
        scale_ = np.abs(want).max() + 1e-12
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-3 * scale_, key
        assert float((got - fast).abs().max()) <= 2e-3 * scale_, key


@pytest.mark.parametrize("spread", [0.06, 0.025])
def test_more_than_8192_tiles_long_lists_and_debug_backward(raster_oracle, spread):
    """Above 8192 tiles (here 2048 x 1152: 9216) tile_scan leaves the identity tile order, so every walk over it must
    SKIP empty tiles instead of stopping at the first one (the deterministic backward stopped, its
    per-pair records stayed unwritten and the gather summed uninitialised memory). The scene is one dense cluster:
    spread 0.06 gives lists above 2048 and 4096 keys (chunk sort + both merge passes in their unordered form),
    spread 0.025 lists above 8192 (the one-workgroup path). Checked: bit-exact lists, the default backward and the
    deterministic one against the oracle, the deterministic one bitwise repeatable."""
    import torch
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    from tests.hip_helpers import scene_tensors, settings_from_scene
    W, H = 2048, 1152
    assert ((W + 15) // 16) * ((H + 15) // 16) > 8192
    sc = random_scene(60_000, W, H, seed=31, kind="avatar", spread=spread, scale_med=0.004)
    ref, got = assert_forward_parity(raster_oracle, sc)
    assert got["status"][3] > (8192 if spread < 0.05 else 4096), got["status"]
    g = torch.tensor(np.random.default_rng(6).normal(0, 1, (3, H, W)).astype(np.float32), device="cuda")
    rb = raster_oracle.backward(ref, g.cpu().numpy())

    def run(debug):
        rs = settings_from_scene(sc, debug=debug)
        t = scene_tensors(sc, requires_grad=True)
        color, radii = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                              colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
        color.backward(g)
        return [t[k].grad.clone() for k in ("means3D", "colors", "opacities", "scales", "rotations")]

    ga, gb, gc = run(True), run(True), run(False)
    for a, b, c, key in zip(ga, gb, gc, ("dmeans3D", "dcolors", "dopacity", "dscales", "drots")):
        assert torch.equal(a, b), key
        want = rb[key].reshape(a.shape)
        bar = 2e-3 * (np.abs(want).max() + 1e-12)
        assert np.abs(a.cpu().numpy() - want).max() <= bar, ("debug", key)
        assert np.abs(c.cpu().numpy() - want).max() <= bar, ("default", key)


def _scene_with_tile_lists(W, H, tile_sizes, seed=17):
    """A scene whose tile `t` (row-major index on the 16-pixel grid) holds exactly `tile_sizes[t]` Gaussians: tiny ones placed
    inside the tile (radius 2 px, centres 5.5 .. 10.5 px from the tile's corner: each touches exactly its tile), depths random
    with some exact ties."""
    gx = (W + 15) // 16
    sc = random_scene(4, W, H, seed=3, kind="avatar", scale_med=1e-4)
    rng = np.random.default_rng(seed)

    # the pixel of a world point (row-vector convention of the scene's matrices); at fixed z the map is nearly affine in (x, y)
    def pixels(pts):
        hom = np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1) @ sc["projmatrix"].astype(np.float32)
        return np.stack([((hom[:, 0] / hom[:, 3] + 1.0) * W - 1.0) * 0.5, ((hom[:, 1] / hom[:, 3] + 1.0) * H - 1.0) * 0.5], 1)
    probe = pixels(np.array([[0, 0.3, 0], [1, 0.3, 0], [0, 1.3, 0]], np.float32)).astype(np.float64)
    Ainv = np.linalg.inv(np.stack([probe[1] - probe[0], probe[2] - probe[0]], 1))      # d (x, y) / d pixel
    tiles = np.concatenate([np.full(n, t, np.int64) for t, n in tile_sizes.items()])
    P = len(tiles)
    want = np.stack([(tiles % gx) * 16 + rng.uniform(5.5, 10.5, P), (tiles // gx) * 16 + rng.uniform(5.5, 10.5, P)], 1)
    xy = (want - probe[0]) @ Ainv.T + np.array([0.0, 0.3])
    z = rng.uniform(-0.02, 0.02, (P, 1))
    for _ in range(4):                                                         # (the map is only nearly affine: refine)
        xy = xy + (want - pixels(np.concatenate([xy, z], 1).astype(np.float32))) @ Ainv.T
    means = np.concatenate([xy, z], 1).astype(np.float32)
    pix = pixels(means)
    ok = ((pix[:, 0] % 16 > 4) & (pix[:, 0] % 16 < 12) & (pix[:, 1] % 16 > 4) & (pix[:, 1] % 16 < 12) &
          (pix[:, 0] // 16 == tiles % gx) & (pix[:, 1] // 16 == tiles // gx))
    assert ok.all(), int((~ok).sum())
    means = means[rng.permutation(P)]
    means[1::97, 2] = means[0:len(means[1::97]) * 97:97, 2][:len(means[1::97])]     # some exact depth ties (same z, other xy)
    sc.update(P=P, means3D=means, colors=rng.uniform(0, 1, (P, 3)).astype(np.float32),
              opacities=rng.uniform(0.01, 0.05, P).astype(np.float32), scales=np.full((P, 3), 1e-4, np.float32),
              rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)))
    return sc


def test_list_lengths_on_every_sort_boundary(raster_oracle):
    """Tile lists of exactly 2047 / 2048 / 2049 / 4095 / 4096 / 4097 / 6144 / 6145 / 8191 / 8192 / 8193 / 10000 entries: one
    chunk, two .. four chunks with full and one-key last chunks (the one-launch merge, gsr_binning.hip), and the long-list path —
    every length at which the sort changes its decomposition. Lists, ranges and image against the oracle as everywhere."""
    sizes = [2047, 2048, 2049, 4095, 4096, 4097, 6144, 6145, 8191, 8192, 8193, 10000]
    sc = _scene_with_tile_lists(128, 128, dict(zip([9, 11, 13, 18, 20, 22, 25, 27, 29, 34, 36, 38], sizes)))
    ref, got = assert_forward_parity(raster_oracle, sc)
    lens = np.sort(ref["ranges"][:, 1].astype(np.int64) - ref["ranges"][:, 0].astype(np.int64))[-len(sizes):]
    assert lens.tolist() == sorted(sizes), lens


def test_a_list_of_exactly_one_chunk_does_not_end_the_merge_walk(raster_oracle):
    """The sort and merge grids walk the tiles in order of size CLASS (1 + floor(log2 n)) and stop at the first list that needs
    no work. A list of exactly 2048 keys needs no merge but shares its class with lists of 2049 .. 4095 keys that do: a
    workgroup that met it first used to end its walk and leave its later lists unmerged (found in round 4 with a 1024-key
    chunk build: a memory fault in render_fwd; with 2048-key chunks it takes more than 256 multi-chunk lists). Here: 700
    lists of 2049 .. 2060 keys and 40 of exactly 2048, i.e. every merge workgroup walks three ranks of that class."""
    W = H = 512
    rng = np.random.default_rng(5)
    interior = [ty * 32 + tx for ty in range(1, 31) for tx in range(1, 31)]          # 900 tiles away from the border
    chosen = rng.permutation(interior)[:740]
    tile_sizes = {int(t): (2048 if i < 40 else int(rng.integers(2049, 2061))) for i, t in enumerate(chosen)}
    sc = _scene_with_tile_lists(W, H, tile_sizes, seed=23)
    assert_forward_parity(raster_oracle, sc)


@pytest.mark.parametrize("seed,P,W,H,spread,scale", [(2, 2_000_000, 512, 384, 0.6, 0.0015), (4, 1_500_000, 512, 512, 0.25, 0.001)])
def test_dense_scenes_with_hundreds_of_multi_chunk_lists(raster_oracle, seed, P, W, H, spread, scale):
    """Millions of small Gaussians on a small image: 130-280 tile lists above one sort chunk (more than the merge launch has
    workgroups), 80-110 above 8192 keys (the whole-list path), the longest 15 k / 61 k keys — the decompositions of the tile
    sort all at once, lists bit-exact against the oracle."""
    sc = random_scene(P, W, H, seed=seed, kind="avatar", spread=spread, scale_med=scale)
    ref, got = assert_forward_parity(raster_oracle, sc)
    ln = ref["ranges"][:, 1].astype(np.int64) - ref["ranges"][:, 0].astype(np.int64)
    assert (ln > 2048).sum() > 100 and (ln > 8192).sum() > 50


def test_list_lengths_around_the_bucket_sort(raster_oracle):
    """Round 6: lists beyond 8192 keys are sorted as independent ~4096-key output buckets (gsr_binning.hip: psrs_bucket — regular
    samples of the list's sorted 2048-key chunks, splitters, per-chunk searches, multi-run LDS merge), lists beyond 64 chunks
    (131072 keys) keep the one-workgroup merge through HBM. Every length at which that path changes its decomposition — the
    number of chunks (5, 6, 8, 9, 20, 32, 33, 63, 64), the sample spacing (512 .. 32), a one-key last chunk, the hand-over
    to the fallback at 131073 — with lists, ranges and image against the oracle as everywhere."""
    sizes = [8193, 10241, 12288, 16384, 16385, 40000, 65536, 65537, 129000, 131072, 131073, 150000]
    sc = _scene_with_tile_lists(128, 128, dict(zip([9, 11, 13, 18, 20, 22, 25, 27, 29, 34, 36, 38], sizes)), seed=29)
    ref, got = assert_forward_parity(raster_oracle, sc)
    lens = np.sort(ref["ranges"][:, 1].astype(np.int64) - ref["ranges"][:, 0].astype(np.int64))[-len(sizes):]
    assert lens.tolist() == sorted(sizes), lens
    assert got["status"][5] == sum((n + 4095) // 4096 for n in sizes if 8192 < n <= 131072)     # the work list's items


@pytest.mark.parametrize("P,W,H,scale,frac", [(3000, 128, 128, 0.03, 0.5), (20000, 256, 256, 0.08, 0.05),
                                              (200000, 512, 512, 0.1, 0.01), (400000, 256, 256, 0.05, 0.03)])
def test_heavy_overflow_leaves_valid_sorted_lists(raster_oracle, P, W, H, scale, frac):
    """A forward pass whose pair buffer holds only a fraction of the pairs (a stale capacity history when the scene suddenly
    grows: the first iterations of a from-scratch training after a smaller run, found in round 6 as a device fault in
    render_fwd) must still leave every list it kept sorted and made of valid indices: the tiles in front of the cut-off
    bit-identical to the oracle's lists, the tile the cut-off falls in a sorted subset of the oracle's list, every tile behind
    it empty for the later kernels — and forward + backward must run through (zero gradients, flag raised)."""
    import torch
    from gaussianavatar_amd import rasterizer as R
    from gaussianavatar_amd.rasterizer import GaussianRasterizer
    from tests.hip_helpers import hip_forward_state, scene_tensors, settings_from_scene
    sc = random_scene(P, W, H, seed=5, kind="avatar", scale_med=scale)
    ref = oracle_forward(raster_oracle, sc)
    cap = int(ref["D"] * frac)
    got = hip_forward_state(sc, max_pairs=cap)
    assert got["status"][1] == 1 and got["status"][0] == ref["D"]
    off = got["tile_offset"].astype(np.int64)
    np.testing.assert_array_equal(off[:-1].astype(np.uint32), ref["ranges"][:, 0])
    depth_bits = got["depth"].view(np.uint32).astype(np.uint64)
    whole = 0
    for t in range(len(off) - 1):
        s, e = min(off[t], cap), min(off[t + 1], cap)
        if e <= s:
            continue
        mine = got["point_list"][s:e].astype(np.int64)
        assert mine.min() >= 0 and mine.max() < P
        full = ref["point_list"][off[t]:off[t + 1]].astype(np.int64)
        if off[t + 1] <= cap:
            np.testing.assert_array_equal(mine, full)
            whole += 1
        else:                                                # the cut-off tile: some subset of its pairs, in key order
            keys = (depth_bits[mine] << np.uint64(32)) | mine.astype(np.uint64)
            assert (np.diff(keys.astype(np.float64)) >= 0).all() and len(np.unique(mine)) == len(mine)
            assert np.isin(mine, full).all()
    assert whole >= 1
    # the steady-state path (history says the buffer is large enough): forward + backward without a host check
    key = (P, W, H)
    saved = (R._capacity.pairs_per_gaussian, R._capacity.floor, dict(R._capacity.seen))
    try:
        R._capacity.pairs_per_gaussian, R._capacity.floor = 0, max(64, cap // 2)
        R._capacity.seen[key] = 10
        R._capacity.stamp[key] = __import__("time").monotonic()
        rs = settings_from_scene(sc)
        t = scene_tensors(sc, requires_grad=True)
        with pytest.warns(UserWarning, match="pair buffer overflow"):     # (raised by whichever poll sees it first)
            color, _ = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                              colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
            color.sum().backward()
            torch.cuda.synchronize()
            assert float(t["means3D"].grad.abs().max()) == 0.0 and int(R.overflow_flag("cuda")) == 1
            R.check_overflow(block=True)
    finally:
        R._capacity.pairs_per_gaussian, R._capacity.floor = saved[0], saved[1]
        R._capacity.seen = saved[2]
        R._capacity.pending.clear()
        R.clear_overflow_flag()

"""GPU parity of the fused net/loss kernels (include/ganet.h) against plain torch fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.grad_check import assert_grads_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(262144, 128, 128), (50000, 128, 66), (33333, 128, 194), (70001, 3, 128),
                                   (4096, 1, 128), (1000, 16, 18), (777, 32, 16)])
def test_linear_wgrad_matches_torch(M, N, K):
    from gaussianavatar_amd import fused
    torch.manual_seed(M % 97)
    x = torch.randn(M, K, device="cuda")
    w = (torch.randn(N, K, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(N, device="cuda").requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    g = torch.randn(M, N, device="cuda")
    y = fused.linear(xg, w, b)
    y.backward(g)
    ref_y = F.linear(x.double(), w.detach().double(), b.detach().double())
    assert float((y.double() - ref_y).abs().max()) < 1e-3
    dW = g.double().t() @ x.double()
    db = g.double().sum(0)
    dx = g.double() @ w.detach().double()
    scale = float(dW.abs().max())
    assert float((w.grad.double() - dW).abs().max()) <= 2e-5 * scale + 1e-4, (float((w.grad.double() - dW).abs().max()), scale)
    assert float((b.grad.double() - db).abs().max()) <= 2e-5 * float(db.abs().max()) + 1e-4
    assert float((xg.grad.double() - dx).abs().max()) < 1e-3


def test_wgrad_strided_input_and_unsupported_shape_fallback():
    from gaussianavatar_amd import fused
    M = 10000
    big = torch.randn(M, 200, device="cuda")
    x = big[:, :66]                                   # row stride 200
    w = torch.randn(128, 66, device="cuda", requires_grad=True)
    g = torch.randn(M, 128, device="cuda")
    fused.linear(x, w, None).backward(g)
    ref = g.double().t() @ x.double()
    assert float((w.grad.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4
    w2 = torch.randn(300, 66, device="cuda", requires_grad=True)      # N > 128 -> vendor GEMM path
    g2 = torch.randn(M, 300, device="cuda")
    fused.linear(x, w2, None).backward(g2)
    ref2 = g2.double().t() @ x.double()
    assert float((w2.grad.double() - ref2).abs().max()) <= 1e-4 * float(ref2.abs().max())


@pytest.mark.parametrize("M,C", [(262144, 128), (5000, 32), (1001, 16), (64, 256)])
@pytest.mark.parametrize("act", ["softplus", "identity"])
def test_batchnorm_act_matches_torch(M, C, act):
    from gaussianavatar_amd import fused
    torch.manual_seed(C)
    x = (torch.randn(M, C, device="cuda") * 3 + torch.linspace(-5, 25, C, device="cuda")).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 2.0)
        bn.bias.uniform_(-1, 1)
    ref_bn = torch.nn.BatchNorm1d(C).cuda().train()
    ref_bn.load_state_dict(bn.state_dict())
    g = torch.randn(M, C, device="cuda")
    y = fused.batchnorm_act(x, bn, act)
    y.backward(g)
    xr = x.detach().clone().requires_grad_(True)
    yr = ref_bn(xr)
    yr = F.softplus(yr) if act == "softplus" else yr
    yr.backward(g)
    torch.testing.assert_close(y, yr, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-3, atol=2e-5)
    gs = float(ref_bn.weight.grad.abs().max())
    assert float((bn.weight.grad - ref_bn.weight.grad).abs().max()) <= 2e-4 * gs + 1e-3
    assert float((bn.bias.grad - ref_bn.bias.grad).abs().max()) <= 2e-4 * float(ref_bn.bias.grad.abs().max()) + 1e-3
    torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn.running_var, ref_bn.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 37, 53), (2, 3, 256, 256)])
def test_fused_ssim_matches_reference_formulation(shape):
    """losses.ssim on the GPU (fused kernel) vs the reference's five grouped convolutions
    (utils/loss_utils.py:23-53) evaluated by torch on the CPU."""
    from gaussianavatar_amd.losses import ssim
    torch.manual_seed(1)
    a = torch.rand(*shape)
    b = (a + 0.15 * torch.randn(*shape)).clamp(0, 1)
    ac = a.clone().requires_grad_(True)
    s_ref = ssim(ac, b)                     # CPU tensors -> torch formulation
    s_ref.backward()
    ag = a.cuda().requires_grad_(True)
    s = ssim(ag, b.cuda())
    s.backward()
    assert abs(float(s) - float(s_ref)) < 2e-6
    scale = float(ac.grad.abs().max())
    assert float((ag.grad.cpu() - ac.grad).abs().max()) <= 2e-4 * scale


@pytest.mark.parametrize("feat,S", [(16, 32), (32, 96), (128, 512)])
def test_separable_bilinear_matmul_equals_grid_sample(feat, S):
    """The two-GEMM up-sampling used on the GPU vs F.grid_sample at the reference's query grid."""
    from gaussianavatar_amd.network import POP_no_unet, uv_to_grid
    torch.manual_seed(3)
    net = POP_no_unet(c_geom=8, hsize=16).cuda()
    idx = torch.stack(torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij"), -1).reshape(-1, 2).float() / (S - 1)
    uv = idx[None].cuda()
    mats = net._separable_bilinear(uv, feat, S)
    assert mats is not None
    pix = torch.randn(1, 8, feat, feat, device="cuda")
    ref = F.grid_sample(pix, uv_to_grid(uv, S), mode="bilinear", align_corners=False)
    ref = ref.reshape(1, 8, S * S).transpose(1, 2)
    Wr, Wc = mats
    t1 = torch.matmul(Wc, pix.permute(0, 2, 3, 1))
    pts = torch.matmul(Wr, t1.reshape(1, feat, S * 8)).reshape(1, S * S, 8)
    torch.testing.assert_close(pts, ref, rtol=1e-4, atol=1e-5)
    # a non-separable query set must fall back to grid_sample
    uv2 = uv.clone()
    uv2[0, 5, 0] += 0.01
    assert net._separable_bilinear(uv2, feat, S) is None


def test_whole_net_gpu_equals_cpu():
    import copy
    from gaussianavatar_amd.network import POP_no_unet
    torch.manual_seed(0)
    net = POP_no_unet(c_geom=8, hsize=16).train()
    geo = torch.randn(1, 8, 16, 16, requires_grad=True)
    S = 32
    idx = torch.stack(torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij"), -1).reshape(-1, 2).float() / (S - 1)
    net_g = copy.deepcopy(net).cuda()
    geo_g = geo.detach().cuda().requires_grad_(True)
    outs = net.forward_points(None, geo.expand(2, -1, -1, -1), idx[None].expand(2, -1, -1))
    outs_g = net_g.forward_points(None, geo_g.expand(2, -1, -1, -1), idx.cuda()[None].expand(2, -1, -1))
    w = [torch.randn_like(o) for o in outs]
    sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
    sum((o * wi.cuda()).sum() for o, wi in zip(outs_g, w)).backward()
    for a, b in zip(outs, outs_g):
        torch.testing.assert_close(a, b.cpu(), rtol=1e-3, atol=1e-4)
    # every tensor against its OWN scale (tests/grad_check.py), not against the largest gradient of any parameter
    assert_grads_close([("geo", geo_g.grad, geo.grad)] +
                       [(n, q.grad, p.grad) for (n, p), (_, q) in zip(net.named_parameters(), net_g.named_parameters())])


def test_decoder_gpu_fused_equals_cpu_torch():
    """The whole decoder (fused kernels on the GPU) against the same module on the CPU."""
    import copy
    from gaussianavatar_amd.network import ShapeDecoder
    torch.manual_seed(0)
    dec = ShapeDecoder(18, 32).train()
    x = torch.randn(5000, 18)
    dec_g = copy.deepcopy(dec).cuda()
    outs = dec.forward_points(x)
    (sum(o.sum() for o in outs) + (outs[0] ** 2).sum()).backward()
    xg = x.cuda()
    outs_g = dec_g.forward_points(xg)
    (sum(o.sum() for o in outs_g) + (outs_g[0] ** 2).sum()).backward()
    for a, b in zip(outs, outs_g):
        torch.testing.assert_close(a, b.cpu(), rtol=1e-3, atol=1e-4)
    assert_grads_close([(n, q.grad, p.grad) for (n, p), (_, q) in zip(dec.named_parameters(), dec_g.named_parameters())])


def _softplus_bn(z, sc, sh):
    return F.softplus(z.double() * sc.double() + sh.double())


@pytest.mark.parametrize("row_order", [0, 2])          # GANET_ROWS_DEFAULT / GANET_ROWS_DOWN
@pytest.mark.parametrize("M,K1,K2,N", [(262144, 0, 128, 128), (40000, 72, 0, 128), (33333, 72, 128, 128),
                                       (70001, 0, 128, 3), (4100, 0, 128, 1), (31, 0, 128, 128),
                                       # the producer / consumer kernel of the hidden layers (M a multiple of 32): fewer
                                       # slabs than workgroups, a ragged last round, one slab
                                       (32 * 100, 0, 128, 128), (32 * 1031, 0, 128, 128), (32, 0, 128, 128)])
def test_mlp_fwd_layer_matches_torch(M, K1, K2, N, row_order):
    """ganet_mlp_fwd: activation-on-load GEMM + column statistics, incl. ragged M (tail slab),
    the skip layer's two operands and the narrow output heads."""
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    torch.manual_seed(M % 89)
    dev = "cuda"
    x1 = torch.randn(M, K1, device=dev) if K1 else None
    x2 = (torch.randn(M, K2, device=dev) * 2 + torch.linspace(-4, 4, K2, device=dev)) if K2 else None
    sc = torch.empty(K2, device=dev).uniform_(0.3, 2.0) if K2 else None
    sh = torch.empty(K2, device=dev).uniform_(-25, 25) if K2 else None          # both softplus tails
    W = torch.randn(N, K1 + K2, device=dev) * 0.1
    b = torch.randn(N, device=dev)
    part = torch.zeros(lib.ganet_mlp_stats_floats(N), device=dev)
    z = fused._mlp_fwd(lib, M, N, x1, x2, sc, sh, W, b, part, dev, row_order)
    if N <= 4:       # the output heads run without column statistics
        z_head = fused._mlp_fwd(lib, M, N, x1, x2, sc, sh, W, b, None, dev, row_order)
    cols = []
    if K1:
        cols.append(x1.double())
    if K2:
        cols.append(_softplus_bn(x2, sc, sh))
    ref = torch.cat(cols, 1) @ W.double().t() + b.double()
    tol = 2e-5 * float(ref.abs().max()) + 1e-5
    assert float((z.double() - ref).abs().max()) <= tol
    if N <= 4:
        assert float((z_head.double() - ref).abs().max()) <= tol
    NP = ((N + 31) // 32) * 32
    p = part.reshape(-1, 2, NP).double().sum(0)
    assert float((p[0, :N] - ref.sum(0)).abs().max()) <= 1e-4 * float(ref.abs().sum(0).max()) + 1e-3
    assert float((p[1, :N] - (ref ** 2).sum(0)).abs().max()) <= 1e-4 * float((ref ** 2).sum(0).max()) + 1e-3
    # statistics -> folded scale/shift and running statistics, as F.batch_norm(training=True)
    if N == 128:
        bn = torch.nn.BatchNorm1d(N).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 2.0)
            bn.bias.uniform_(-1, 1)
        mean, rstd, s2, h2 = (torch.empty(N, device=dev) for _ in range(4))
        rm, rv, nbt = bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()
        _native.ganet_check(lib.ganet_mlp_stats(M, N, fused._ptr(part), fused._ptr(bn.weight), fused._ptr(bn.bias),
                                                bn.eps, fused._ptr(mean), fused._ptr(rstd), fused._ptr(s2),
                                                fused._ptr(h2), fused._ptr(rm), fused._ptr(rv), bn.momentum,
                                                fused._ptr(nbt), None, fused._stream(torch.device(dev))))
        y_ref = bn(z)
        torch.testing.assert_close(z * s2 + h2, y_ref, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(rm, bn.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv, bn.running_var, rtol=1e-4, atol=1e-6)
        assert int(nbt) == int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("row_order", [0, 1, 2])       # contiguous ranges / common front up / down
@pytest.mark.parametrize("M,N,K,act", [(262144, 128, 128, True), (50001, 128, 128, True), (70001, 3, 128, True),
                                         (4097, 1, 128, True), (40000, 128, 72, False), (15, 128, 128, True),
                                         (3000, 20, 70, False)])
def test_wgrad_act_matches_torch(M, N, K, act, row_order):
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    torch.manual_seed(N + M % 13)
    dev = "cuda"
    x = torch.randn(M, K, device=dev) * 2
    sc = torch.empty(K, device=dev).uniform_(0.3, 2.0) if act else None
    sh = torch.empty(K, device=dev).uniform_(-3, 3) if act else None
    g = torch.randn(M, N, device=dev)
    dW = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    nbytes = lib.ganet_wgrad_act_workspace(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    gz = coef = None
    gd = g.double()
    if N == 128:          # g operand assembled on load: coef[0] g + coef[1] gz + coef[2]
        gz = torch.randn(M, N, device=dev)
        coef = torch.randn(3, N, device=dev)
        gd = coef[0].double() * g.double() + coef[1].double() * gz.double() + coef[2].double()
    _native.ganet_check(lib.ganet_wgrad_act(M, N, K, fused._ptr(g), g.stride(0), fused._ptr(gz),
                                            0 if gz is None else gz.stride(0), fused._ptr(coef), fused._ptr(x),
                                            x.stride(0), fused._ptr(sc), fused._ptr(sh), fused._ptr(dW), fused._ptr(db),
                                            fused._ptr(ws), nbytes, row_order, fused._stream(torch.device(dev))))
    g = gd
    ref = g.double().t() @ (_softplus_bn(x, sc, sh) if act else x.double())
    assert float((dW.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4
    rb = g.double().sum(0)
    assert float((db.double() - rb).abs().max()) <= 2e-5 * float(rb.abs().max()) + 1e-4


@pytest.mark.parametrize("M,one_pass,native", [(20000, False, False), (4099, False, False), (20000 // 32 * 32, True, False),
                                               (20000 // 32 * 32, True, True), (262144, True, True),
                                               # the three-branch conv6 forward launch: its smallest M (80 slabs: one round),
                                               # one slab more, a slab count that is not a multiple of its 80 workgroups per
                                               # branch, and one slab below its minimum (the single-branch launches)
                                               (2560, True, True), (2592, True, True), (7712, True, True), (2528, True, True)])
def test_fused_decoder_equals_per_layer_formulation(M, one_pass, native, monkeypatch):
    """The whole-decoder function on the fused layer kernels (reference widths: 66 -> 128 x 5 -> three
    heads) against the per-layer formulation (vendor GEMM + fused BN kernels) on the same device:
    outputs, every parameter gradient, the input gradient and the BatchNorm running statistics.
    one_pass: the single-pass hidden-layer backward (ganet_mlp_bwd_fused, the default when M % 32 == 0) vs the
    separate weight- / data-gradient kernels. native: the whole decoder as one C call each way
    (csrc/ganet_decoder.hip) vs the same launches issued one by one from Python."""
    import copy
    from gaussianavatar_amd import fused
    from gaussianavatar_amd.network import ShapeDecoder
    monkeypatch.setattr(fused, "_FUSED_BWD", one_pass)
    monkeypatch.setattr(fused, "_NATIVE_DECODER", native)
    torch.manual_seed(1)
    dec_a = ShapeDecoder(66, 128).cuda().train()
    with torch.no_grad():
        for m in dec_a.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.5, 0.5)
    dec_b = copy.deepcopy(dec_a)
    x_a = torch.randn(M, 66, device="cuda", requires_grad=True)
    x_b = x_a.detach().clone().requires_grad_(True)
    w = [torch.randn(M, k, device="cuda") for k in (3, 1, 3)]
    assert fused.decoder_supported(dec_a, x_a)
    outs_a = dec_a.forward_points(x_a)
    sum((o * wi).sum() for o, wi in zip(outs_a, w)).backward()
    monkeypatch.setattr(fused, "decoder_supported", lambda dec, x: False)
    outs_b = dec_b.forward_points(x_b)
    sum((o * wi).sum() for o, wi in zip(outs_b, w)).backward()
    for a, b in zip(outs_a, outs_b):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4)
    for n, p in dec_a.named_parameters():
        assert p.grad is not None, n
    assert_grads_close([("x", x_a.grad, x_b.grad)] +
                       [(n, p.grad, q.grad) for (n, p), (_, q) in zip(dec_a.named_parameters(), dec_b.named_parameters())])
    for (n, p), (_, q) in zip(dec_a.named_buffers(), dec_b.named_buffers()):
        torch.testing.assert_close(p.float(), q.float(), rtol=1e-4, atol=1e-5, msg=n)
    # evaluation mode (running statistics) runs on the same kernels
    dec_a.eval(); dec_b.eval()
    with torch.no_grad():
        monkeypatch.undo()
        monkeypatch.setattr(fused, "_FUSED_BWD", one_pass)
        monkeypatch.setattr(fused, "_NATIVE_DECODER", native)
        assert fused.decoder_supported(dec_a, x_a.detach())
        e_a = dec_a.forward_points(x_a.detach())
        monkeypatch.setattr(fused, "decoder_supported", lambda dec, x: False)
        e_b = dec_b.forward_points(x_b.detach())
    for a, b in zip(e_a, e_b):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4)


def test_native_decoder_with_a_partial_objective(monkeypatch):
    """An objective that uses only one head (torch.autograd.grad on a partial loss): the one-call decoder backward
    takes zeros for the unused heads' gradients instead of raising and must agree with the per-launch
    path, which skips those heads."""
    import copy
    from gaussianavatar_amd import fused
    from gaussianavatar_amd.network import ShapeDecoder
    torch.manual_seed(5)
    M = 32 * 300
    dec_a = ShapeDecoder(66, 128).cuda().train()
    dec_b = copy.deepcopy(dec_a)
    x = torch.randn(M, 66, device="cuda")
    w = torch.randn(M, 3, device="cuda")
    grads = []
    for dec, native in ((dec_a, True), (dec_b, False)):
        monkeypatch.setattr(fused, "_NATIVE_DECODER", native)
        outs = dec.forward_points(x)
        params = [p for n, p in dec.named_parameters() if not n.startswith(("conv6N", "conv7N", "conv8N", "bn6N", "bn7N",
                                                                             "conv6SH", "conv7SH", "conv8SH", "bn6SH", "bn7SH"))]
        grads.append(torch.autograd.grad((outs[0] * w).sum(), params, allow_unused=True))
    names = [n for n, _ in dec_a.named_parameters()]
    assert_grads_close([(f"param {k}", a, b) for k, (a, b) in enumerate(zip(*grads)) if b is not None and a is not None])
    assert sum(a is not None for a in grads[0]) >= sum(b is not None for b in grads[1]) > 10


@pytest.mark.parametrize("row_order", [0, 2])
@pytest.mark.parametrize("M,O,accumulate,sig", [(262144, 128, False, True), (33001, 128, True, True),
                                                (20000, 128, False, False), (20011, 128, True, False),
                                                (40000, 72, False, False), (4099, 72, True, False), (17, 128, False, True)])
def test_mlp_bwd_data_matches_torch(M, O, accumulate, sig, row_order):
    """ganet_mlp_bwd_data: dz assembled on load from (G, z, coef), times W, optional accumulate, optional
    softplus' epilogue of the source layer with its column sums."""
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    torch.manual_seed(M % 31 + O)
    dev = torch.device("cuda")
    G = torch.randn(M, 128, device=dev)
    z = torch.randn(M, 128, device=dev) * 2
    coef = torch.randn(3, 128, device=dev)
    Wfull = torch.randn(128, O + 5, device=dev) * 0.1          # the kernel reads a column slice [128, O]
    Wt = Wfull[:, 3:3 + O].t()
    out = torch.randn(M, O, device=dev)
    prev = out.clone()
    src_z = torch.randn(M, O, device=dev) * 2 if sig else None
    sc = torch.empty(O, device=dev).uniform_(0.3, 2.0) if sig else None
    sh = torch.empty(O, device=dev).uniform_(-12, 25) if sig else None
    parts = lib.ganet_mlp_bwd_data_parts()
    part = torch.zeros(parts * 256, device=dev)
    _native.ganet_check(lib.ganet_mlp_bwd_data(M, O, fused._ptr(G), 128, fused._ptr(z), 128, fused._ptr(coef),
                                               fused._ptr(Wfull[:, 3:]), Wfull.stride(0), fused._ptr(out), O,
                                               int(accumulate), fused._ptr(src_z),
                                               O if sig else 0, fused._ptr(sc), fused._ptr(sh),
                                               fused._ptr(part) if sig else None, row_order, fused._stream(dev)))
    dz = coef[0].double() * G.double() + coef[1].double() * z.double() + coef[2].double()
    ref = dz @ Wt.double().t()
    if accumulate:
        ref = ref + prev.double()
    if sig:
        u = src_z.double() * sc.double() + sh.double()
        ref = ref * torch.where(u > 20, torch.ones_like(u), torch.sigmoid(u))
    assert float((out.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 2e-5
    if sig:
        p = part.reshape(parts, 2, 128).double().sum(0)
        assert float((p[0] - ref.sum(0)).abs().max()) <= 1e-4 * float(ref.abs().sum(0).max()) + 1e-3
        r2 = (ref * src_z.double()).sum(0)
        assert float((p[1] - r2).abs().max()) <= 1e-4 * float((ref * src_z.double()).abs().sum(0).max()) + 1e-3


@pytest.mark.parametrize("ride", [False, True])
@pytest.mark.parametrize("M,N8", [(262144, 3), (5001, 1), (37, 3)])
def test_mlp_head_bwd_and_stats_match_torch_batchnorm_backward(M, N8, ride):
    """Head kernel + coefficient kernel: dz = A G + q z + p must equal autograd's gradient of
    softplus(batch_norm(z)) w.r.t. z; d gamma / d beta likewise. ride: the head's own weight / bias
    gradient comes out of the same pass (wgrad_part + ganet_wgrad_reduce_batch)."""
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    torch.manual_seed(N8)
    dev = torch.device("cuda")
    z = (torch.randn(M, 128, device=dev) * 1.5 + torch.linspace(-2, 2, 128, device=dev)).requires_grad_(True)
    gamma = torch.empty(128, device=dev).uniform_(0.5, 2).requires_grad_(True)
    beta = torch.empty(128, device=dev).uniform_(-1, 1).requires_grad_(True)
    W8 = (torch.randn(N8, 128, device=dev) * 0.3).requires_grad_(True)
    b8 = torch.zeros(N8, device=dev, requires_grad=True)
    g = torch.randn(M, N8, device=dev)
    y = F.softplus(F.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5))
    ((y @ W8.t() + b8) * g).sum().backward()
    with torch.no_grad():
        mean = z.mean(0)
        rstd = torch.rsqrt(z.var(0, unbiased=False) + 1e-5)
        sc = (gamma * rstd).contiguous()
        sh = (beta - mean * sc).contiguous()
    G = torch.empty(M, 128, device=dev)
    parts = lib.ganet_mlp_head_bwd_parts()
    part = torch.zeros(parts * 256, device=dev)
    wpart = torch.full((parts * (N8 * 128 + N8),), float("nan"), device=dev) if ride else None
    _native.ganet_check(lib.ganet_mlp_head_bwd(M, N8, fused._ptr(g), fused._ptr(W8.detach()), fused._ptr(z.detach()), 128,
                                               fused._ptr(sc), fused._ptr(sh), fused._ptr(G), 128, fused._ptr(part),
                                               fused._ptr(wpart), fused._stream(dev)))
    if ride:
        dW8, db8 = torch.empty(N8, 128, device=dev), torch.empty(N8, device=dev)
        jobs = (_native.GanetWgradJob * 1)()
        jobs[0].workspace, jobs[0].M, jobs[0].N, jobs[0].K = wpart.data_ptr(), M, N8, 128
        jobs[0].dW, jobs[0].db, jobs[0].nblocks = dW8.data_ptr(), db8.data_ptr(), parts
        _native.ganet_check(lib.ganet_wgrad_reduce_batch(1, jobs, fused._stream(dev)))
        assert float((dW8 - W8.grad).abs().max()) <= 2e-5 * float(W8.grad.abs().max()) + 1e-4
        assert float((db8 - b8.grad).abs().max()) <= 2e-5 * float(b8.grad.abs().max()) + 1e-4
    coef = torch.empty(3, 128, device=dev)
    dg = torch.empty(128, device=dev)
    db = torch.empty(128, device=dev)
    _native.ganet_check(lib.ganet_mlp_bwd_stats(M, parts, fused._ptr(part), fused._ptr(mean), fused._ptr(rstd),
                                                fused._ptr(sc), fused._ptr(coef), fused._ptr(dg), fused._ptr(db),
                                                fused._stream(dev)))
    dz = coef[0] * G + coef[1] * z.detach() + coef[2]
    tol = 2e-3 * float(z.grad.abs().max()) + 1e-7
    assert float((dz - z.grad).abs().max()) <= tol
    assert float((dg - gamma.grad).abs().max()) <= 1e-3 * float(gamma.grad.abs().max()) + 1e-4
    assert float((db - beta.grad).abs().max()) <= 1e-3 * float(beta.grad.abs().max()) + 1e-4


@pytest.mark.parametrize("b,HW,N", [(1, 512 * 512, 200000), (2, 4096, 3000), (1, 100, 0)])
def test_decode_pack_matches_torch_chain(b, HW, N):
    """ganet_decode_pack_fwd/bwd against the reference's element-wise chain: x0.02, sigmoid heads, scale
    warm-up, gather of the valid texels, the offset regulariser mean((0.02 res)^2) over all texels and
    the scale regulariser mean(scales) over the valid ones."""
    from gaussianavatar_amd import fused
    torch.manual_seed(HW % 11)
    dev = "cuda"
    valid = torch.randperm(HW, device=dev)[:N].sort().values
    inv = torch.full((HW,), -1, dtype=torch.int64, device=dev)
    inv[valid] = torch.arange(N, device=dev)
    mk = lambda c: (torch.randn(b, HW, c, device=dev) * 2).requires_grad_(True)
    r1, s1, c1 = mk(3), mk(1), mk(3)
    r2, s2, c2 = (t.detach().clone().requires_grad_(True) for t in (r1, s1, c1))
    flat, sq, sc = fused.decode_pack(r1, s1, c1, valid, inv, 0.02, 0.007)
    res, scale, col = fused.split_records(flat, b, N)
    assert res.is_contiguous() and scale.is_contiguous() and col.is_contiguous()
    pick = lambda t: t.index_select(1, valid)
    ref_res, ref_scale, ref_col = pick(r2 * 0.02), pick(torch.sigmoid(s2) * 0.007), pick(torch.sigmoid(c2))
    ref_sq = ((r2 * 0.02) ** 2).mean()
    ref_sc = ref_scale.mean() if N else ref_scale.sum()
    torch.testing.assert_close(res, ref_res, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(scale, ref_scale, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(col, ref_col, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(sq, ref_sq, rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(sc, ref_sc, rtol=1e-4, atol=1e-9)
    w = [torch.randn_like(t) for t in (ref_res, ref_scale, ref_col)]
    ((res * w[0]).sum() + (scale.expand(-1, -1, 3) * w[0]).sum() + (col * w[2]).sum() + 3.0 * sq + 0.5 * sc).backward()
    ((ref_res * w[0]).sum() + (ref_scale.expand(-1, -1, 3) * w[0]).sum() + (ref_col * w[2]).sum()
     + 3.0 * ref_sq + 0.5 * ref_sc).backward()
    for a, bb in ((r1, r2), (s1, s2), (c1, c2)):
        torch.testing.assert_close(a.grad, bb.grad, rtol=1e-4, atol=1e-7)


def test_decode_pack_regularisers_alone_have_gradients():
    """Only the two regulariser means are differentiated (no gradient reaches the records)."""
    from gaussianavatar_amd import fused
    HW, N, dev = 1000, 400, "cuda"
    valid = torch.arange(0, 2 * N, 2, device=dev)
    inv = torch.full((HW,), -1, dtype=torch.int64, device=dev)
    inv[valid] = torch.arange(N, device=dev)
    r1, s1, c1 = (torch.randn(1, HW, c, device=dev, requires_grad=True) for c in (3, 1, 3))
    r2, s2 = r1.detach().clone().requires_grad_(True), s1.detach().clone().requires_grad_(True)
    _flat, sq, sc = fused.decode_pack(r1, s1, c1, valid, inv, 0.02, 1.0)
    (2.0 * sq + sc).backward()
    (2.0 * ((r2 * 0.02) ** 2).mean() + torch.sigmoid(s2).index_select(1, valid).mean()).backward()
    torch.testing.assert_close(r1.grad, r2.grad, rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(s1.grad, s2.grad, rtol=1e-4, atol=1e-9)
    assert float(c1.grad.abs().max()) == 0.0


def test_l1_and_ssim_share_one_pass():
    """losses.l1_loss_w + losses.ssim on the same pair: values and the combined gradient against torch;
    the second call is served from the first one's pass (one autograd node), in either order."""
    from gaussianavatar_amd import fused, losses
    torch.manual_seed(2)
    shape = (2, 3, 75, 130)
    a = torch.rand(*shape)
    b = (a + 0.1 * torch.randn(*shape)).clamp(0, 1)
    ac = a.clone().requires_grad_(True)
    ref = 0.8 * torch.abs(ac - b).mean() + 0.2 * (1 - losses.ssim(ac, b))
    ref.backward()
    for order in ("l1_first", "ssim_first"):
        ag, bg = a.cuda().requires_grad_(True), b.cuda()
        fused.profile_enable(["ssim_fwd", "ssim_bwd"]); fused.profile_read(True)
        if order == "l1_first":
            l1 = losses.l1_loss_w(ag, bg); s = losses.ssim(ag, bg)
        else:
            s = losses.ssim(ag, bg); l1 = losses.l1_loss_w(ag, bg)
        loss = 0.8 * l1 + 0.2 * (1 - s)
        loss.backward()
        prof = fused.profile_read(True); fused.profile_enable(False)
        assert prof["ssim_fwd"][1] == 1 and prof["ssim_bwd"][1] == 1, prof
        assert abs(float(loss) - float(ref)) < 2e-6
        assert float((ag.grad.cpu() - ac.grad).abs().max()) <= 2e-4 * float(ac.grad.abs().max())
        # a different pair does not hit the parked value
        other = losses.ssim(ag.detach() * 0.5, bg)
        assert abs(float(other) - float(s)) > 1e-3
    # L1 alone (nothing parked is left behind for an unrelated later call)
    ag = a.cuda().requires_grad_(True)
    l1 = losses.l1_loss_w(ag, b.cuda())
    l1.backward()
    torch.testing.assert_close(ag.grad.cpu(), torch.sign(a - b) / a.numel(), rtol=1e-6, atol=1e-12)


def test_mean_sq_and_weighted_sum():
    from gaussianavatar_amd import fused, losses
    x = torch.randn(1, 64, 37, 41, device="cuda", requires_grad=True)
    y = x.detach().clone().requires_grad_(True)
    m = fused.mean_sq(x)
    terms_a = [m, (x.sum() * 1e-3), torch.tensor(0.25, device="cuda", requires_grad=True)]
    terms_b = [(y ** 2).mean(), (y.sum() * 1e-3), terms_a[2].detach().clone().requires_grad_(True)]
    la = losses.weighted_sum(terms_a, [2.0, -0.5, 3.0], bias=0.2)
    lb = 0.2 + 2.0 * terms_b[0] - 0.5 * terms_b[1] + 3.0 * terms_b[2]
    torch.testing.assert_close(la, lb, rtol=1e-5, atol=1e-7)
    la.backward(); lb.backward()
    torch.testing.assert_close(x.grad, y.grad, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(terms_a[2].grad, terms_b[2].grad)
    # CPU tensors take the plain torch composition
    assert float(losses.weighted_sum([torch.tensor(2.0), torch.tensor(3.0)], [1.0, -1.0], bias=0.5)) == -0.5


@pytest.mark.parametrize("b,feat,S", [(1, 128, 512), (2, 16, 64), (1, 8, 24)])
def test_upsample_cat_matches_grid_sample(b, feat, S):
    """ganet_upsample_cat_fwd/bwd (2x2 bilinear taps at the separable texel grid + uv columns + zero
    padding) against F.grid_sample(align_corners=False, zero padding) + cat, forward and backward."""
    from gaussianavatar_amd import fused
    from gaussianavatar_amd.network import POP_no_unet, uv_to_grid
    torch.manual_seed(S)
    net = POP_no_unet(c_geom=64, hsize=128).cuda()
    idx = torch.stack(torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij"), -1).reshape(-1, 2).float() / (S - 1)
    uv = idx.cuda()[None].expand(b, -1, -1).contiguous()
    mats = net._separable_bilinear(uv, feat, S)
    assert mats is not None
    taps = net._bilinear_taps(mats)
    pix1 = torch.randn(b, 64, feat, feat, device="cuda", requires_grad=True)
    pix2 = pix1.detach().clone().requires_grad_(True)
    x = fused.upsample_cat(pix1, uv, taps[0], taps[1], 72)
    ref = F.grid_sample(pix2, uv_to_grid(uv, S), mode="bilinear", align_corners=False)
    ref = torch.cat([ref.reshape(b, 64, S * S).transpose(1, 2), uv, uv.new_zeros(b, S * S, 6)], 2).reshape(b * S * S, 72)
    torch.testing.assert_close(x, ref, rtol=1e-5, atol=1e-6)
    w = torch.randn_like(ref)
    (x * w).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(pix1.grad, pix2.grad, rtol=1e-4, atol=1e-5)


def test_one_launch_adam_matches_torch_adam():
    """optim.Adam (ganet_adam_step) against torch.optim.Adam: two groups with different learning rates,
    more tensors than one launch's table holds, a scheduler step in between, and state_dict interchange
    in both directions."""
    from gaussianavatar_amd.optim import Adam
    torch.manual_seed(5)
    shapes = [(128, 66, 1), (128,), (3, 128, 1), (1, 64, 33, 31), (7,)] + [(5, 3)] * 70
    mk = lambda: [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    pa = mk()
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:3], "lr": 3e-3}, {"params": ps[3:], "lr": 5e-4}]
    oa, ob = Adam(groups(pa)), torch.optim.Adam(groups(pb))
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, [3], gamma=0.1)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, [3], gamma=0.1)

    def run(o, s, ps, steps, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        for _ in range(steps):
            o.zero_grad()
            for i, p in enumerate(ps):
                if i != 4:                               # a parameter that never gets a gradient
                    p.grad = torch.randn(p.shape, device="cuda", generator=g) * (1 + i % 3)
            o.step()
            s.step()

    run(oa, sa, pa, 6, 1); run(ob, sb, pb, 6, 1)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)
    assert torch.equal(pa[4], pb[4])
    # interchange: continue each optimiser from the OTHER one's state
    sda, sdb = oa.state_dict(), ob.state_dict()
    assert set(sda["state"][0].keys()) == set(sdb["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sda["state"][0]["step"]) == float(sdb["state"][0]["step"]) == 6.0
    oa2, ob2 = Adam(groups(pa)), torch.optim.Adam(groups(pb))
    oa2.load_state_dict(sdb); ob2.load_state_dict(sda)
    sa2 = torch.optim.lr_scheduler.MultiStepLR(oa2, [100]); sb2 = torch.optim.lr_scheduler.MultiStepLR(ob2, [100])
    run(oa2, sa2, pa, 3, 2); run(ob2, sb2, pb, 3, 2)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=4e-6, atol=4e-7)
    assert float(oa2.state_dict()["state"][0]["step"]) == 9.0
    # a fused-Adam checkpoint (device-side step tensors) loads too
    of = torch.optim.Adam(groups(pb), fused=True)
    for p in pb:
        p.grad = torch.ones_like(p)
    of.step()
    oa3 = Adam(groups(pa))
    oa3.load_state_dict(of.state_dict())
    for p in pa:
        p.grad = torch.ones_like(p)
    oa3.step()
    assert float(oa3.state_dict()["state"][0]["step"]) == 2.0


def test_adam_on_a_channels_last_parameter():
    """The geometry feature map is kept channels-last on a HIP device (avatar_model.net_set): the one-launch Adam walks
    memory, so parameter, gradient and moments must share one layout whatever layout gradients arrive in or a checkpoint
    was written with. Against torch.optim.Adam on the row-major twin."""
    from gaussianavatar_amd.optim import Adam
    torch.manual_seed(9)
    base = torch.randn(1, 64, 16, 24, device="cuda")
    pa = torch.nn.Parameter(base.clone().contiguous(memory_format=torch.channels_last))
    pb = torch.nn.Parameter(base.clone())
    oa, ob = Adam([pa], lr=2e-3), torch.optim.Adam([pb], lr=2e-3)
    g = torch.Generator(device="cuda").manual_seed(3)
    for step in range(4):
        grad = torch.randn(base.shape, device="cuda", generator=g)
        # gradients arrive row-major, channels-last, or as a permuted view of an NHWC buffer (what geom_convs returns)
        pa.grad = (grad.clone(), grad.contiguous(memory_format=torch.channels_last),
                   grad.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), grad.clone())[step]
        pb.grad = grad.clone()
        oa.step(); ob.step()
        assert pa.is_contiguous(memory_format=torch.channels_last) and pa.grad.stride() == pa.stride()
    torch.testing.assert_close(pa.detach().contiguous(), pb.detach(), rtol=2e-6, atol=2e-7)
    # a checkpoint written by the row-major optimiser continues on the channels-last parameter
    oa2 = Adam([pa], lr=2e-3)
    oa2.load_state_dict(ob.state_dict())
    grad = torch.randn(base.shape, device="cuda", generator=g)
    pa.grad, pb.grad = grad.clone(), grad.clone()
    oa2.step(); ob.step()
    torch.testing.assert_close(pa.detach().contiguous(), pb.detach(), rtol=4e-6, atol=4e-7)
    st = oa2.state[pa]
    assert st["exp_avg"].stride() == pa.stride() and st["exp_avg_sq"].stride() == pa.stride()


def test_production_width_nets_match_reference_on_the_fused_path():
    """The reference's own POP_no_unet / UnetNoCond5DS at c_geom 64 / hsize 128 / nf 32 (golden outputs AND
    gradients, oracle/make_golden.py:make_net_full) against the HIP path these widths select: fused
    up-sampling + fused fp32-MFMA decoder (stage-1 and stage-2 call patterns)."""
    from gaussianavatar_amd import fused
    from gaussianavatar_amd.network import POP_no_unet
    from tests.test_oracle_golden import check_net_full
    probe = POP_no_unet(c_geom=64, hsize=128).cuda()
    assert fused.decoder_supported(probe.decoder, torch.empty(8, 66, device="cuda"))
    check_net_full("cuda", tol_out=1e-4, tol_grad=2e-3)


def test_l1_ssim_pair_cache_respects_grad_mode():
    """Advisor finding r1: a value parked under no_grad must not answer a call that wants gradients."""
    from gaussianavatar_amd.losses import l1_loss_w, ssim
    img = torch.rand(1, 3, 64, 64, device="cuda", requires_grad=True)
    gt = torch.rand(1, 3, 64, 64, device="cuda")
    with torch.no_grad():
        l1_loss_w(img, gt)                      # logging-style call; parks the SSIM value without a graph
    s = ssim(img, gt)
    assert s.requires_grad
    s.backward()
    assert img.grad is not None and float(img.grad.abs().sum()) > 0


@pytest.mark.parametrize("row_order", [0, 2])
@pytest.mark.parametrize("M,accumulate,act,wslice", [(262144, False, True, False), (32 * 1031, True, True, True),
                                                     (32 * 700, True, False, False), (32 * 257, False, False, True),
                                                     (64, False, True, False), (32, True, True, True)])
def test_mlp_bwd_fused_matches_float64(M, accumulate, act, wslice, row_order):
    """ganet_mlp_bwd_fused (csrc/ganet_layer_bwd.hip): data gradient (accumulating or not; with act: -> G_src and
    its column sums) AND weight / bias gradient of a hidden 128 -> 128 layer in one pass over the activations,
    against float64 torch. wslice: W is a column slice of a wider weight (row stride 194, as conv5's)."""
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    torch.manual_seed(M % 29)
    dev = torch.device("cuda")
    G = torch.randn(M, 128, device=dev)
    z = torch.randn(M, 128, device=dev) * 2
    coef = torch.randn(3, 128, device=dev)
    Wfull = torch.randn(128, 194, device=dev) * 0.1               # [out n, in]
    W = Wfull[:, 66:] if wslice else Wfull[:, :128].contiguous()
    src_z = torch.randn(M, 128, device=dev) * 2
    sc = torch.empty(128, device=dev).uniform_(0.3, 2.0)
    sh = torch.empty(128, device=dev).uniform_(-12, 25)
    out = torch.randn(M, 128, device=dev) if accumulate else torch.full((M, 128), float("nan"), device=dev)
    prev = out.clone()
    parts = lib.ganet_mlp_bwd_fused_parts()
    part = torch.full((parts * 256,), float("nan"), device=dev)
    wsb = lib.ganet_mlp_bwd_fused_workspace()
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ws.view(torch.float32).fill_(float("nan"))
    _native.ganet_check(lib.ganet_mlp_bwd_fused(M, fused._ptr(G), fused._ptr(z), fused._ptr(coef), fused._ptr(W),
                                                W.stride(0), fused._ptr(out), int(accumulate), fused._ptr(src_z),
                                                fused._ptr(sc), fused._ptr(sh), int(act), fused._ptr(part),
                                                ws.data_ptr(), wsb, row_order, fused._stream(dev)))
    dW, db = torch.empty(128, 128, device=dev), torch.empty(128, device=dev)
    jobs = (_native.GanetWgradJob * 1)()
    jobs[0].workspace, jobs[0].M, jobs[0].N, jobs[0].K = ws.data_ptr(), M, 128, 128
    jobs[0].dW, jobs[0].db, jobs[0].nblocks = dW.data_ptr(), db.data_ptr(), parts
    _native.ganet_check(lib.ganet_wgrad_reduce_batch(1, jobs, fused._stream(dev)))
    dz = coef[0].double() * G.double() + coef[1].double() * z.double() + coef[2].double()
    u = src_z.double() * sc.double() + sh.double()
    ref = dz @ W.double()
    if accumulate:
        ref = ref + prev.double()
    if act:
        ref = ref * torch.where(u > 20, torch.ones_like(u), torch.sigmoid(u))
    assert float((out.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 2e-5
    if act:
        p = part.reshape(parts, 2, 128).double().sum(0)
        assert float((p[0] - ref.sum(0)).abs().max()) <= 1e-4 * float(ref.abs().sum(0).max()) + 1e-3
        r2 = (ref * src_z.double()).sum(0)
        assert float((p[1] - r2).abs().max()) <= 1e-4 * float((ref * src_z.double()).abs().sum(0).max()) + 1e-3
    x = torch.nn.functional.softplus(u)
    refW = dz.t() @ x
    assert float((dW.double() - refW).abs().max()) <= 2e-5 * float(refW.abs().max()) + 1e-4
    assert float((db.double() - dz.sum(0)).abs().max()) <= 2e-5 * float(dz.abs().sum(0).max()) + 1e-4


@pytest.mark.parametrize("row_order", [0, 2])
@pytest.mark.parametrize("M,accumulate,wide", [(262144, False, True), (32 * 1031, True, False), (64, True, True), (32, False, False)])
def test_mlp_bwd_fused_input_matches_float64(M, accumulate, wide, row_order):
    """ganet_mlp_bwd_fused_input: the one-pass backward of a layer fed by the raw decoder input x [M,72] (conv1, the
    input half of conv5): input gradient [M,72] (columns >= 66 zero-filled) and weight / bias gradient, against
    float64. wide: W is the left column slice of conv5's [128, 194] weight."""
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    torch.manual_seed(M % 23 + 1)
    dev = torch.device("cuda")
    cin = 66
    G = torch.randn(M, 128, device=dev)
    z = torch.randn(M, 128, device=dev) * 2
    coef = torch.randn(3, 128, device=dev)
    Wfull = torch.randn(128, 194 if wide else cin, device=dev) * 0.1
    W = Wfull[:, :cin]
    x = torch.randn(M, 72, device=dev)
    x[:, cin:] = 0
    out = torch.randn(M, 72, device=dev) if accumulate else torch.full((M, 72), float("nan"), device=dev)
    prev = out.clone()
    parts = lib.ganet_mlp_bwd_fused_parts()
    wsb = lib.ganet_mlp_bwd_fused_workspace()
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ws.view(torch.float32).fill_(float("nan"))
    _native.ganet_check(lib.ganet_mlp_bwd_fused_input(M, fused._ptr(G), fused._ptr(z), fused._ptr(coef), fused._ptr(W),
                                                      W.stride(0), cin, fused._ptr(out), 72, int(accumulate), fused._ptr(x),
                                                      ws.data_ptr(), wsb, row_order, fused._stream(dev)))
    dW, db = torch.empty(128, 128, device=dev), torch.empty(128, device=dev)
    jobs = (_native.GanetWgradJob * 1)()
    jobs[0].workspace, jobs[0].M, jobs[0].N, jobs[0].K = ws.data_ptr(), M, 128, 128
    jobs[0].dW, jobs[0].db, jobs[0].nblocks = dW.data_ptr(), db.data_ptr(), parts
    _native.ganet_check(lib.ganet_wgrad_reduce_batch(1, jobs, fused._stream(dev)))
    dz = coef[0].double() * G.double() + coef[1].double() * z.double() + coef[2].double()
    ref = dz @ W.double()
    if accumulate:
        ref = ref + prev[:, :cin].double()
    got = out[:, :cin].double()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 2e-5
    assert torch.isfinite(out).all()
    refW = dz.t() @ x[:, :cin].double()
    assert float((dW[:, :cin].double() - refW).abs().max()) <= 2e-5 * float(refW.abs().max()) + 1e-4
    assert float((db.double() - dz.sum(0)).abs().max()) <= 2e-5 * float(dz.abs().sum(0).max()) + 1e-4


def test_batchnorm_statistics_survive_a_large_mean():
    """Advisor finding r1: E[z^2] - mean^2 from raw fp32 sums cancels once |mean| >> std. The fused layer
    accumulates its statistics about the BatchNorm layer's running mean: with columns at mean ~1e3, std ~1
    the folded scale/shift must still match float64 BatchNorm (and the unshifted sums must visibly not)."""
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    dev = torch.device("cuda")
    torch.manual_seed(5)
    M, N = 262144, 128
    x2 = torch.randn(M, 128, device=dev)
    sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    W = torch.randn(N, 128, device=dev) * 0.05
    b = torch.full((N,), 1000.0, device=dev) + torch.randn(N, device=dev)          # |mean| ~ 1e3, std ~ 0.4
    gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    z64 = torch.nn.functional.softplus(x2.double()) @ W.double().t() + b.double()
    var64 = z64.var(0, unbiased=False)
    errs = {}
    for name, shift in (("shifted", z64.mean(0).float() + 0.3), ("raw", None)):      # running mean: close, not exact
        part = torch.zeros(lib.ganet_mlp_stats_floats(N), device=dev)
        fused._mlp_fwd(lib, M, N, None, x2, sc, sh, W, b, part, dev, 0, shift)
        mean, rstd, s2, h2 = (torch.empty(N, device=dev) for _ in range(4))
        _native.ganet_check(lib.ganet_mlp_stats(M, N, fused._ptr(part), fused._ptr(gamma), fused._ptr(beta), 1e-5,
                                                fused._ptr(mean), fused._ptr(rstd), fused._ptr(s2), fused._ptr(h2),
                                                None, None, 0.1, None, fused._ptr(shift), fused._stream(dev)))
        var = 1.0 / rstd.double() ** 2 - 1e-5
        errs[name] = float(((var - var64).abs() / var64).max())
        if name == "shifted":
            assert float((mean.double() - z64.mean(0)).abs().max()) <= 1e-3
    assert errs["shifted"] <= 1e-3, errs
    assert errs["raw"] > 10 * errs["shifted"], errs            # the cancellation the shift removes


def test_split_mfma_is_fp32_accurate():
    """The decoder GEMMs feed the bf16 matrix pipe with an exact three-way split of their fp32 operands
    (csrc/ganet_split.h: six bf16 products per fp32 product, fp32 accumulation). That is an fp32 GEMM, not a
    bf16 one: against float64 its error must not exceed that of an fp32 GEMM of the same operands (measured against
    round 1's v_mfma_f32_32x32x2_f32 kernels before they were removed: equal or smaller, because an MFMA step adds
    16 exact products before it rounds), while a plain bf16 GEMM is three orders of magnitude away."""
    from gaussianavatar_amd import _native, fused
    lib = _native.ganet()
    dev = torch.device("cuda")
    torch.manual_seed(5)
    M = 262144
    P, st = fused._ptr, fused._stream(dev)
    z = torch.randn(M, 128, device=dev) * 1.5 + 0.3
    sc, sh = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
    W = torch.randn(128, 128, device=dev) * 0.1
    b = torch.randn(128, device=dev)
    g, gz = torch.randn(M, 128, device=dev), torch.randn(M, 128, device=dev)
    coef = torch.randn(3, 128, device=dev)
    act64 = F.softplus(z.double() * sc.double() + sh.double())
    dz64 = g.double() * coef[0].double() + gz.double() * coef[1].double() + coef[2].double()
    part = torch.zeros(lib.ganet_mlp_stats_floats(128), device=dev)
    bpart = torch.zeros(lib.ganet_mlp_bwd_data_parts() * 256, device=dev)
    nb = lib.ganet_wgrad_act_workspace(M, 128, 128)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dW, db = torch.empty(128, 128, device=dev), torch.empty(128, device=dev)
    out = torch.empty(M, 128, device=dev)

    def fwd():
        return fused._mlp_fwd(lib, M, 128, None, z, sc, sh, W, b, part, dev)

    def bwd():
        _native.ganet_check(lib.ganet_mlp_bwd_data(M, 128, P(g), 128, P(gz), 128, P(coef), P(W), 128, P(out), 128, 0,
                                                   P(z), 128, P(sc), P(sh), P(bpart), 0, st))
        return out.clone()

    def wgrad():
        _native.ganet_check(lib.ganet_wgrad_act(M, 128, 128, P(g), 128, P(gz), 128, P(coef), P(z), 128, P(sc), P(sh),
                                                P(dW), P(db), P(ws), nb, 0, st))
        return dW.clone()

    refs = {"fwd": act64 @ W.double().t() + b.double(),
            "bwd": (dz64 @ W.double()) * torch.sigmoid(z.double() * sc.double() + sh.double()),
            "wgrad": dz64.t() @ act64}
    err = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd), ("wgrad", wgrad)):
        err[name] = float((fn().double() - refs[name]).abs().max() / refs[name].abs().max())
    bf16 = (act64.float().bfloat16() @ W.bfloat16().t()).double() + b.double()
    err_bf16 = float((bf16 - refs["fwd"]).abs().max() / refs["fwd"].abs().max())
    # an fp32 FMA chain of the same GEMM (torch's fp32 matmul) as the yardstick the split kernels must match
    fma = {"fwd": (act64.float() @ W.t() + b).double(),
           "bwd": ((dz64.float() @ W).double()) * torch.sigmoid(z.double() * sc.double() + sh.double()),
           "wgrad": (dz64.float().t() @ act64.float()).double()}
    for name in ("fwd", "bwd", "wgrad"):
        e_fma = float((fma[name] - refs[name]).abs().max() / refs[name].abs().max())
        assert err[name] <= 1.25 * e_fma + 4e-7, (name, err, e_fma)
        assert err[name] < 2e-6, (name, err)
    assert err_bf16 > 100 * err["fwd"], (err_bf16, err)


@pytest.mark.parametrize("b,H,W", [(1, 128, 128), (2, 64, 128), (1, 8, 64)])
def test_geom_convs_match_torch_float64(b, H, W):
    """csrc/ganet_conv.hip (forward, input gradient, weight gradient of the three 5x5 convolutions of
    GeomConvLayers) against F.conv2d and its autograd in float64."""
    from gaussianavatar_amd import fused
    torch.manual_seed(H + W)
    x = torch.randn(b, 64, H, W, device="cuda", requires_grad=True)
    ws = [(torch.randn(64, 64, 5, 5, device="cuda") * 0.03).requires_grad_(True) for _ in range(3)]
    assert fused.geom_convs_supported(x, ws)
    y = fused.geom_convs(x, ws)
    g = torch.randn(b, 64, H, W, device="cuda")
    y.backward(g)
    xr = x.detach().double().requires_grad_(True)
    wr = [w.detach().double().requires_grad_(True) for w in ws]
    yr = xr
    for w in wr:
        yr = F.conv2d(yr, w, padding=2)
    yr.backward(g.double())
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    assert tuple(y.shape) == (b, 64, H, W)
    assert rel(y, yr) < 2e-6, rel(y, yr)
    assert rel(x.grad, xr.grad) < 2e-6, rel(x.grad, xr.grad)
    for w, r in zip(ws, wr):
        assert rel(w.grad, r.grad) < 2e-6, rel(w.grad, r.grad)


@pytest.mark.parametrize("b,B,N,missing", [(1, 3, 5000, None), (2, 2, 777, None), (1, 2, 200_000, None), (1, 2, 100, "scale")])
def test_expand_records_backward_is_the_expand_backward(b, B, N, missing):
    """fused.expand_records: the packed record buffer's views broadcast over the frames (and the scale over three
    axes); its one-kernel backward (ganet_records_bwd) against autograd's own expand / slice backward."""
    from gaussianavatar_amd import fused
    torch.manual_seed(N)
    flat = torch.randn(b * N * 7, device="cuda")
    fa = flat.clone().requires_grad_()
    fb = flat.clone().requires_grad_()
    res, sc3, col = fused.expand_records(fa, b, N, B)
    r0, s1, c0 = fb[:b * N * 3].view(b, N, 3), fb[b * N * 3:b * N * 4].view(b, N, 1), fb[b * N * 4:].view(b, N, 3)
    if b != B:
        r0, s1, c0 = (t.expand(B, -1, -1) for t in (r0, s1, c0))
    s3 = s1.expand(-1, -1, 3)
    assert res.shape == r0.shape == (B, N, 3) and sc3.shape == s3.shape and col.shape == c0.shape
    assert torch.equal(res, r0) and torch.equal(sc3, s3) and torch.equal(col, c0)
    w = [torch.randn(B, N, 3, device="cuda") for _ in range(3)]
    terms_a = [(res * w[0]).sum(), (sc3 * w[1]).sum(), (col * w[2]).sum()]
    terms_b = [(r0 * w[0]).sum(), (s3 * w[1]).sum(), (c0 * w[2]).sum()]
    if missing == "scale":                      # a view nobody differentiates: its gradient arrives as None
        terms_a.pop(1); terms_b.pop(1)
    sum(terms_a).backward()
    sum(terms_b).backward()
    torch.testing.assert_close(fa.grad, fb.grad, rtol=1e-6, atol=1e-6)

"""GPU parity of the decoder's map path (csrc/ganet_upz.hip, include/ganet.h "commuted with the bilinear up-sampling"):
conv1 and the input half of conv5 (/root/reference/model/modules.py:555,559) evaluated on the feature map and
gathered by the up-sampling instead of GEMMs over the up-sampled input rows (/root/reference/model/network.py:60-81).
Every kernel against a float64 torch restatement, then the assembled path against the path it replaces."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.grad_check import assert_grads_close

pytestmark = pytest.mark.gpu


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _grid_setup(b, feat, S, shared_uv=False):
    """(net, uv [b,S*S,2], dense bilinear matrices (Wr, Wc) [S,R], tap lists, GanetUpGrid)."""
    from gaussianavatar_amd import fused
    from gaussianavatar_amd.network import POP_no_unet
    net = POP_no_unet(c_geom=64, hsize=128).cuda()
    idx = torch.stack(torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij"), -1).reshape(-1, 2).float() / (S - 1)
    uv = idx.cuda()[None].expand(b, -1, -1) if shared_uv else idx.cuda()[None].expand(b, -1, -1).contiguous()
    mats = net._separable_bilinear(uv, feat, S)
    assert mats is not None
    taps = net._bilinear_taps(mats)
    grid = fused._up_grid(b, S, feat, taps[0], taps[1], uv)
    return net, uv, mats, taps, grid


def _upsample64(mats, maps, b, R, S):
    """bilinear up-sampling of maps [b*R*R, C] (float64) at the texel grid -> [b*S*S, C]"""
    Wr, Wc = (m.double() for m in mats)
    C = maps.shape[1]
    t = maps.reshape(b, R, R, C)
    return torch.einsum("ip,jq,bpqc->bijc", Wr, Wc, t).reshape(b * S * S, C)


@pytest.mark.parametrize("M,N,K,acc", [(16384, 256, 64, False), (16384, 64, 256, False), (32, 32, 64, True), (4096, 128, 128, True)])
def test_rowgemm_matches_float64(M, N, K, acc):
    from gaussianavatar_amd import _native
    lib = _native.ganet()
    torch.manual_seed(N + K)
    lda, ldb, ldc = K + 8, K, N + 4
    A = torch.randn(M, lda, device="cuda")
    Bt = torch.randn(N, ldb, device="cuda")
    C = torch.randn(M, ldc, device="cuda")
    C0 = C.clone()
    _native.ganet_check(lib.ganet_rowgemm(M, N, K, _ptr(A), lda, _ptr(Bt), ldb, _ptr(C), ldc, int(acc), _stream()))
    ref = A[:, :K].double() @ Bt.double().t() + (C0[:, :N].double() if acc else 0)
    err = float((C[:, :N].double() - ref).abs().max())
    assert err <= 2e-6 * float(ref.abs().max()) * (K ** 0.5), err
    assert torch.equal(C[:, N:], C0[:, N:])                     # columns beyond N untouched


@pytest.mark.parametrize("b,feat,S,shared_uv", [(1, 128, 512, False), (2, 16, 64, False), (3, 8, 32, True)])
def test_upsample_z_fwd_matches_float64(b, feat, S, shared_uv):
    """z = bilinear(P[:, :128]) + Wuv uv + bias and its shifted column sums against the float64 restatement of
    F.grid_sample(bilinear, align_corners=False) at the texel grid (the dense tap matrices are themselves checked
    against grid_sample in tests/test_fused_gpu.py::test_separable_bilinear_matmul_equals_grid_sample)."""
    from gaussianavatar_amd import _native
    lib = _native.ganet()
    torch.manual_seed(S)
    net, uv, mats, taps, grid = _grid_setup(b, feat, S, shared_uv)
    M = b * S * S
    P = torch.randn(b * feat * feat, 256, device="cuda")
    Wuv = torch.randn(128, 2, device="cuda")
    bias = torch.randn(128, device="cuda")
    shift = torch.randn(128, device="cuda") * 0.1
    z = torch.empty(M, 128, device="cuda")
    cp = torch.full((lib.ganet_mlp_stats_floats(128),), float("nan"), device="cuda")
    _native.ganet_check(lib.ganet_upsample_z_fwd(ctypes.byref(grid), _ptr(P), 256, _ptr(Wuv), _ptr(bias), _ptr(shift),
                                                 _ptr(z), _ptr(cp), _stream()))
    uvf = uv.reshape(M, 2).double()
    ref = _upsample64(mats, P[:, :128].double(), b, feat, S) + uvf @ Wuv.double().t() + bias.double()
    assert float((z.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    sums = cp.view(-1, 2, 128).double().sum(0)
    d = ref - shift.double()
    assert float((sums[0] - d.sum(0)).abs().max()) <= 1e-5 * float(d.abs().sum(0).max())
    assert float((sums[1] - (d * d).sum(0)).abs().max()) <= 1e-5 * float((d * d).sum(0).max())


@pytest.mark.parametrize("row_order", [0, 2])
@pytest.mark.parametrize("b,feat,S", [(1, 128, 512), (2, 16, 64), (1, 8, 32)])
def test_mlp_fwd_add_matches_float64(b, feat, S, row_order):
    """The skip layer as a hidden-layer launch with the gathered input half: act(bn(z4)) W5y^T + b + bilinear(P5) + W5uv uv."""
    from gaussianavatar_amd import _native
    lib = _native.ganet()
    torch.manual_seed(S + row_order)
    net, uv, mats, taps, grid = _grid_setup(b, feat, S)
    M = b * S * S
    P = torch.randn(b * feat * feat, 256, device="cuda")
    x2 = torch.randn(M, 128, device="cuda") * 2
    sc = torch.rand(128, device="cuda") + 0.5
    sh = torch.randn(128, device="cuda") * 0.3
    W = torch.randn(128, 128, device="cuda") * 0.1
    bias = torch.randn(128, device="cuda")
    Wuv = torch.randn(128, 2, device="cuda")
    shift = torch.randn(128, device="cuda") * 0.1
    z = torch.empty(M, 128, device="cuda")
    cp = torch.full((lib.ganet_mlp_stats_floats(128),), float("nan"), device="cuda")
    _native.ganet_check(lib.ganet_mlp_fwd_add(ctypes.byref(grid), _ptr(x2), _ptr(sc), _ptr(sh), _ptr(W), _ptr(bias),
                                              _ptr(P[:, 128:]), 256, _ptr(Wuv), _ptr(z), _ptr(cp), _ptr(shift), row_order,
                                              _stream()))
    a = F.softplus(x2.double() * sc.double() + sh.double())
    ref = (a @ W.double().t() + bias.double() + _upsample64(mats, P[:, 128:].double(), b, feat, S)
           + uv.reshape(M, 2).double() @ Wuv.double().t())
    err = (z.double() - ref).abs()
    bar = 2e-6 * float(ref.abs().max()) * 4
    bad = torch.nonzero(~(err <= bar))
    assert bad.numel() == 0, (f"{bad.shape[0]} elements beyond {bar:.2e} (max {float(err.max()):.3e}): slabs "
                              f"{torch.unique(bad[:, 0] // 32)[:8].tolist()}, rows in slab {torch.unique(bad[:, 0] % 32)[:32].tolist()}, "
                              f"columns {torch.unique(bad[:, 1])[:32].tolist()}")
    sums = cp.view(-1, 2, 128).double().sum(0)
    d = ref - shift.double()
    assert float((sums[0] - d.sum(0)).abs().max()) <= 1e-5 * float(d.abs().sum(0).max())
    assert float((sums[1] - (d * d).sum(0)).abs().max()) <= 1e-5 * float((d * d).sum(0).max())


@pytest.mark.parametrize("b,feat,S", [(1, 128, 512), (2, 16, 64), (1, 8, 32), (1, 6, 24)])
def test_dz_upsample_t_matches_float64(b, feat, S):
    """dP = (bilinear up-sampling)^T dz with dz = A G + q Z + p assembled on load, plus the sums dz uv^T and dz over all
    texels (finished by ganet_wgrad_reduce_batch), against float64."""
    from gaussianavatar_amd import _native
    lib = _native.ganet()
    torch.manual_seed(S)
    net, uv, mats, taps, grid = _grid_setup(b, feat, S)
    M = b * S * S
    G = torch.randn(M, 128, device="cuda")
    Z = torch.randn(M, 128, device="cuda")
    coef = torch.randn(3, 128, device="cuda")
    dP = torch.full((b * feat * feat, 256), float("nan"), device="cuda")
    nparts = lib.ganet_dz_upsample_t_parts(ctypes.byref(grid))
    part = torch.full((nparts, 128 * 2 + 128), float("nan"), device="cuda")
    _native.ganet_check(lib.ganet_dz_upsample_t(ctypes.byref(grid), _ptr(G), _ptr(Z), _ptr(coef), _ptr(dP[:, 128:]), 256,
                                                _ptr(part), _stream()))
    dz = coef[0].double() * G.double() + coef[1].double() * Z.double() + coef[2].double()
    Wr, Wc = (m.double() for m in mats)
    ref = torch.einsum("ip,jq,bijc->bpqc", Wr, Wc, dz.reshape(b, S, S, 128)).reshape(b * feat * feat, 128)
    assert float((dP[:, 128:].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * 8
    assert torch.isnan(dP[:, :128]).all()                        # the other half of the buffer untouched
    # the partial sums through the library's own batched reduction
    dWuv = torch.empty(128, 2, device="cuda")
    db = torch.empty(128, device="cuda")
    jobs = (_native.GanetWgradJob * 1)()
    jobs[0].workspace, jobs[0].M, jobs[0].N, jobs[0].K = part.data_ptr(), M, 128, 2
    jobs[0].dW, jobs[0].db, jobs[0].nblocks = dWuv.data_ptr(), db.data_ptr(), nparts
    _native.ganet_check(lib.ganet_wgrad_reduce_batch(1, jobs, _stream()))
    ref_w = dz.t() @ uv.reshape(M, 2).double()
    ref_b = dz.sum(0)
    scale = float(dz.abs().sum(0).max())
    assert float((dWuv.double() - ref_w).abs().max()) <= 2e-6 * scale
    assert float((db.double() - ref_b).abs().max()) <= 2e-6 * scale


def _bn_spread(net):
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.5, 0.5)


@pytest.mark.parametrize("b,feat,S,stage2", [(1, 32, 128, False), (2, 16, 64, True), (1, 128, 512, False)])
def test_decoder_from_map_equals_upsampled_input_path(b, feat, S, stage2, monkeypatch):
    """POP_no_unet.forward_points through the map path (one native call each way, no x [M,72]) against the path it
    replaces (ganet_upsample_cat + the decoder on the up-sampled rows): outputs, every parameter gradient, the feature
    map's gradient, the BatchNorm running statistics. stage2: per-frame pose features (b frames of decoder rows)."""
    import copy
    from gaussianavatar_amd import fused
    from gaussianavatar_amd.network import POP_no_unet
    torch.manual_seed(feat)
    net_a = POP_no_unet(c_geom=64, hsize=128).cuda().train()
    _bn_spread(net_a)
    net_b = copy.deepcopy(net_a)
    idx = torch.stack(torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij"), -1).reshape(-1, 2).float() / (S - 1)
    uv = idx.cuda()[None]
    geo = (torch.randn(1, 64, feat, feat, device="cuda") * 0.5).contiguous(memory_format=torch.channels_last)
    pose = torch.randn(b, 64, feat, feat, device="cuda") * 0.3 if stage2 else None
    B = b if stage2 else 2
    w = [torch.randn(b * S * S, k, device="cuda") for k in (3, 1, 3)]
    results = []
    calls = []
    real = fused.decoder_from_map
    monkeypatch.setattr(fused, "decoder_from_map", lambda *a: (calls.append(1), real(*a))[1])
    for net, use_map in ((net_a, True), (net_b, False)):
        monkeypatch.setattr(fused, "_DECODER_MAP", use_map)
        g = geo.clone().requires_grad_(True)
        pf = None if pose is None else pose.clone().requires_grad_(True)
        outs = net.forward_points(pf, g.expand(B, -1, -1, -1), uv.expand(B, -1, -1), raw_heads=True)
        loss = sum((o[:b].reshape(b * S * S, -1) * wi).sum() for o, wi in zip(outs, w))
        loss.backward()
        results.append((outs, g.grad, None if pf is None else pf.grad))
    assert len(calls) == 1, "the map path was not taken exactly once"
    (oa, ga, pa), (ob, gb, pb) = results
    for x, y in zip(oa, ob):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=2e-5)
    pairs = [("geo", ga, gb)] + ([("pose", pa, pb)] if stage2 else [])
    pairs += [(n, p.grad, q.grad) for (n, p), (_, q) in zip(net_a.named_parameters(), net_b.named_parameters())]
    assert_grads_close(pairs)
    for (n, p), (_, q) in zip(net_a.named_buffers(), net_b.named_buffers()):
        torch.testing.assert_close(p.float(), q.float(), rtol=1e-4, atol=1e-5, msg=n)

"""GPU parity of the fused net/loss kernels (include/ganet.h) against plain torch fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

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
    gmax = max(float(p.grad.abs().max()) for p in net.parameters())
    assert float((geo.grad - geo_g.grad.cpu()).abs().max()) <= 2e-3 * max(1.0, float(geo.grad.abs().max()))
    for (n, p), (_, q) in zip(net.named_parameters(), net_g.named_parameters()):
        assert float((p.grad - q.grad.cpu()).abs().max()) <= 2e-3 * max(1.0, gmax), n


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
    gmax = max(float(p.grad.abs().max()) for p in dec.parameters())
    for (n, p), (_, q) in zip(dec.named_parameters(), dec_g.named_parameters()):
        assert float((p.grad - q.grad.cpu()).abs().max()) <= 2e-3 * max(1.0, gmax), n

"""GPU parity of the hand-written stage-2 pose encoder (csrc/ganet_unet.hip: ganet_unet_fwd / _bwd) against the same
module in float64 on the CPU (network.UnetNoCond5DS, itself pinned to the reference's UnetNoCond5DS by
tests/golden/net_golden.npz and net_full_golden.npz; /root/reference/model/modules.py:185-232)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(B, S, nf, cout, seed):
    from gaussianavatar_amd.network import UnetNoCond5DS
    torch.manual_seed(seed)
    enc = UnetNoCond5DS(input_nc=3, output_nc=cout, nf=nf).train()
    with torch.no_grad():                 # running statistics away from their initial values
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(B, 3, S, S) * 0.3
    return enc, x


@pytest.mark.parametrize("B,S,nf,cout", [(2, 128, 32, 64), (1, 64, 32, 64), (3, 32, 32, 32), (2, 64, 64, 32)])
def test_unet_forward_backward_match_float64(B, S, nf, cout):
    from gaussianavatar_amd import fused
    enc, x = _pair(B, S, nf, cout, seed=S + B)
    e64 = copy.deepcopy(enc).double()
    eg = copy.deepcopy(enc).cuda()
    xg = x.cuda()
    assert fused.unet_supported(eg, xg)
    w = torch.randn(B, cout, S, S)
    y64 = e64(x.double())
    (y64 * w.double()).sum().backward()
    yg = eg(xg)
    assert yg.shape == (B, cout, S, S) and yg.permute(0, 2, 3, 1).is_contiguous()       # channels-last underneath
    (yg * w.cuda()).sum().backward()
    err = float((yg.detach().cpu().double() - y64.detach()).abs().max() / y64.detach().abs().max())
    assert err <= 2e-5, err
    # The net is piecewise linear (LeakyReLU / ReLU): a pre-activation within float32 rounding of zero takes the other
    # branch and moves ONE element's gradient by its full value (tests/test_assembled_headline_gpu.py). Per tensor:
    # relative L2 error <= 3e-3, and all but 1 % of the elements within 1e-3 of the tensor's maximum.
    for (n, p64), (_, pg) in zip(e64.named_parameters(), eg.named_parameters()):
        a, b = pg.grad.detach().cpu().double().reshape(-1), p64.grad.reshape(-1)
        l2 = float((a - b).norm() / (b.norm() + 1e-300))
        assert l2 <= (3e-3 if B * (S // 16) ** 2 >= 64 else 2e-2), (n, "relative L2", l2)
        if B * (S // 16) ** 2 >= 64:      # (BatchNorm over a handful of samples amplifies a flipped gate into every element)
            frac = float(((a - b).abs() > 1e-3 * b.abs().max()).double().mean())
            assert frac <= 1e-2, (n, "elements off", frac)
    # BatchNorm running statistics (unbiased variance, momentum 0.1), num_batches_tracked
    for (n, b64), (_, bg) in zip(e64.named_buffers(), eg.named_buffers()):
        torch.testing.assert_close(bg.cpu().double(), b64.double(), rtol=1e-4, atol=1e-5, msg=n)


def test_unet_native_equals_the_torch_formulation_on_the_device(monkeypatch):
    """the path it replaces (im2col + vendor GEMM + torch BatchNorm), same device, same float32"""
    from gaussianavatar_amd import fused
    enc, x = _pair(2, 128, 32, 64, seed=11)
    ea, eb = copy.deepcopy(enc).cuda(), copy.deepcopy(enc).cuda()
    xg = x.cuda()
    w = torch.randn(2, 64, 128, 128, device="cuda")
    ya = ea(xg)
    (ya * w).sum().backward()
    monkeypatch.setattr(fused, "_NATIVE_UNET", False)
    yb = eb(xg)
    (yb * w).sum().backward()
    torch.testing.assert_close(ya, yb, rtol=1e-4, atol=1e-5)
    for (n, pa), (_, pb) in zip(ea.named_parameters(), eb.named_parameters()):
        l2 = float((pa.grad - pb.grad).norm() / (pb.grad.norm() + 1e-30))
        assert l2 <= 3e-3, (n, l2)


def test_unet_eval_mode_uses_running_statistics():
    from gaussianavatar_amd import fused
    enc, x = _pair(2, 64, 32, 64, seed=5)
    enc.eval()
    eg = copy.deepcopy(enc).cuda().eval()
    with torch.no_grad():
        assert fused.unet_supported(eg, x.cuda())
        y = eg(x.cuda())
        ref = copy.deepcopy(enc).double()(x.double())
    assert float((y.cpu().double() - ref).abs().max() / ref.abs().max()) <= 2e-5
    for (n, b0), (_, b1) in zip(enc.named_buffers(), eg.named_buffers()):
        assert torch.equal(b0, b1.cpu()), n                      # nothing updated

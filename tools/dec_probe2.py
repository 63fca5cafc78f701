import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from gaussianavatar_amd.avatar_model import AvatarModel, default_params
def bench(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
mp, npar, op = default_params(batch_size=2, num_points=200000)
m = AvatarModel(mp, npar, op, train=True); m.training_setup()
dec = m.net.decoder
M = 262144
x = torch.randn(M, 66, device='cuda')
with torch.no_grad():
    print('decoder(randn) us', bench(lambda: dec.forward_points(x)))
    geom = m.geo_feature.expand(2, -1, 128, 128); uv = m.uv_coord_map[None].expand(2, -1, -1)
    print('net.forward_points us', bench(lambda: m.net.forward_points(None, geom, uv)))
    g1 = m.net.geom_proc_layers(geom[:1])
    pix = F.grid_sample(g1, __import__('gaussianavatar_amd.network', fromlist=['uv_to_grid']).uv_to_grid(uv[:1], 512), mode='bilinear', align_corners=False)
    xx = torch.cat([pix.reshape(1, 64, M).transpose(1, 2), uv[:1]], dim=2).reshape(M, 66)
    print('x strides', xx.stride(), xx.is_contiguous(), xx.dtype, xx.data_ptr() % 256)
    print('decoder(real x) us', bench(lambda: dec.forward_points(xx)))
    c = dec.conv2; h = torch.randn(M, 128, device='cuda')
    print('conv2 linear us', bench(lambda: F.linear(h, c.weight.squeeze(-1), c.bias)), c.weight.stride(), c.weight.squeeze(-1).stride(), c.bias.stride(), c.weight.data_ptr() % 256, c.bias.data_ptr() % 256)
    w2 = c.weight.squeeze(-1).clone(); b2 = c.bias.clone()
    print('conv2 linear cloned params us', bench(lambda: F.linear(h, w2, b2)), w2.data_ptr() % 256, b2.data_ptr() % 256)

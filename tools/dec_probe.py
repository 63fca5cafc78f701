import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from gaussianavatar_amd.network import ShapeDecoder
def bench(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
M=262144
dec = ShapeDecoder(66, 128).cuda().train()
x = torch.randn(M, 66, device='cuda')
with torch.no_grad():
    print('decoder fwd no_grad us', bench(lambda: dec.forward_points(x)))
    c = dec.conv2; h = torch.randn(M,128,device='cuda')
    print('conv2 linear', bench(lambda: F.linear(h, c.weight.squeeze(-1), c.bias)))
    y = F.linear(h, c.weight.squeeze(-1), c.bias)
    sp = F.softplus(dec.bn2(y))
    print('linear on softplus(bn) output', bench(lambda: F.linear(sp, c.weight.squeeze(-1), c.bias)))
print('decoder fwd grad us', bench(lambda: dec.forward_points(x)))

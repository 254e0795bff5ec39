"""Micro-benchmark of pnl_gemv on a random symmetric matrix: the full product (8 N^2 bytes) against the two-sided product of the
upper triangle (symmetric_half = 2, 4 N^2 bytes).  usage: gemv_probe.py [N]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pynucleus_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48769
ld = (N+7) & ~7
ctx = _lib.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device='cuda'); g.manual_seed(1)
A = torch.empty((N, ld), dtype=torch.float64, device='cuda')
for i in range(0, N, 4096):
    A[i:i+4096, :N] = torch.rand((min(4096, N-i), N), dtype=torch.float64, device='cuda', generator=g)-0.5
# symmetrise in place block-wise: upper triangle wins
for i in range(0, N, 4096):
    for j in range(i, N, 4096):
        blk = A[i:i+4096, j:min(j+4096, N)]
        if i == j:
            blk.copy_(torch.triu(blk)+torch.triu(blk, 1).T)
        else:
            A[j:min(j+4096, N), i:i+4096] = blk.T
x = torch.rand(N, dtype=torch.float64, device='cuda', generator=g)
y0 = torch.empty(N, dtype=torch.float64, device='cuda'); y2 = torch.empty_like(y0)
ctx.gemv(A.data_ptr(), ld, N, x.data_ptr(), y0.data_ptr(), 0); ctx.synchronize()
ctx.gemv(A.data_ptr(), ld, N, x.data_ptr(), y2.data_ptr(), 2); ctx.synchronize()
print('max rel diff sym vs full', float((y0-y2).abs().max()/y0.abs().max()))
for mode in (0, 2):
    for _ in range(3): ctx.gemv(A.data_ptr(), ld, N, x.data_ptr(), y2.data_ptr(), mode)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ctx.gemv(A.data_ptr(), ld, N, x.data_ptr(), y2.data_ptr(), mode)
    ctx.synchronize(); dt = (time.perf_counter()-t0)/30
    print('mode', mode, 'ms', 1e3*dt, 'TB/s on', (8 if mode == 0 else 4), 'N^2:', (8 if mode == 0 else 4)*N*N/dt/1e12)

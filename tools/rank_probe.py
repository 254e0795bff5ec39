#!/usr/bin/env python3
"""Time the share of every rank of a P-rank row-owned dense assembly on ONE GPU, one rank after the other (load balance of
builder.row_slab_of_rank without a multi-GPU node): tools/rank_probe.py [P] [noRef]"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder, row_slab_of_rank, tile_cells

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
noRef = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dm = P1_DoFMap(disc(noRef), PHYSICAL)
b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5}, zeroExterior=True)
ctx = b.context()
dev = torch.device('cuda', ctx.device)
T = tile_cells(b.dm.dofs_per_element, 2)
tot = 0.
for r in range(P):
    costs = None if os.environ.get('PNL_EQUAL_TILES') else ctx.block_row_costs((b.dm.mesh.num_cells+T-1)//T)
    c0, c1, tiles, rows, cols = row_slab_of_rank(b.dm, T, r, P, costs)
    A = torch.zeros((max(rows.shape[0], 1), max(cols.shape[0], 1)), dtype=torch.float64, device=dev)
    ctx.set_row_slab(rows, cols)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        A.zero_()
        ctx.assemble_dense_tiles(A.data_ptr(), A.stride(0), True, tiles, c0, c1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    ms = ctx.phase_ms()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for rep in range(5):                                  # back to back like bench.py: host work of step k+1 behind the GPU work of step k
        A.zero_()
        ctx.assemble_dense_tiles(A.data_ptr(), A.stride(0), True, tiles, c0, c1)
    t3h = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print('   pipelined: {:.2f} ms/step (host {:.2f} ms/step); kernel ms {}'.format(1e3*(t3-t2)/5, 1e3*(t3h-t2)/5, {k: round(v, 2) for k, v in ctx.kernel_ms().items()}), flush=True)
    print('rank {}: cells [{}, {}) tiles {} slab {} x {} ({:.2f} GB): step {:.2f} ms, phases {}'.format(
        r, c0, c1, tiles.shape[0], rows.shape[0], cols.shape[0], A.numel()*8/1e9, 1e3*(t1-t0), {k: round(v, 2) for k, v in ms.items()}), flush=True)
    tot += t1-t0
    del A
print('sum over ranks {:.1f} ms'.format(1e3*tot))

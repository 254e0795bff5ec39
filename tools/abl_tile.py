"""Ablation of the mixed-tile kernel on the headline operator (debug build of the library only:
make BUILD=.../build_abl OUT=.../libpnl_abl.so EXTRA="-DPNL_DEBUG_ABLATE -DPNL_TUNING"; run with PNL_LIB=<that .so> PNL_ABLATE=<bits>).
Bits of k_tile_distant: 1 no accumulation, 2 no evaluation, 4 no flush, 8 no pair passes the classification, 16 every pair order 2,
64 no diagonal blocks.  The matrix of an ablated run is wrong by construction; only the kernel times are read."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynucleus_amd import disc, PHYSICAL, P1_DoFMap, getFractionalKernel  # noqa: E402
from pynucleus_amd.builder import nonlocalBuilder  # noqa: E402

noRef = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dm = P1_DoFMap(disc(noRef), PHYSICAL)
b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5}, zeroExterior=True)
tot, gen = [], []
for rep in range(4):
    A = b.getDense()
    torch.cuda.synchronize()
    if rep:
        tot.append(A.info['phase_ms']['total'])
        gen.append(b.dense_context().kernel_ms()['tile_general'])
        A0 = A.info['phase_ms']
    del A
print('PNL_ABLATE', os.environ.get('PNL_ABLATE', '0'), 'total_ms %.2f' % float(np.median(tot)), 'tile_general_ms %.2f' % float(np.median(gen)))
print('phases', {k: round(v, 2) for k, v in A0.items()}, 'kernels', {k: round(v, 2) for k, v in b.dense_context().kernel_ms().items()})
if os.environ.get('PNL_VERBOSE'):
    A = b.getDense()
    print('orders', A.info['counters']['orders'], 'pairs', A.info['counters']['numAssembledCellPairs'])

#!/usr/bin/env python3
"""How conservative is the host's uniform-tile classification?  Exact per-tile order ranges (numpy, all cell pairs) against the
counts the library prints with PNL_VERBOSE.  tools/tile_exact.py [noRef]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['PNL_VERBOSE'] = '1'
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
noRef = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dm = P1_DoFMap(disc(noRef), PHYSICAL)
b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5})
A = b.getDense()
T = b.tables
F = T.qo
mesh = b.mesh
cen = mesh.vertices[mesh.cells].mean(axis=1)
h = mesh.hVector
H0 = T.H0
nc = mesh.num_cells
TILE = 64
nb = (nc+TILE-1)//TILE
L = np.abs(np.log(h/H0))
cnt = {'uniform2': 0, 'uniform3': 0, 'uniform4': 0, 'mixed<=4': 0, 'mixed_with_wl': 0, 'touching': 0}
for a in range(nb):
    ia = slice(a*TILE, min(nc, (a+1)*TILE))
    for bb in range(a, nb):
        ib = slice(bb*TILE, min(nc, (bb+1)*TILE))
        d = np.sqrt(((cen[ia][:, None, :]-cen[ib][None, :, :])**2).sum(-1))
        h1, h2 = h[ia][:, None], h[ib][None, :]
        L1, L2 = L[ia][:, None], L[ib][None, :]
        with np.errstate(divide='ignore'):
            l1, l2 = np.log(d/h1), np.log(d/h2)
        Lm = np.maximum(L1, L2)
        n1, n2 = (np.maximum(l1, 0.), np.maximum(l2, 0.)) if F.clip_num else (l1, l2)
        p1 = np.ceil((F.c0+F.a*L2+F.b*Lm-F.e*n2)/(np.maximum(l1, 0.)+F.den0))
        p2 = np.ceil((F.c0+F.a*L1+F.b*Lm-F.e*n1)/(np.maximum(l2, 0.)+F.den0))
        q = np.maximum(np.maximum(p1, 2.), np.maximum(p2, 2.))
        # touching pairs (shared vertices) make a tile non-uniform in any case
        va, vb = mesh.cells[ia], mesh.cells[ib]
        touch = (va[:, None, :, None] == vb[None, :, None, :]).any(axis=(2, 3)).any()
        if touch:
            cnt['touching'] += 1
        elif q.min() == q.max() and q.max() <= 4:
            cnt['uniform%d' % int(q.max())] += 1
        elif q.max() <= 4:
            cnt['mixed<=4'] += 1
        else:
            cnt['mixed_with_wl'] += 1
print('exact:', cnt, 'tiles', nb*(nb+1)//2)

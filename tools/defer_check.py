"""A/B check of the deferred (cell, facet) pairs of the cluster-local boundary term (k_cluster_boundary -> k_boundary_items):
run with a tuning build (PNL_LIB=.../build_tune/libpnl_tune.so) and PNL_CB_DEFER=0 / 200 / 8, compare the saved near-field data.
Usage: python tools/defer_check.py <noRef> <out.npy>.  Measured (round 3): max difference / max entry 6e-16 at noRef 5 and 7, the
boundary pair / integration counters identical."""
import sys, os, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
dm = P1_DoFMap(disc(int(sys.argv[1])), PHYSICAL)
b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
h2 = b.getH2()
torch.cuda.synchronize()
A = h2.Anear
d = A.data_t.cpu().numpy() if hasattr(A, 'data_t') else np.asarray(A.data)
c = A.info['counters']
np.save(sys.argv[2], d)
print('nnz', d.shape, 'boundary pairs', c['numBoundaryPairs'], 'integrations', c['numBoundaryIntegrations'], 'sum', float(d.sum()), 'abs', float(np.abs(d).sum()))

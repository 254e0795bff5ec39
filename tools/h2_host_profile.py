#!/usr/bin/env python3
"""Host-side cost of getH2 (tree, admissibility, near-field plan, pattern, far-field plan); no GPU needed.  usage: h2_host_profile.py [noRef]"""
import sys
import time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL
from pynucleus_amd import clusters
from pynucleus_amd.builder import nonlocalBuilder
from pynucleus_amd.h2 import h2Plan

noRef = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mesh = disc(noRef)
dm = P1_DoFMap(mesh, PHYSICAL)
b = nonlocalBuilder.__new__(nonlocalBuilder)
b.dm, b.mesh, b.params = dm, mesh, {}
rp = nonlocalBuilder.getH2RefinementParams(b)
t0 = time.time()
root, Pnear, Pfar = clusters.getNearFieldClusters(dm, rp['eta'], rp['minSize'], rp['maxLevels'])
t1 = time.time()
plan = clusters.nearFieldPlan(dm, Pnear)
t2 = time.time()
ip, ix = clusters.getSparseNearField(dm, Pnear, symmetric=False)[:2]
t3 = time.time()
hp = h2Plan(dm, root, Pfar, 7)
t4 = time.time()
print('noRef {} N {}: tree+admissibility {:.2f} s ({} near, {} far), near-field plan {:.2f} s, pattern {:.2f} s (nnz {}), far-field plan {:.2f} s'.format(
    noRef, dm.num_dofs, t1-t0, len(Pnear), sum(len(v) for v in Pfar.values()), t2-t1, t3-t2, ix.shape[0], t4-t3))

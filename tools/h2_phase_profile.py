#!/usr/bin/env python3
"""Where a getH2 of a NEW builder spends its wall time (disc noRef 7, s = 0.75): cProfile of the host side around the device work.
usage: h2_phase_profile.py"""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel, _lib
from pynucleus_amd.builder import nonlocalBuilder
dm = P1_DoFMap(disc(7), PHYSICAL)
for rep in range(3):
    t00 = time.perf_counter()
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
    b.context()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    h2 = b.getH2()
    torch.cuda.synchronize()
    pr.disable(); t1 = time.perf_counter()
    print('rep %d: builder + context %.1f ms, getH2 %.1f ms' % (rep, 1e3*(t0-t00), 1e3*(t1-t0)), flush=True)
    if rep == 2:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])

#!/usr/bin/env python3
"""getH2 with a NEW builder per call (nothing cached) at disc noRef 7: wall time and the planner's own timing lines
(option PNL_PLAN_TIMING).  usage: h2_cold_profile.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel, _lib
from pynucleus_amd.builder import nonlocalBuilder
_lib.set_option('PNL_PLAN_TIMING', '1')
dm = P1_DoFMap(disc(7), PHYSICAL)
for rep in range(3):
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
    b.context()
    torch.cuda.synchronize(); t0 = time.perf_counter(); h2 = b.getH2(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print('new builder: getH2 %.1f ms, near field device %.2f ms' % (1e3*(t1-t0), h2.Anear.info['interior_ms']), flush=True)

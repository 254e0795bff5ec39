#!/usr/bin/env python3
"""Turn a tools/profile_legs.sh output directory (gpurun_out/<tag>_legs) into tracked artefacts under profiles/:
  <tag>_<leg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of tools/leg_probe.py <leg>
  <tag>_<leg>_counters.json      the leg's result line, mean SQ counters per dispatch and kernel, and derived per kernel:
                                 VALU issue utilisation (SQ_INSTS_VALU x 4 / SQ_BUSY-normalised SIMD cycles), VALU busy, wait fraction
usage: tools/collect_legs.py <tag> [legs...]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag+'_legs')
dst = os.path.join(root, 'profiles')
legs = sys.argv[2:] or sorted({os.path.basename(p).split('_stats')[0] for p in glob.glob(os.path.join(src, '*_stats.log'))})


def short(name):
    head = name.split('(')[0]
    n = head.split('<')[0].split()[-1]
    if '<' in head and n.startswith('k_'):
        args = [a.strip() for a in head.split('<', 1)[1].rstrip('>').split(',')]
        return n+'<'+','.join(args)+'>'
    return n


for leg in legs:
    stats = sorted(glob.glob(os.path.join(src, leg+'_stats', '*', '*kernel_stats.csv')), key=os.path.getmtime)[-1:]
    if stats:
        shutil.copy(stats[0], os.path.join(dst, '{}_{}_kernel_stats.csv'.format(tag, leg)))
    out = {'leg': leg}
    log = os.path.join(src, leg+'_stats.log')
    if os.path.exists(log):
        for line in open(log):
            if line.startswith('LEG '):
                name, js = line[4:].split(' ', 1)
                out['name'], out['result'] = name, json.loads(js)
    avg_ns = {}
    if stats:
        for r in csv.DictReader(open(stats[0])):
            avg_ns[short(r['Name'])] = float(r['AverageNs'])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for fn in sorted(glob.glob(os.path.join(src, leg+'_sq', '*', '*counter_collection.csv')), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(fn)):
            k = short(r['Kernel_Name'])
            if not k.startswith('k_'):
                continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[k].add(r['Dispatch_Id'])
    kernels = {}
    for k in agg:
        c = {n: v/max(1, len(calls[k])) for n, v in agg[k].items()}
        d = dict(counters=c, dispatches=len(calls[k]))
        if k in avg_ns:
            d['avg_us'] = avg_ns[k]/1e3
            if 'SQ_INSTS_VALU' in c:
                # 256 CUs x 4 SIMDs, a wave64 VALU instruction occupies its SIMD for 4 cycles, 2.4 GHz
                d['valu_issue_util'] = c['SQ_INSTS_VALU']*4./(1024.*avg_ns[k]*2.4)
        if c.get('SQ_WAVE_CYCLES'):
            d['wait_frac'] = c.get('SQ_WAIT_ANY', 0.)/c['SQ_WAVE_CYCLES']
        if c.get('SQ_BUSY_CYCLES') and 'SQ_ACTIVE_INST_VALU' in c:
            d['valu_busy'] = c['SQ_ACTIVE_INST_VALU']/c['SQ_BUSY_CYCLES']/4.
        kernels[k] = d
    out['kernels'] = kernels
    with open(os.path.join(dst, '{}_{}_counters.json'.format(tag, leg)), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(leg, out.get('name'), {k: round(v.get('valu_issue_util', 0.), 3) for k, v in kernels.items() if v.get('avg_us', 0) > 500})

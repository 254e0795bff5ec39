#!/usr/bin/env python3
"""gpurun_out/<tag>_p2 (tools/profile_p2.sh) -> profiles/<tag>_{p2,c5}_noRef6_kernel_stats.csv and profiles/<tag>_{p2,c5}_noRef6_counters.json
(mean counter value per dispatch and kernel, separate --pmc passes; derived: VALU-busy estimate and waves per SIMD).
usage: tools/collect_p2_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag+'_p2')
dst = os.path.join(root, 'profiles')
CLOCK_GHZ, SIMDS = 2.4, 1024


def newest(pattern):
    """the most recent match only: gpurun merges new output into gpurun_out/ next to the files of earlier runs"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] 


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


for cfg in ('p2', 'c5'):
    stats = newest(os.path.join(src, cfg+'_stats', '*', '*kernel_stats.csv'))
    avg_ns = {}
    if stats:
        shutil.copy(stats[0], os.path.join(dst, '{}_{}_noRef6_kernel_stats.csv'.format(tag, cfg)))
        for r in csv.DictReader(open(stats[0])):
            avg_ns[short(r['Name'])] = (float(r['AverageNs']), int(r['Calls']))
    summary = collections.defaultdict(dict)
    for d in ('sq', 'grbm', 'fetch', 'write'):
        for fn in newest(os.path.join(src, '{}_{}'.format(cfg, d), '*', '*counter_collection.csv')):
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            calls = collections.defaultdict(set)
            for r in csv.DictReader(open(fn)):
                k = short(r['Kernel_Name'])
                if k.startswith('at::') or 'rocclr' in k or 'vectorized' in k:
                    continue
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                calls[k].add(r['Dispatch_Id'])
            for k in agg:
                for c, v in agg[k].items():
                    summary[k][c] = v/max(1, len(calls[k]))
    for k, v in summary.items():
        if k in avg_ns:
            v['avg_duration_ns'], v['calls'] = avg_ns[k]
            cyc = v['avg_duration_ns']*CLOCK_GHZ
            if 'SQ_ACTIVE_INST_VALU' in v:
                # a wave64 VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs; 2.4 GHz assumed
                v['valu_busy_estimate'] = v['SQ_ACTIVE_INST_VALU']*4./(SIMDS*cyc)
            if 'SQ_WAVE_CYCLES' in v:
                v['mean_waves_per_simd'] = v['SQ_WAVE_CYCLES']*4./(SIMDS*cyc)        # the SQ cycle counters tick every 4 cycles
    with open(os.path.join(dst, '{}_{}_noRef6_counters.json'.format(tag, cfg)), 'w') as f:
        json.dump({'comment': 'mean per dispatch; valu_busy_estimate = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * duration * 2.4 GHz); '
                              'mean_waves_per_simd = SQ_WAVE_CYCLES * 4 / (1024 * duration * 2.4 GHz) (SQ cycle counters tick every 4 cycles: a 512-thread workgroup per CU gives 2.0); FETCH_SIZE / WRITE_SIZE in KB',
                   'kernels': summary}, f, indent=1, sort_keys=True)
    for k in sorted(summary, key=lambda k: -summary[k].get('avg_duration_ns', 0))[:8]:
        v = summary[k]
        print(cfg, k[:60], 'ms', round(v.get('avg_duration_ns', 0)/1e6, 3), 'valu_busy', round(v.get('valu_busy_estimate', 0), 3),
              'waves/simd', round(v.get('mean_waves_per_simd', 0), 2))

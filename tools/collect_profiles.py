#!/usr/bin/env python3
"""Turn one tools/profile_round.sh output directory (gpurun_out/<tag>) into the tracked artefacts under profiles/:
  <tag>_bench_noRef<N>.json            the bench line
  <tag>_bench_noRef<N>_kernel_stats.csv rocprofv3 --kernel-trace --stats of the same command
  <tag>_pmc_summary_noRef<N>.json      mean counter value per dispatch and kernel (separate --pmc passes)
  pmc_traffic.json                     HBM bytes per launch of the tile kernels, corrected as MI355X_MICROARCH.md prescribes
usage: tools/collect_profiles.py <tag> <noRef> [sectors]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, noRef = sys.argv[1], sys.argv[2]
sectors = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sfx = '' if sectors == 6 else '_s{}'.format(sectors)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag+sfx)
dst = os.path.join(root, 'profiles')
shutil.copy(os.path.join(src, 'bench_noRef{}{}.json'.format(noRef, sfx)), os.path.join(dst, '{}_bench_noRef{}{}.json'.format(tag, noRef, sfx)))
stats = sorted(glob.glob(os.path.join(src, 'stats', '*', '*kernel_stats.csv')), key=os.path.getmtime)[-1:]
if stats:
    shutil.copy(stats[0], os.path.join(dst, '{}_bench_noRef{}{}_kernel_stats.csv'.format(tag, noRef, sfx)))


def newest(pattern):
    """the most recent match only: gpurun merges new output into gpurun_out/ next to the files of earlier runs"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] 


def short(name):
    head = name.split('(')[0]
    parts = head.split('<')[0].split()
    n = parts[-1] if parts else name
    if n == 'k_tile_uniform' and '<' in head:
        # the instantiations are different kernels of the step: <DPE, NP, KT> -> k_tile_uniform_<DPE>_<NP>
        args = [a.strip() for a in head.split('<', 1)[1].rstrip('>').split(',')]
        if len(args) >= 2:
            return 'k_tile_uniform_{}_{}'.format(args[0], args[1])
    return n


summary = collections.defaultdict(dict)
for d in ('pmc_fetch', 'pmc_write', 'pmc_tcc', 'pmc_sq'):
    for fn in newest(os.path.join(src, d, '*', '*counter_collection.csv')):
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
with open(os.path.join(dst, '{}_pmc_summary_noRef{}{}.json'.format(tag, noRef, sfx)), 'w') as f:
    json.dump(summary, f, indent=1, sort_keys=True)

traffic_fn = os.path.join(dst, 'pmc_traffic.json')
rec = {}
if os.path.exists(traffic_fn):
    rec = json.load(open(traffic_fn))
rec['comment'] = ('HBM traffic per launch of the tile kernels from rocprofv3 --pmc passes of bench.py (tools/profile_round.sh): '
                  'FETCH_SIZE and WRITE_SIZE in separate passes (KB units x 1024).  FETCH_SIZE doubled per MI355X_MICROARCH.md '
                  '(gfx950 reports half of wide coalesced reads; not for k_fold_mirror, whose 8-byte gathers are calibrated on their known volume); WRITE_SIZE: plain 8- and 16-byte stores into the block-slot storage.  '
                  'Raw values: profiles/<tag>_pmc_summary_noRef<N>.json.')
import hashlib
with open(os.path.join(root, 'pynucleus_amd', 'libpnl_hip.so'), 'rb') as f:
    lib_sha = hashlib.sha256(f.read()).hexdigest()[:16]
sys.path.insert(0, root)
from pynucleus_amd._lib import source_sha16
entry = {'tag': tag, 'lib_sha16': lib_sha, 'src_sha16': source_sha16()}
for k in ('k_tile_distant', 'k_tile_pure', 'k_tile_uniform_3_3', 'k_tile_uniform_3_6', 'k_tile_uniform_6_3', 'k_tile_uniform_6_6', 'k_tile_p2', 'k_fold_mirror', 'k_boundary_tile', 'k_gemv_two_sided'):
    if k in summary and 'FETCH_SIZE' in summary[k] and 'WRITE_SIZE' in summary[k]:
        # FETCH_SIZE x 2 for the tile kernels (their known input volume, ~13 KB of cell data per tile, matches the doubled value);
        # the fold pass reads every stored entry of the block-slot storage exactly once with 8-byte gathers -- a known byte count
        # (8 x slot entries, 20 GB at 48,769 DoFs) that the RAW counter reproduces to 25 %, so it is not doubled there
        fac = 1 if k == 'k_fold_mirror' else 2
        entry[k+'_hbm_bytes_per_launch'] = int(1024*(fac*summary[k]['FETCH_SIZE']+summary[k]['WRITE_SIZE']))
        entry[k+'_fetch_size_kb'] = summary[k]['FETCH_SIZE']
        entry[k+'_write_size_kb'] = summary[k]['WRITE_SIZE']
    if k in summary and 'SQ_INSTS_VALU' in summary[k]:
        entry[k+'_sq_insts_valu'] = summary[k]['SQ_INSTS_VALU']
rec['noRef{}{}'.format(noRef, sfx)] = entry
with open(traffic_fn, 'w') as f:
    json.dump(rec, f, indent=1)
print(json.dumps(entry, indent=1))

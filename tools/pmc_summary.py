#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for fn in glob.glob(d+'/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.defaultdict(set)
        for r in csv.DictReader(open(fn)):
            k = r['Kernel_Name'].split('(')[0][-40:]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[k].add(r['Dispatch_Id'])
        for k in agg:
            if k.startswith('void at::') or 'rocclr' in k:
                continue
            n = max(1, len(calls[k]))
            print(d, k, {c: '%.3e' % (v/n) for c, v in sorted(agg[k].items())})

#!/usr/bin/env python3
"""Average every counter of a rocprofv3 --pmc run per kernel: pmc_agg.py <dir> [kernel-substring]"""
import csv, glob, re, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('chsph::', '')
        if filt in n:
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in sorted(acc.items()):
    print(n, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "calls", max(len(v) for v in cs.values()))

#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: kernel stats table and per-kernel PMC averages."""
import csv, glob, re, sys, collections
def short(n): return re.sub(r'\(.*', '', n).replace('void ', '').replace('chsph::', '')
def kstats(d):
    f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    out = []
    for r in rows:
        out.append((short(r['Name']), int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    return out
def pmc(d):
    f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
if __name__ == '__main__':
    mode, d = sys.argv[1], sys.argv[2]
    if mode == 'kt':
        for n, c, t, a, p in kstats(d): print("%-46s calls=%6d total_ms=%9.3f avg_us=%9.2f pct=%6.2f" % (n[:46], c, t, a, p))
    else:
        filt = sys.argv[3] if len(sys.argv) > 3 else ''
        for k, cs in pmc(d).items():
            if filt in k: print(k[:60], {c: round(v, 1) for c, v in cs.items()})

#!/usr/bin/env python3
"""Turn one tools/collect_profiles.sh directory into a markdown table: per kernel calls, average duration, and HBM traffic
per launch from the PMC passes.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request
for wide coalesced reads, so it is doubled (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is taken as is."""
import csv, glob, json, os, re, sys, collections


def short(n):
    return re.sub(r'\(.*', '', n).replace('void ', '').replace('chsph::', '')


def kstats(d):
    f = glob.glob(d + '/kt/**/*kernel_stats.csv', recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    return [(short(r['Name']), int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])) for r in rows]


def pmc(d, sub):
    f = glob.glob(d + '/' + sub + '/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    d = sys.argv[1]
    fetch, write, tcc = pmc(d, 'fetch'), pmc(d, 'write'), pmc(d, 'tcc')
    print("| kernel | calls | total ms | avg us | % | HBM read MB/launch (2x FETCH_SIZE) | HBM write MB/launch | L2 hit % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    summary = {}
    for n, c, t, a, p in kstats(d):
        fb = fetch.get(n, {}).get('FETCH_SIZE')
        wb = write.get(n, {}).get('WRITE_SIZE')
        h, m = tcc.get(n, {}).get('TCC_HIT_sum'), tcc.get(n, {}).get('TCC_MISS_sum')
        rd = None if fb is None else 2 * fb * 1024 / 1e6
        wr = None if wb is None else wb * 1024 / 1e6
        hit = None if not h and not m else 100.0 * h / max(h + m, 1)
        fmt = lambda v, f: '' if v is None else f % v
        print("| %s | %d | %.3f | %.2f | %.2f | %s | %s | %s |" % (n, c, t, a, p, fmt(rd, '%.1f'), fmt(wr, '%.1f'), fmt(hit, '%.1f')))
        summary[n] = dict(calls=c, avg_us=a, hbm_read_bytes=None if rd is None else rd * 1e6, hbm_write_bytes=None if wr is None else wr * 1e6, l2_hit_pct=hit)
    json.dump(summary, open(os.path.join(d, 'summary.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()

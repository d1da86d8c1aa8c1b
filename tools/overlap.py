#!/usr/bin/env python3
"""How much do kernels of different batches run side by side?  Reads a rocprofv3 kernel trace (csv with start / end
timestamps per dispatch) and prints, for the steady part of the run, the busy time of every kernel, the time during
which two given kernels were both on the chip, and the wall time covered by at least one kernel.

    python tools/overlap.py <kernel_trace.csv> [--a select_scan --b nms_segments_pairs] [--skip 0.5]
"""
import argparse
import csv
import json
import re
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    out, cur_s, cur_e = 0, None, None
    merged = []
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                merged.append((cur_s, cur_e))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        merged.append((cur_s, cur_e))
    return merged


def inter(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--a', default='select_scan')
    ap.add_argument('--b', default='nms_segments')
    ap.add_argument('--skip', type=float, default=0.5, help='fraction of the trace (by time) to skip as warm-up')
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    cut = t0 + args.skip * (t1 - t0)
    rows = [r for r in rows if r[1] >= cut]
    by = defaultdict(list)
    for name, s, e in rows:
        mm = re.search(r'(\w+)_kernel', name)
        short = mm.group(1) if mm else name.split('(')[0].split('<')[0].split('::')[-1].strip()[:40]
        by[short].append((s, e))
    wall = max(r[2] for r in rows) - min(r[1] for r in rows)
    busy = sum(e - s for s, e in union([(s, e) for _, s, e in rows]))
    res = {'wall_us': wall / 1e3, 'some_kernel_running_us': busy / 1e3, 'kernels': {}}
    print(f'steady part: {wall / 1e3:.0f} us wall, some kernel running {busy / 1e3:.0f} us ({100 * busy / wall:.0f} %)')
    for k, iv in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        tot = sum(e - s for s, e in iv)
        res['kernels'][k] = {'launches': len(iv), 'sum_us': tot / 1e3, 'avg_us': tot / len(iv) / 1e3,
                             'union_us': sum(e - s for s, e in union(iv)) / 1e3}
        print(f'  {k:40s} n={len(iv):5d} avg {tot / len(iv) / 1e3:8.1f} us  sum {tot / 1e3:10.0f} us  ({100 * tot / wall:5.1f} % of wall)')
    a = union([iv for k, v in by.items() if args.a in k for iv in v])
    b = union([iv for k, v in by.items() if args.b in k for iv in v])
    ta, tb, tab = sum(e - s for s, e in a), sum(e - s for s, e in b), inter(a, b)
    res['pair'] = {'a': args.a, 'b': args.b, 'a_us': ta / 1e3, 'b_us': tb / 1e3, 'both_us': tab / 1e3}
    print(f'{args.a}: {ta / 1e3:.0f} us, {args.b}: {tb / 1e3:.0f} us, both on the chip: {tab / 1e3:.0f} us '
          f'({100 * tab / max(min(ta, tb), 1):.0f} % of the shorter)')
    # every pair of the eight kernels with the most busy time: how long both were on the chip
    big = [k for k, _ in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:8]]
    un = {k: union(by[k]) for k in big}
    res['pairs_us'] = {}
    print('both on the chip (us):')
    for i, ka in enumerate(big):
        for kb in big[i + 1:]:
            t = inter(un[ka], un[kb]) / 1e3
            res['pairs_us'][ka + ' | ' + kb] = t
            if t >= 1.0:
                print(f'  {ka:24s} {kb:24s} {t:9.0f}')
    sum_all = sum(e - s for _, s, e in rows)
    res['sum_of_kernel_durations_us'] = sum_all / 1e3
    print(f'sum of kernel durations {sum_all / 1e3:.0f} us = {sum_all / max(busy, 1):.2f} x the busy time')
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()

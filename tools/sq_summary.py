"""Per-kernel means of one rocprofv3 SQ counter pass of the bench command, as profiles/<tag>_sq_counters.txt holds them.

    python tools/sq_summary.py <pass_dir> > profiles/r2_sq_counters.txt
"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r'((?:select|nms|assign|pack|score|activate|direct)_[a-z_]+?)(?:<|\()', r['Kernel_Name'])
        if not m:
            continue
        k = m.group(1)
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        acc[k]['_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        meta[k] = dict(vgpr=r.get('VGPR_Count') or r.get('Arch_VGPR_Count'), sgpr=r.get('SGPR_Count'), grid=r.get('Grid_Size'))
    for k, v in acc.items():
        us = sum(v['_us']) / len(v['_us'])
        if us < 15:
            continue
        c = {n: int(sum(x) / len(x)) for n, x in sorted(v.items()) if n != '_us'}
        print(f"{k} {dict(meta[k], us=round(us, 3))} {c}")


if __name__ == '__main__':
    main()

"""Turns two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, one run each) of the bench command into
profiles/traffic.json: HBM bytes per launch of every kernel of the path.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <workload>/<regime>

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts a 128-byte request of a wide
(16 bytes per lane) coalesced read as 64 bytes, so the read side is doubled; WRITE_SIZE is exact for wide
stores.  Kernels whose accesses are narrower are uncalibrated and flagged so.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd.build import source_hash  # noqa: E402


def recorded_hash(pass_dir):
    """The hash tools/profile_round.sh left next to the pass directories, else the hash of the sources as they are now."""
    f = os.path.join(os.path.dirname(os.path.abspath(pass_dir.rstrip('/'))), 'source_sha256.txt')
    if os.path.exists(f):
        return open(f).read().strip()
    return source_hash()

WIDE_READERS = {'select_scan', 'select_sample', 'score_hist', 'select_emit', 'activate_scores', 'stream_read'}
PAT = re.compile(r'(select_[a-z]+|nms_[a-z]+|score_hist|activate_scores|assign_[a-z]+|pack_detections|stream_read)')


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        m = PAT.search(r['Kernel_Name'])
        if m:
            acc[m.group(1)].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, key = sys.argv[1:4]
    fetch, nf = per_kernel(fetch_dir, 'FETCH_SIZE')
    write, _ = per_kernel(write_dir, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
        wide = k in WIDE_READERS
        rd = f_kib * 1024.0 * (2.0 if wide else 1.0)
        out[k] = dict(fetch_size_kib_raw=f_kib, write_size_kib_raw=w_kib, read_bytes=rd, write_bytes=w_kib * 1024.0,
                      hbm_bytes_per_launch=rd + w_kib * 1024.0, launches_sampled=nf.get(k, 0),
                      read_correction='x2 (wide coalesced reads on gfx950)' if wide else 'none (narrow accesses: uncalibrated)')
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'traffic.json')
    try:
        with open(path) as fh:
            allt = json.load(fh)
    except (OSError, ValueError):
        allt = {}
    out['_source_sha256'] = recorded_hash(sys.argv[1])          # the kernels these counters were taken on
    allt[key] = out
    with open(path, 'w') as fh:
        json.dump(allt, fh, indent=1, sort_keys=True)
    for k, v in out.items():
        if k.startswith('_'):
            continue
        print(k, round(v['hbm_bytes_per_launch'] / 1e6, 2), 'MB/launch')


if __name__ == '__main__':
    main()

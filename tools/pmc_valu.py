"""Turns one rocprofv3 SQ counter pass of the bench command (SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_WAVES, GRBM_GUI_ACTIVE)
into profiles/valu.json: VALU wave-instructions per launch of the NMS kernel and the shader clock the pass saw.

    python tools/pmc_valu.py <pass_dir> <workload>/<regime>

bench.py turns these into `roofline.valu_frac` = VALU instructions x 4 cycles / (SIMDs x clock x kernel duration):
the share of the chip's VALU issue slots the kernel fills, which is what bounds a greedy NMS (its HBM share is tiny).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd.build import source_hash  # noqa: E402


def recorded_hash(pass_dir):
    """The hash tools/profile_round.sh left next to the pass directories, else the hash of the sources as they are now."""
    f = os.path.join(os.path.dirname(os.path.abspath(pass_dir.rstrip('/'))), 'source_sha256.txt')
    if os.path.exists(f):
        return open(f).read().strip()
    return source_hash()


def main():
    d, key = sys.argv[1:3]
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'nms_segments' not in r['Kernel_Name']:
            continue
        acc[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
        acc['_us']['v'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    mean = {k: sum(v['v']) / len(v['v']) for k, v in acc.items()}
    xcds = 8
    clock_mhz = mean['GRBM_GUI_ACTIVE'] / xcds / mean['_us']            # the counter sums the eight XCDs
    out = dict(valu_insts_per_launch=mean['SQ_INSTS_VALU'], salu_insts_per_launch=mean.get('SQ_INSTS_SALU'),
               waves_per_launch=mean.get('SQ_WAVES'), simds=1024, clock_mhz=clock_mhz, kernel_us_in_pass=mean['_us'],
               valu_frac_in_pass=mean['SQ_INSTS_VALU'] * 4.0 / (1024 * clock_mhz * mean['_us']))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'valu.json')
    try:
        with open(path) as fh:
            allv = json.load(fh)
    except (OSError, ValueError):
        allv = {}
    out['source_sha256'] = recorded_hash(sys.argv[1])           # the kernels these counters were taken on
    allv.setdefault(key, {})['nms_segments'] = out
    with open(path, 'w') as fh:
        json.dump(allv, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Condense the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) of tools/pmc_kernels.py into per-launch
HBM traffic figures: profiles/pmc_kernels.json (all groups) and profiles/pmc_match_top2.json (the roofline kernel).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide (16 B / lane) coalesced reads by exactly 2x
(MI355X_MICROARCH.md, HBM) -> corrected; other access widths are uncalibrated, the raw counters are kept beside it.
usage: pmc_to_json.py <fetch_dir> <write_dir> <out_dir>"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_kernels import GROUPS, REPS  # noqa: E402


def per_group(d, counter):
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    out, cur = {}, None
    for r in rows:
        if 'max2_kernel' in r['Kernel_Name']:
            gi = int(r['Grid_Size']) // 64 - 1 if int(r['Grid_Size']) % 64 == 0 else None
            # the marker's grid is rounded up to the block size: recover the group from the order instead
            cur = len(out)
            out[cur] = []
            continue
        if cur is not None:
            out[cur].append((r['Kernel_Name'], float(r['Counter_Value'])))
    res = {}
    for gi, name in enumerate(GROUPS):
        d_ = out.get(gi, [])
        # the measured kernel is the one launched REPS times (helper launches such as fills are ignored: keep the
        # dispatches of the most frequent kernel name)
        names = {}
        for k, v in d_:
            names.setdefault(k, []).append(v)
        if not names:
            continue
        k = max(names, key=lambda n: (len(names[n]), sum(names[n])))
        res[name] = (k, sum(names[k]) / len(names[k]), len(names[k]))
    return res


if __name__ == '__main__':
    fetch_dir, write_dir, out_dir = sys.argv[1:4]
    F, Wr = per_group(fetch_dir, 'FETCH_SIZE'), per_group(write_dir, 'WRITE_SIZE')
    # FETCH_SIZE calibration per access pattern (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
    # pattern"): the streaming kernels (conv / resblock tile staging, match row stages: 16 B per lane, consecutive lanes
    # consecutive addresses) are under-reported 2x; for the gather kernels the RAW counter already equals the bytes a
    # perfect cache would fetch (warp 2x: raw 29.1 MB vs 24.9 MB map + 4.1 MB flow), so no correction applies there.
    gathers = ('warp LR', 'warp 2x', 'gather 2x', 'aligned_sample 2x', 'bicubic x4', 'warp up2 2x', 'conf_alpha LR', 'conf_alpha 2x')
    traffic, detail = {}, {}
    for name in GROUPS:
        if name in F and name in Wr:
            corr = 1.0 if name in gathers else 2.0
            traffic[name] = (corr * F[name][1] + Wr[name][1]) * 1024.0
            detail[name] = {'kernel': F[name][0][:80], 'FETCH_SIZE_KiB': F[name][1], 'WRITE_SIZE_KiB': Wr[name][1],
                            'fetch_correction': corr, 'launches': [F[name][2], Wr[name][2]]}
    d = {'traffic_bytes_per_launch': traffic, 'detail': detail,
         'how': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes on tools/pmc_kernels.py (270x480, RefVSR_small); '
                'traffic = fetch_correction x FETCH_SIZE + WRITE_SIZE (correction 2 for the streaming kernels, 1 for the gathers)'}
    json.dump(d, open(os.path.join(out_dir, 'pmc_kernels.json'), 'w'), indent=1)
    if 'match_top2' in traffic:
        m = {'kernel': 'match_top2_kernel', 'FETCH_SIZE_KiB': detail['match_top2']['FETCH_SIZE_KiB'],
             'WRITE_SIZE_KiB': detail['match_top2']['WRITE_SIZE_KiB'], 'launches': detail['match_top2']['launches'],
             'fetch_correction': 2.0, 'traffic_bytes_per_launch': traffic['match_top2'],
             'algorithmic_read_bytes': (32400 + 129600) * 304.0, 'algorithmic_write_bytes': 129600 * 16.0, 'how': d['how']}
        json.dump(m, open(os.path.join(out_dir, 'pmc_match_top2.json'), 'w'), indent=1)
    print(json.dumps(d))

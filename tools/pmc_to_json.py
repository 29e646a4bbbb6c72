#!/usr/bin/env python3
"""Condense the two rocprofv3 --pmc passes of tools/pmc_match.py (FETCH_SIZE, WRITE_SIZE; separate runs) into the
per-launch HBM traffic figure used by bench.py's roofline object.  FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x (MI355X_MICROARCH.md, HBM) -> corrected."""
import csv, glob, json, sys

def mean_counter(d, name, kernel):
    vals = []
    for f in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name'] and r['Counter_Name'] == name:
                vals.append(float(r['Counter_Value']))
    return sum(vals) / len(vals), len(vals)

if __name__ == '__main__':
    fetch_dir, write_dir, out = sys.argv[1:4]
    f, nf = mean_counter(fetch_dir, 'FETCH_SIZE', 'match_top2')
    w, nw = mean_counter(write_dir, 'WRITE_SIZE', 'match_top2')
    d = {'kernel': 'match_top2_kernel', 'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'launches': [nf, nw],
         'fetch_correction': 2.0, 'traffic_bytes_per_launch': (2.0 * f + w) * 1024.0,
         'algorithmic_read_bytes': (32400 + 129600) * 304.0, 'algorithmic_write_bytes': 129600 * 16.0,
         'how': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes on tools/pmc_match.py '
                '(270x480 vs 135x240, variant 4)'}
    json.dump(d, open(out, 'w'), indent=1)
    print(json.dumps(d))

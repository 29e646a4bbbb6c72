#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc run per (kernel, grid size):  python tools/pmc_summary.py <out_dir> [name filter]"""
import collections
import csv
import glob
import sys


def main():
    fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = collections.OrderedDict()
    for f in fs:
        for r in csv.DictReader(open(f)):
            if flt and flt not in r['Kernel_Name']:
                continue
            k = (r['Kernel_Name'][:70], r.get('Grid_Size', ''), r.get('Workgroup_Size', ''))
            acc.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, m in acc.items():
        print('%s grid=%s wg=%s n=%d' % (k[0], k[1], k[2], max(len(v) for v in m.values())))
        print('    ' + '  '.join('%s=%.4g' % (c.replace('SQ_', ''), sum(v) / len(v)) for c, v in sorted(m.items())))


if __name__ == '__main__':
    main()

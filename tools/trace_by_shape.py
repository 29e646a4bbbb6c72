#!/usr/bin/env python3
"""Per-shape kernel durations from a rocprofv3 --kernel-trace CSV: dispatches grouped by
(kernel name, grid, workgroup size, LDS bytes) -> calls, mean / min / median duration, total.
usage: trace_by_shape.py <kernel_trace.csv> [min_total_us]"""
import collections
import csv
import statistics
import sys


def short(name):
    n = name.replace('void ', '')
    if n.startswith('_Z'):
        # mangled: keep the readable core
        for key in ('match_top2_kernel_v4', 'match_top2_kernel', 'match_patches_kernel', 'warp_nhwc16_kernel', 'pack_nhwc16_kernel',
                    'spynet_level_input_kernel', 'aligned_sample_kernel', 'block_gather_nhwc16_kernel', 'block_gather_rgb_kernel',
                    'match_exact_kernel'):
            if key in n:
                return key
    return n.split('(')[0][:60]


def main():
    path = sys.argv[1]
    min_total = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])),
               int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']), int(r['Workgroup_Size_X']), int(r['LDS_Block_Size']),
               int(r['VGPR_Count']) + int(r['Accum_VGPR_Count']))
        groups[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    tot_all = sum(sum(v) for v in groups.values())
    print('%-58s %6s %4s %5s %7s %5s %6s %8s %8s %8s %10s %6s' % ('kernel', 'wgs', 'yz', 'wgsz', 'lds', 'vgpr', 'calls', 'mean_us',
                                                                 'min_us', 'med_us', 'total_us', 'pct'))
    for (name, wgs, yz, wgsz, lds, vg), d in rows:
        tot = sum(d)
        if tot < min_total:
            continue
        print('%-58s %6d %4d %5d %7d %5d %6d %8.1f %8.1f %8.1f %10.0f %6.2f' % (name, wgs, yz, wgsz, lds, vg, len(d), tot / len(d), min(d),
                                                                              statistics.median(d), tot, 100 * tot / tot_all))


if __name__ == '__main__':
    main()

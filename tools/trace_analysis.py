#!/usr/bin/env python3
"""Steady-state frame anatomy from a rocprofv3 --kernel-trace CSV of bench.py: per-kernel time and launches per frame,
per-queue totals and the kernel-concurrency histogram between two match_top2 launches (one per frame).
usage: trace_analysis.py <bench_kernel_trace.csv> [first_frame last_frame]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    f0, f1 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 20)
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Queue_Id'])) for r in rows)
    ms = [e for e in ev if 'match_top2' in e[2]]
    t0, t1, nfr = ms[f0][0], ms[f1][0], f1 - f0
    win = [e for e in ev if t0 <= e[0] < t1]
    print('%d kernels, %d match launches; frames %d..%d: %.3f ms/frame (profiled)' % (len(ev), len(ms), f0, f1, (t1 - t0) / 1e6 / nfr))
    pts = []
    for s, e, _, _ in win:
        pts += [(s, 1), (min(e, t1), -1)]
    pts.sort()
    conc, cur, last = collections.Counter(), 0, t0
    for t, d in pts:
        conc[cur] += t - last
        last, cur = t, cur + d
    conc[cur] += t1 - last
    tot = float(sum(conc.values()))
    print('kernels in flight: ' + ', '.join('%d: %.1f%%' % (k, 100 * conc[k] / tot) for k in sorted(conc)))
    ksum, kcnt, qs = collections.Counter(), collections.Counter(), collections.Counter()
    for s, e, n, q in win:
        key = n.split('(')[0].replace('void ', '')[-48:]
        ksum[key] += e - s
        kcnt[key] += 1
        qs[q] += e - s
    print('kernel time per frame: %.3f ms in %.1f launches per frame (all kernels of the window, copies / fills included)' % (sum(ksum.values()) / 1e6 / nfr, len(win) / float(nfr)))
    for k, v in ksum.most_common(16):
        print('  %-48s %6.3f ms  %6.1f launches/frame  avg %7.1f us' % (k, v / 1e6 / nfr, kcnt[k] / float(nfr), v / kcnt[k] / 1e3))
    print('per HIP queue, ms per frame: ' + ', '.join('q%d %.3f' % (q, v / 1e6 / nfr) for q, v in sorted(qs.items())))
    # which logical stream is which queue: the three kernels with the most time per queue, and the Stream_Id column when present
    qk = collections.defaultdict(collections.Counter)
    qstream = collections.defaultdict(set)
    sid = 'Stream_Id' if rows and 'Stream_Id' in rows[0] else None
    for r in rows:
        s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if t0 <= s_ < t1:
            q = int(r['Queue_Id'])
            qk[q][r['Kernel_Name'].split('(')[0].replace('void ', '')[-40:]] += e_ - s_
            if sid:
                qstream[q].add(r[sid])
    for q in sorted(qk):
        print('  q%d%s: ' % (q, (' streams ' + ','.join(sorted(qstream[q]))) if sid else '') +
              '; '.join('%s %.2f ms' % (k, v / 1e6 / nfr) for k, v in qk[q].most_common(4)))


if __name__ == '__main__':
    main()

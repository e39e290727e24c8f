#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`) as the per-kernel stats table that
`--output-format csv` would give: name, calls, total/avg/min/max duration, share of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--csv out.csv] [--top 40] [--skip-first N]
"""
import argparse
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--csv')
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--steady', metavar='PATTERN:N',
                    help='restrict to the window spanned by the LAST N dispatches of the kernel matching PATTERN '
                         '(e.g. prefilter_scan:10 = the 10 timed steps of bench.py), minus one step of lead-in')
    args = ap.parse_args()
    con = sqlite3.connect(args.db)
    cur = con.cursor()
    where, params = '', ()
    if args.steady:
        pat, n = args.steady.rsplit(':', 1)
        marks = cur.execute("select d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on "
                            "d.kernel_id = s.id where s.kernel_name like ? order by d.start", ('%' + pat + '%',)).fetchall()
        n = int(n)
        first, last = marks[-n], marks[-1]
        lead = (last[0] - first[0]) // max(n - 1, 1)          # one step before the first marker
        where, params = 'where d.start >= ? and d.end <= ?', (first[0] - lead, last[1] + 2000000)
        print('steady window: %.3f ms for %d steps' % ((last[1] - first[0] + lead) / 1e6, n))
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        + where + " group by s.kernel_name order by 3 desc", params).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['name,calls,total_ns,avg_ns,min_ns,max_ns,percent']
    for name, calls, tot, avg, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.3f' % (name.replace('"', "'"), calls, tot, avg, mn, mx, 100.0 * tot / total))
    if args.csv:
        open(args.csv, 'w').write('\n'.join(lines) + '\n')
    print('%-90s %7s %12s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
    for name, calls, tot, avg, mn, mx in rows[:args.top]:
        print('%-90s %7d %12.1f %12.2f %7.2f' % (name[:90], calls, tot / 1e3, avg / 1e3, 100.0 * tot / total))
    print('total GPU kernel time: %.1f us over %d kernels' % (total / 1e3, len(rows)))


if __name__ == '__main__':
    sys.exit(main())

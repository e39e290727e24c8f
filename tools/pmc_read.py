#!/usr/bin/env python
"""Per-kernel averages of ANY counters of a rocprofv3 --pmc pass (rocpd database): for every kernel whose name contains
--match, the dispatches' counter values (SUM over hardware instances per dispatch), averaged, next to the dispatch duration
under the counters.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d DIR -o pmc -- python tools/loss_form_probe.py --pmc
    python tools/pmc_read.py DIR/**/pmc_results.db --match retina_loss

Counter semantics (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
per wave; WAIT_ANY (parked on s_waitcnt / a barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES; FETCH_SIZE is
in KB and reports half the bytes of a wide coalesced stream on gfx950."""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db', nargs='+')
    ap.add_argument('--match', default='odtk')
    ap.add_argument('--skip', type=int, default=0, help='dispatches of each kernel to drop from the front (warm-up)')
    a = ap.parse_args()
    for db in a.db:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select d.id, s.kernel_name, d.end - d.start, p.name, sum(e.value) from rocpd_pmc_event e "
                           "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
                           "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by d.id, p.name order by d.start").fetchall()
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for did, name, ns, pmc, v in rows:
            if a.match in name:
                per[name][pmc].append((did, v, ns))
        print('#', db)
        for name, counters in sorted(per.items()):
            short = name[:110]
            for pmc, vals in sorted(counters.items()):
                vals = vals[a.skip:]
                if not vals:
                    continue
                print('%-110s %-28s dispatches %3d  avg %16.1f  min %16.1f  max %16.1f  avg_ns %10.0f'
                      % (short, pmc, len(vals), sum(v for _, v, _ in vals) / len(vals), min(v for _, v, _ in vals),
                         max(v for _, v, _ in vals), sum(ns for _, _, ns in vals) / len(vals)))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""HBM traffic per launch of the post-processing kernels from two rocprofv3 --pmc passes (one counter
per pass, as MI355X_MICROARCH.md prescribes):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d F -o pmc -- python tools/postproc_bench.py ... --iters 5
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d W -o pmc -- python tools/postproc_bench.py ... --iters 5
    python tools/pmc_traffic.py F/.../pmc_results.db W/.../pmc_results.db --csv profiles/x.csv \
        --json profiles/r01_pmc_traffic.json --key bf16_logits_channels_last --scores 122860800 --bytes-per-score 2

Counter rows come per dispatch and per hardware instance: a dispatch's value is the SUM over instances.
FETCH_SIZE / WRITE_SIZE are in KB (1024 B); on gfx950 FETCH_SIZE counts HALF the bytes of a wide
coalesced stream, so it is doubled before comparing with a byte count (guide, section HBM)."""
import argparse
import collections
import json
import sqlite3


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select d.id, s.kernel_name, d.end - d.start, sum(e.value) from rocpd_pmc_event e "
                       "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
                       "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ? group by d.id", (counter,)).fetchall()
    agg = collections.defaultdict(list)
    for _, name, ns, v in rows:
        agg[name].append((v, ns))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_db')
    ap.add_argument('write_db')
    ap.add_argument('--csv')
    ap.add_argument('--json')
    ap.add_argument('--key')
    ap.add_argument('--scores', type=int)
    ap.add_argument('--bytes-per-score', type=int, default=2)
    ap.add_argument('--comment', default='')
    a = ap.parse_args()
    lines = ['# ' + a.comment, 'kernel,counter,dispatches,avg,min,max,avg_kernel_ns_under_pmc']
    got = {}
    for db, counter in ((a.fetch_db, 'FETCH_SIZE'), (a.write_db, 'WRITE_SIZE')):
        for name, vals in sorted(per_kernel(db, counter).items()):
            if 'odtk' not in name:
                continue
            v = [x[0] for x in vals]
            lines.append('%s,%s,%d,%.3f,%.1f,%.1f,%.2f' % (name, counter, len(v), sum(v) / len(v), min(v), max(v),
                                                          sum(x[1] for x in vals) / len(vals)))
            if 'prefilter_scan' in name:
                got[counter] = sum(v) / len(v)
    print('\n'.join(lines))
    if a.csv:
        open(a.csv, 'w').write('\n'.join(lines) + '\n')
    if a.json and a.key and 'FETCH_SIZE' in got and 'WRITE_SIZE' in got:
        doc = json.load(open(a.json))
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        doc['kernel_src_sha16'] = bench.kernel_src_hash()      # bench.py quotes this file only for these kernel sources
        traffic = int(round((2.0 * got['FETCH_SIZE'] + got['WRITE_SIZE']) * 1024))
        doc[a.key] = {'scores_per_launch': a.scores, 'fetch_kb': round(got['FETCH_SIZE'], 1), 'write_kb': round(got['WRITE_SIZE'], 1),
                      'traffic_bytes': traffic, 'algorithmic_bytes': a.scores * a.bytes_per_score}
        json.dump(doc, open(a.json, 'w'), indent=1)
        print('prefilter traffic %.1f MB vs %.1f MB algorithmic' % (traffic / 1e6, a.scores * a.bytes_per_score / 1e6))


if __name__ == '__main__':
    main()

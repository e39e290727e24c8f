#!/usr/bin/env python
"""Compact table of `make -C retinanet-examples_amd/csrc resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage):
one line per kernel with VGPRs / AGPRs / SGPRs, scratch bytes per lane, occupancy (waves per SIMD) and static LDS.
No GPU needed.  `--filter nms` restricts the kernels; exit code 1 when any listed kernel uses scratch."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
    return [re.sub(r'\(.*', '', o).replace('odtk::', '').replace('void ', '') for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--filter', default='')
    args = ap.parse_args()
    log = subprocess.run(['make', '-C', os.path.join(ROOT, 'retinanet-examples_amd', 'csrc'), 'resource-usage'],
                         capture_output=True, text=True)
    text = log.stdout + log.stderr
    rows, cur = [], None
    for line in text.split('\n'):
        m = re.search(r'remark: +([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: +(\S+)', line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == 'Function Name':
            cur = {'name': val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    names = demangle([r['name'] for r in rows])
    bad = 0
    print('%-78s %5s %5s %5s %8s %4s %8s' % ('kernel', 'VGPR', 'AGPR', 'SGPR', 'scratch', 'occ', 'LDS'))
    for r, n in zip(rows, names):
        if args.filter and args.filter not in n:
            continue
        scratch = int(r.get('ScratchSize', '0'))
        bad += scratch != 0
        print('%-78s %5s %5s %5s %8d %4s %8s' % (n[:78], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), scratch,
                                                  r.get('Occupancy'), r.get('LDS Size')))
    print('kernels with scratch: %d' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

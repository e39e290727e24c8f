#!/usr/bin/env python
"""Compact table of `make -C retinanet-examples_amd/csrc resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage):
one line per kernel with VGPRs / AGPRs / SGPRs, scratch bytes per lane, occupancy (waves per SIMD) and static LDS.
No GPU needed.  `--filter nms` restricts the kernels; exit code 1 when any listed kernel uses scratch.
`--from-library` (what tests/test_kernel_resources.py runs: seconds instead of a five-minute recompile) reads the same figures
out of the BUILT libodtk_hip.so -- the gfx950 code object's AMDGPU metadata notes (.vgpr_count, .private_segment_fixed_size ...);
it refuses a library older than its sources."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
    return [re.sub(r'\(.*', '', o).replace('odtk::', '').replace('void ', '') for o in out]


LLVM = '/opt/rocm/lib/llvm/bin'


def rows_from_library(lib):
    """[{name, VGPRs, AGPRs, TotalSGPRs, ScratchSize, LDS Size}] of the gfx950 code object inside `lib`."""
    import glob
    import tempfile
    src = os.path.join(ROOT, 'retinanet-examples_amd', 'csrc')
    newest = max(os.path.getmtime(f) for f in glob.glob(os.path.join(src, '*.hpp')) + glob.glob(os.path.join(src, '*.hip')))
    if not os.path.isfile(lib) or os.path.getmtime(lib) < newest:
        raise SystemExit('%s is missing or older than its sources: build it first (make -C retinanet-examples_amd/csrc)' % lib)
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'gfx950.co')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fat], check=True)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--input=' + fat, '--output=' + co, '--unbundle'], check=True)
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
    rows, cur = [], None
    keys = {'.vgpr_count': 'VGPRs', '.agpr_count': 'AGPRs', '.sgpr_count': 'TotalSGPRs', '.private_segment_fixed_size': 'ScratchSize',
            '.group_segment_fixed_size': 'LDS Size'}
    for line in notes.split('\n'):
        m = re.match(r'\s*(-\s+)?(\.[a-z_]+):\s+(\S+)\s*$', line)
        if not m:
            continue
        if m.group(1) and m.group(2) in ('.agpr_count', '.args'):        # a new kernel entry of amdhsa.kernels starts with "- <first key>"
            cur = {}
            rows.append(cur)
        key, val = m.group(2), m.group(3)
        if cur is None:
            continue
        if key == '.name':
            cur['name'] = val
        elif key in keys:
            cur[keys[key]] = val
    return [r for r in rows if 'name' in r and 'VGPRs' in r]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--filter', default='')
    ap.add_argument('--from-library', action='store_true')
    args = ap.parse_args()
    if args.from_library:
        rows = rows_from_library(os.path.join(ROOT, 'retinanet-examples_amd', 'odtk', 'libodtk_hip.so'))
        return report(rows, args)
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
    return report(rows, args)


def report(rows, args):
    names = demangle([r['name'] for r in rows])
    bad = 0
    print('%-78s %5s %5s %5s %8s %4s %8s' % ('kernel', 'VGPR', 'AGPR', 'SGPR', 'scratch', 'occ', 'LDS'))
    for r, n in zip(rows, names):
        if args.filter and args.filter not in n:
            continue
        scratch = int(r.get('ScratchSize', '0'))
        bad += scratch != 0
        print('%-78s %5s %5s %5s %8d %4s %8s' % (n[:78], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), scratch,
                                                  r.get('Occupancy') or '-', r.get('LDS Size')))
    print('kernels with scratch: %d' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

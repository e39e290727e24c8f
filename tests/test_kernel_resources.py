"""No kernel of the library may use scratch memory (round 2's review found the rotated NMS spilling 32 B per lane at the
128-VGPR cap of its 1024-thread workgroup).  The figures are in the built library itself: tools/resource_usage.py --from-library
reads the AMDGPU metadata notes of libodtk_hip.so's gfx950 code object (registers, scratch, LDS of every kernel; no GPU, no
recompile -- the five-minute `hipcc -Rpass-analysis=kernel-resource-usage` form of the tool prints the same table plus occupancy)
and exits non-zero when any kernel has a scratch size or the library is older than its sources."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_uses_scratch():
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'resource_usage.py'), '--from-library'], capture_output=True, text=True,
                         timeout=300)
    table = run.stdout
    assert 'kernels with scratch: 0' in table, table[-3000:] + run.stderr[-1000:]
    assert run.returncode == 0
    rows = {line[:78].strip(): line[78:].split() for line in table.splitlines()[1:] if len(line) > 80}
    # the kernels this is about exist under the names the table prints
    for name in ('nms_kernel<6, false, 1>', 'nms_kernel<6, false, 2>', 'nms_kernel<4, false, 0>', 'rotated_sup_matrix_kernel',
                 'select_decode_kernel<6, BF16, true, 4096>', 'prefilter_scan_kernel<BF16, true, true>'):
        assert name in rows, (name, sorted(rows)[:10])
    # 1024-thread workgroups: 128 registers per lane is the cap (4 waves per SIMD x 128 = the 512-entry file)
    for name in ('nms_kernel<6, false, 2>', 'nms_kernel<6, true, 2>', 'nms_kernel<4, false, 0>'):
        vgpr, agpr = int(rows[name][0]), int(rows[name][1])
        assert vgpr + agpr <= 128, (name, vgpr, agpr)

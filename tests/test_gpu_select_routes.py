"""select_decode's routes for segments that several workgroups share (csrc/select_decode.hpp): the cooperative route behind a
segment-local barrier, and its fall-back, the tournament.  Which one a segment takes is decided at run time by ONE
compare-and-swap -- a barrier that times out, a slice that does not fit LDS, a plateau wider than the sort all send the whole
segment to the tournament -- so the result must not depend on it.  The library reads ODTK_SELECT_COOP_TICKS once per process:
the selection suites run again in child processes with
    1  every barrier times out unless the partners are already there: segments of ONE launch take different routes
    0  the cooperative route off: every shared segment through the tournament (rounds 3-4)
(the default, 3000 ticks = 30 us, is what every other GPU test runs under; the full-size, rotated, corner and thread suites were run
the same way by hand: profiles/r05_pytest_gpu_final_tail.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('ticks', ['1', '0'])
def test_selection_is_route_independent(ticks):
    env = dict(os.environ, ODTK_SELECT_COOP_TICKS=ticks)
    run = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-p', 'no:cacheprovider',
                          os.path.join(ROOT, 'tests', 'test_gpu_parity.py') + '::test_selection_passes_every_route',
                          os.path.join(ROOT, 'tests', 'test_gpu_parity.py') + '::test_pyramid_vs_oracle',
                          os.path.join(ROOT, 'tests', 'test_gpu_parity.py') + '::test_batch_sizes',
                          os.path.join(ROOT, 'tests', 'test_gpu_fuzz.py')],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-1000:]
    assert ' passed' in run.stdout and 'failed' not in run.stdout, run.stdout[-1500:]

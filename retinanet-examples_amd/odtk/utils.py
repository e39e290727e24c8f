"""Small helpers the callers of the post-processing path use (reference odtk/utils.py).

`order_points` / `rotate_boxes` live next to the anchor code (odtk/box.py, batched forms) and are
re-exported here under the reference's names; `rotate_box` is the single-box corner helper of the
detection hand-off; `Profiler` and `ignore_sigint` are the loop utilities of infer.py / train.py.
Drawing helpers (reference utils.py:104-130) are not part of the path and are not provided.
"""
import signal
import time
import warnings
from contextlib import contextmanager

import torch

from .box import _order_quads, rotate_boxes      # noqa: F401  (reference: utils.rotate_boxes)


def order_points(pts):
    """[Q, 4, 2] corner sets -> [top-left, top-right, bottom-right, bottom-left] per quad
    (reference utils.py:15-31, one quad at a time there)."""
    return _order_quads(torch.as_tensor(pts))


def rotate_box(bbox):
    """[x, y, w, h, theta] -> the 8 corner coordinates of the rotated rectangle, as a flat list
    (reference utils.py:83-101; the 'segmentation' polygon of a rotated detection)."""
    from .infer import rotated_corners
    x, y, w, h, theta = (float(v) for v in bbox)
    return rotated_corners([x], [y], [w], [h], [theta])[0].tolist()


@contextmanager
def ignore_sigint():
    """Checkpoint writes are not interrupted by Ctrl-C (reference utils.py:133-140)."""
    previous = signal.getsignal(signal.SIGINT)
    signal.signal(signal.SIGINT, signal.SIG_IGN)
    try:
        yield
    finally:
        signal.signal(signal.SIGINT, previous)


class Profiler:
    """Named wall-clock accumulators (reference utils.py:143-172): `start` / `stop` bracket a span,
    `bump` closes one span and opens the next; `totals`, `counts`, `means` per name."""

    def __init__(self, names=('main',)):
        self.names = list(names)
        self.reset()

    def reset(self):
        now = time.time()
        self.lasts = {n: now for n in self.names}
        self.totals = {n: 0 for n in self.names}
        self.counts = {n: 0 for n in self.names}
        self.means = {n: 0 for n in self.names}

    def start(self, name='main'):
        self.lasts[name] = time.time()

    def stop(self, name='main'):
        self.totals[name] += time.time() - self.lasts[name]
        self.counts[name] += 1
        self.means[name] = self.totals[name] / self.counts[name]

    def bump(self, name='main'):
        self.stop(name)
        self.start(name)


def post_metrics(url, metrics):
    """POST each metric as form data (reference utils.py:174-177); a failure is a warning, never an error."""
    import urllib.parse
    import urllib.request
    try:
        for key, value in metrics.items():
            body = urllib.parse.urlencode({'time': int(time.time() * 1e9), 'metric': key, 'value': value}).encode()
            urllib.request.urlopen(urllib.request.Request(url, data=body), timeout=5).close()
    except Exception as exc:                                   # noqa: BLE001 -- metrics must not stop training
        warnings.warn('Warning: posting metrics failed: {}'.format(exc))

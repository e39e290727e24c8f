"""odtk/utils.py: the reference's helper names (utils.py) on the batched implementations."""
import math
import signal
import time
import warnings

import numpy as np
import pytest
import torch

from odtk import utils


def test_order_points_and_rotate_boxes_follow_the_reference_conventions():
    quad = torch.tensor([[[10., 0.], [0., 0.], [0., 5.], [10., 5.]]])          # shuffled corners of a 10 x 5 rectangle
    assert utils.order_points(quad).tolist() == [[[0., 0.], [10., 0.], [10., 5.], [0., 5.]]]     # tl, tr, br, bl
    axis, quads = utils.rotate_boxes(torch.tensor([[2., 3., 10., 4., 0.0]]))
    assert axis.tolist() == [[2., 3., 11., 6., 0., 1.]]                         # x1, y1, x + w - 1, y + h - 1, sin, cos
    assert quads.view(4, 2).tolist() == [[2., 3.], [12., 3.], [12., 7.], [2., 7.]]
    _, turned = utils.rotate_boxes(torch.tensor([[0., 0., 4., 2., math.pi / 2]]))
    assert torch.allclose(turned.view(4, 2).sort(0).values, torch.tensor([[1., -1.], [1., -1.], [3., 3.], [3., 3.]]), atol=1e-5)


def test_rotate_box_is_the_segmentation_polygon_of_a_detection():
    assert utils.rotate_box([1., 2., 5., 3., 0.]) == [1., 2., 1., 4., 5., 4., 5., 2.]           # (w - 1, h - 1) extents
    c = np.array(utils.rotate_box([0., 0., 3., 3., math.pi / 2])).reshape(4, 2)
    assert np.allclose(c, [[2, 0], [0, 0], [0, 2], [2, 2]], atol=1e-12)


def test_profiler_accumulates_spans():
    p = utils.Profiler(['a', 'b'])
    p.start('a'); time.sleep(0.01); p.stop('a')
    p.bump('b'); time.sleep(0.01); p.bump('b')
    assert p.counts == {'a': 1, 'b': 2} and p.totals['a'] >= 0.009 and p.means['b'] == p.totals['b'] / 2
    p.reset()
    assert p.totals == {'a': 0, 'b': 0} and p.means == {'a': 0, 'b': 0}


def test_ignore_sigint_restores_the_handler_and_metrics_never_raise():
    before = signal.getsignal(signal.SIGINT)
    with utils.ignore_sigint():
        assert signal.getsignal(signal.SIGINT) == signal.SIG_IGN
    assert signal.getsignal(signal.SIGINT) == before
    with pytest.raises(ValueError):
        with utils.ignore_sigint():
            raise ValueError('inside')
    assert signal.getsignal(signal.SIGINT) == before
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        utils.post_metrics('http://127.0.0.1:9/unreachable', {'loss': 1.0})
    assert any('posting metrics failed' in str(w.message) for w in caught)

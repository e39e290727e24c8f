"""hipGraph capture of the hot path (torch.cuda.CUDAGraph = hipGraph on ROCm).

Round 2's open fault: capture -> destroy -> capture again ended in a GPU memory fault.  Cause (tools/graph_bisect.py,
profiles/r03_graph_bisect_*.txt): the hipMemsetAsync of decode's counters became a MEMSET NODE, and a graph holding one
faulted on a replay that followed an eager call -- graphs of kernel nodes only never did.  The counters are now cleared by
a kernel of the library, and the binding's scratch is allocated per call inside a capture (owned by the graph) instead of
being cached across graphs (odtk/_C.py:_workspace)."""
import gc

import pytest
import torch

from odtk import _C, box, synthetic
from odtk.model import Model

pytestmark = pytest.mark.gpu

RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def _heads(batch, height, width, seed, strides=(8, 16, 32)):
    cls, dl = [], []
    for i, s in enumerate(strides):
        lg, d = synthetic.make_level(batch, 9, 16, height // s, width // s, 'dense', seed + i, dtype=torch.bfloat16)
        cls.append(lg.cuda().contiguous(memory_format=torch.channels_last))
        dl.append(d.cuda().contiguous(memory_format=torch.channels_last))
    return cls, dl, list(strides), {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}


@pytest.mark.parametrize('rotated', [False, True], ids=['axis', 'rotated'])
def test_capture_replay_destroy_capture_again(rotated):
    """capture -> replay -> (eager call that grows the eager scratch) -> replay -> destroy, three times over: every replay
    bit-identical to the eager result, no fault; nothing cached by the binding belongs to a dead graph."""
    if rotated:
        import numpy as np
        strides = [8, 16]
        anchors = {s: box.generate_anchors_rotated(s, RATIOS, SCALES, [-np.pi / 6, 0, np.pi / 6]) for s in strides}
        cls, dl = [], []
        for i, s in enumerate(strides):
            lg, d = synthetic.make_level(2, 27, 8, 128 // s, 160 // s, 'dense', 40 + i, num_box=6, dtype=torch.bfloat16)
            cls.append(lg.cuda().contiguous(memory_format=torch.channels_last))
            dl.append(d.cuda().contiguous(memory_format=torch.channels_last))
    else:
        cls, dl, strides, anchors = _heads(2, 128, 160, 5)
    run = lambda: box.detect(cls, dl, strides, anchors, 0.05, 300, 0.5, 100, rotated, logits=True)
    eager = [t.clone() for t in run()]
    assert int((eager[0] > 0).sum()) > 50
    cached_before = dict(_C._workspaces)
    for round_ in range(3):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = run()
        assert {k: v.data_ptr() for k, v in _C._workspaces.items()} == {k: v.data_ptr() for k, v in cached_before.items()}, \
            'a capture must not touch the cached eager scratch'
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(eager, captured)), 'round %d' % round_
        big = _heads(4, 256, 320, 9 + round_)                       # an eager call in between, larger than anything before
        box.detect(big[0], big[1], big[2], big[3], 0.05, 1000, 0.5, 100, logits=True)
        cached_before = dict(_C._workspaces)
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(eager, captured)), 'round %d, after the eager call' % round_
        del graph, captured
        gc.collect()
        torch.cuda.empty_cache()
        again = run()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(eager, again))


def test_model_forward_as_one_graph():
    """Model.forward(x, graph=True): one hipGraph per input geometry, owned by the engine; equal to the eager call; a weight
    update rebuilds engine and graph (the old graph is destroyed -- the re-capture case); a second geometry gets its own.
    MIOpen's find mode may pick convolution kernels that are not run-to-run reproducible (tools/determinism_probe.py,
    profiles/r03_determinism_probe.txt: 5 of 111 convolutions at batch 1), so the comparison runs with
    torch.backends.cudnn.deterministic = True, under which every convolution of the engine reproduces bit for bit."""
    saved = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        torch.manual_seed(0)
        model = Model('ResNet18FPN', classes=8)
        model.initialize(None)
        model = model.cuda().to(memory_format=torch.channels_last).eval()
        with torch.no_grad():
            model.cls_head[-1].weight.mul_(60.0)                        # detections exist
        x = torch.randn(2, 3, 256, 320, device='cuda').contiguous(memory_format=torch.channels_last)
        x2 = torch.randn(1, 3, 128, 256, device='cuda').contiguous(memory_format=torch.channels_last)

        def call(inp, graph):
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                return model(inp, graph=graph)
        eager = [t.clone() for t in call(x, False)]
        assert int((eager[0] > 0).sum()) > 20
        for _ in range(3):
            out = call(x, True)
            assert all(torch.equal(a, b) for a, b in zip(eager, out))
        engine = model.inference_engine(torch.bfloat16)
        assert len(engine._graphs) == 1
        other = call(x2, True)                                          # second geometry: its own graph
        assert len(engine._graphs) == 2
        assert all(torch.equal(a, b) for a, b in zip(call(x2, False), other))
        assert all(torch.equal(a, b) for a, b in zip(eager, call(x, True)))   # ... and the first one still replays
        # new input values through the same graph
        x3 = torch.randn_like(x)
        assert all(torch.equal(a, b) for a, b in zip(call(x3, False), call(x3, True)))
        # the prefilter's threshold table is baked into a capture: a graph co-owns the table it captured, so another score
        # threshold (which drops the engine's own reference to the old table) leaves the first graph replayable ...
        model.threshold = 0.3
        at_03 = [t.clone() for t in call(x, False)]
        assert all(torch.equal(a, b) for a, b in zip(at_03, call(x, True)))
        filler = [torch.full((724,), float('nan'), device='cuda') for _ in range(64)]   # would land in a freed 2.9 KB table
        model.threshold = 0.05
        assert all(torch.equal(a, b) for a, b in zip(eager, call(x, True)))
        del filler
        # ... and the ENGINE's bias updated in place is part of the graph key: no replay with stale thresholds
        n_graphs = len(engine._graphs)
        with torch.no_grad():
            engine.cls_head[-1].bias.add_(0.5)
        fresh = [t.clone() for t in engine.forward(x)]
        assert not torch.equal(fresh[0], eager[0])
        assert all(torch.equal(a, b) for a, b in zip(fresh, engine.replay(x))) and len(engine._graphs) == n_graphs + 1
        with torch.no_grad():
            engine.cls_head[-1].bias.sub_(0.5)
        # weights change: engine re-folded, its graphs dropped, a new capture happens (capture after destroy)
        with torch.no_grad():
            model.cls_head[-1].bias.add_(0.25)
        eager2 = [t.clone() for t in call(x, False)]
        assert not torch.equal(eager2[0], eager[0])
        gc.collect()
        for _ in range(2):
            assert all(torch.equal(a, b) for a, b in zip(eager2, call(x, True)))
        assert model.inference_engine(torch.bfloat16) is not engine
        # graph=True without a fused engine is an error, not a silent eager call
        model.fused_graph = False
        with pytest.raises(RuntimeError, match='graph=True'):
            call(x, True)
    finally:
        torch.backends.cudnn.deterministic = saved

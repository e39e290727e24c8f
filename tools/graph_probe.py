#!/usr/bin/env python
"""hipGraph capture of the path (torch.cuda.CUDAGraph = hipGraph on ROCm).

A. the post-processing alone (`box.detect`: 1 memset + 6 launches, no host sync, nothing uploaded) captured at the bench
   geometry; replay must reproduce the eager result bit for bit; wall time per call eager vs replay.
B. `Model.forward` at batch 1 (the latency figure the reference quotes for its TensorRT engines): eager engine
   vs the whole forward captured in one graph.
Every stage prints as soon as it has a number; a failing stage prints its exception and the script goes on."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch

torch.backends.cudnn.benchmark = True
from odtk import box, synthetic
from odtk.model import Model

dev = torch.device('cuda', 0)
RATIOS, SCALES = [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]
STRIDES = [8, 16, 32, 64, 128]


def wall(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def stage_a():
    g = torch.Generator(device='cuda').manual_seed(3)
    cls, dl = [], []
    for (h, w) in synthetic.level_shapes(800, 1280):
        lg = torch.randn(8, 720, h, w, device=dev, generator=g) * 0.573 - 4.595
        cls.append(lg.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
        dl.append((torch.randn(8, 36, h, w, device=dev, generator=g) * 0.2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in STRIDES}
    run = lambda: box.detect(cls, dl, STRIDES, anchors, 0.05, 1000, 0.5, 100, logits=True)
    for _ in range(3):
        eager = run()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = run()
    graph.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(eager, captured))
    print('A. post-processing bs 8: replay == eager: %s; detections %d' % (same, int((captured[0] > 0).sum())), flush=True)
    print('A. wall per call: eager %.1f us, graph replay %.1f us' % (wall(run, 200), wall(graph.replay, 200)), flush=True)


def stage_b():
    torch.manual_seed(0)
    model = Model('ResNet50FPN')
    model.initialize(None)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn(1, 3, 800, 1280, device=dev).contiguous(memory_format=torch.channels_last)

    def step():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            return model(x)
    for _ in range(8):
        eager = step()
    print('B. Model.forward bs 1 bf16, eager engine: %.0f us per image' % wall(step, 30), flush=True)
    eager = [t.clone() for t in step()]
    print('B. eager detections: %d, finite: %s' % (int((eager[0] > 0).sum()), bool(torch.isfinite(eager[1]).all())), flush=True)
    for streams in (True, False):
        try:
            model.inference_engine(torch.bfloat16).level_streams = streams
            for _ in range(3):
                again = step()
            torch.cuda.synchronize()
            print('B. level streams %s: eager run == first eager run: %s' % (streams, all(torch.equal(a, b) for a, b in zip(eager, again))), flush=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                captured = step()
            verdicts = []
            for _ in range(3):
                graph.replay()
                torch.cuda.synchronize()
                verdicts.append(all(torch.equal(a, b) for a, b in zip(eager, captured)))
            print('B. one graph for the whole forward (level streams %s): replay == eager: %s; detections %d; max |dscore| %.3g; %.0f us per image'
                  % (streams, verdicts, int((captured[0] > 0).sum()), float((captured[0] - eager[0]).abs().max()), wall(graph.replay, 30)), flush=True)
        except Exception:                                              # noqa: BLE001 -- a probe reports and moves on
            print('B. capture with level streams %s failed:' % streams, flush=True)
            traceback.print_exc()
            torch.cuda.synchronize()


for stage in ((stage_b,) if os.environ.get('PROBE_ONLY_B') else (stage_a, stage_b)):
    try:
        stage()
    except Exception:                                                  # noqa: BLE001
        print('%s failed:' % stage.__name__, flush=True)
        traceback.print_exc()

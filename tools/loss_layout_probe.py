#!/usr/bin/env python
"""Round 6: the two levers DESIGN.md section 4 left untested on the loss kernels, plus the backward's box-delta walk.

  * forward through the workspace with per-WAVE partial sums (no workgroup barrier; odtk_debug_loss_layout per_wave) over launch
    shapes, with the vectors of a trip one grid stride apart (window 0) or contiguous per wave (window 1);
  * backward: the box-delta walk in d(deltas)' memory order (one vector store per cell, box_rows 1 = the default since this round)
    against the element-per-store walk of rounds 2-5 (box_rows 0), and the two trip layouts.

Every layout is first checked against the default (sums to 1e-7, gradients bit for bit), then timed with the kernels' own dispatch
timestamps over calls that rotate through three input sets (tools/loss_form_probe.py's sets: RN50FPN 800x1280, 2 images).

    python tools/loss_layout_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from odtk import _C  # noqa: E402
import loss_form_probe as P  # noqa: E402

say = P.say
FWD, BWD, WS = 0, 1, 2
DEFAULT = {(FWD, True): (512, 1, 4, 64), (FWD, False): (512, 1, 2, 64), (BWD, True): (256, 8, 2, 1024), (BWD, False): (256, 4, 1, 256),
           (WS, True): (256, 4, 2, 256), (WS, False): (256, 4, 1, 256)}


def timed(fn, sets, iters, split=False):
    """us per call by the kernels' own dispatch timestamps; split: (walk, reduce launch of the workspace form) apart"""
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    _C.profile_collect()
    for i in range(iters):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    got = _C.profile_collect()
    ms, n = got['retina_loss_kernel']
    assert n == iters, (n, iters)
    ms2, n2 = got['loss_reduce_kernel']
    assert n2 in (0, iters), (n2, iters)
    return (ms * 1e3 / iters, ms2 * 1e3 / iters) if split else (ms + ms2) * 1e3 / iters


def reset():
    for (which, fp32), shape in DEFAULT.items():
        _C.loss_tuning(which, fp32, *shape)
        _C.loss_layout(which, fp32, 0, 0 if which == FWD else 1, 1)


def main():
    torch.cuda.init()
    ok = True
    n_lv = len(P.SIZES)
    gc = torch.full((n_lv,), 0.37, device='cuda')
    gb = torch.full((n_lv,), 1.21, device='cuda')
    fwd = lambda s: _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11)
    ws = lambda s: _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, reproducible=True)
    bwd = lambda s: _C.retina_loss_levels_backward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, gc, gb)
    reset()
    # ---- 1. every layout against the default ----
    for dtype, name in ((torch.float32, 'fp32'), (torch.bfloat16, 'bf16'), (torch.float16, 'fp16')):
        fp32 = dtype == torch.float32
        for rotated_nb in (False, True):
            if rotated_nb:
                P.NB = 6
            s = P.make_set(dtype, 3)
            P.NB = 4
            _C.loss_layout(BWD, fp32, 0, 0, 0)
            want_f = fwd(s).clone()
            want_g = [[t.clone() for t in side] for side in bwd(s)]
            for per_wave, window, threads, per_cu, unroll in ((1, 0, 256, 8, 4), (1, 1, 256, 16, 4), (0, 1, 512, 2, 2), (1, 1, 64, 32, 1), (1, 1, 1024, 2, 2)):
                _C.loss_tuning(WS, fp32, threads, per_cu, unroll, 256)
                _C.loss_layout(WS, fp32, per_wave, window, 1)
                got = ws(s)
                rel = float(((got - want_f).abs() / want_f.abs().clamp_min(1e-30)).max())
                good = rel <= 2e-7
                ok &= good
                say('agree %s NB %d forward ws per_wave %d window %d %4d x %2d x %d: sums rel %.2e  %s' % (name, 6 if rotated_nb else 4, per_wave, window, threads, per_cu, unroll, rel, 'ok' if good else 'BAD'))
            for window, box_rows in ((0, 1), (1, 1), (1, 0)):
                _C.loss_layout(BWD, fp32, 0, window, box_rows)
                got_g = bwd(s)
                good = all(torch.equal(a, b) for a, b in zip(got_g[0] + got_g[1], want_g[0] + want_g[1]))
                ok &= good
                say('agree %s NB %d backward window %d box_rows %d: gradients bit-equal  %s' % (name, 6 if rotated_nb else 4, window, box_rows, 'ok' if good else 'BAD'))
            reset()
            del s
    # ---- 2. time ----
    _C.profile_enable(True, ('retina_loss_kernel', 'loss_reduce_kernel'))
    logits = sum(P.B * P.A * P.C * h * w for h, w in P.SIZES)
    for dtype, name in ((torch.float32, 'fp32'), (torch.bfloat16, 'bf16')):
        fp32 = dtype == torch.float32
        elem = 4 if fp32 else 2
        sets = [P.make_set(dtype, 10 + i) for i in range(3)]
        frac = lambda us, passes: logits * elem * passes / us / 1e3 / 8000
        # backward
        rows = []
        for rep in range(2):
            for window, box_rows in ((0, 0), (0, 1), (1, 0), (1, 1)):
                _C.loss_layout(BWD, fp32, 0, window, box_rows)
                rows.append((window, box_rows, timed(bwd, sets, 30)))
        for window, box_rows in ((0, 0), (0, 1), (1, 0), (1, 1)):
            t = [r[2] for r in rows if r[:2] == (window, box_rows)]
            say('time %s backward window %d box_rows %d: %6.2f / %6.2f us  -> %.3f of 8 TB/s' % (name, window, box_rows, t[0], t[1], frac(min(t), 2)))
        shapes = []
        for window in (0, 1):
            for threads in (256, 512, 1024):
                for per_cu in (4, 8, 16):
                    for unroll in (1, 2, 4):
                        _C.loss_tuning(BWD, fp32, threads, per_cu, unroll, 1024)
                        _C.loss_layout(BWD, fp32, 0, window, 1)
                        shapes.append((timed(bwd, sets, 15), window, threads, per_cu, unroll))
        shapes.sort()
        for r in shapes[:5] + shapes[-1:]:
            say('shape %s backward box_rows 1: %6.2f us  window %d  threads %4d  per_cu %2d  unroll %d' % ((name,) + r))
        reset()
        # forward: atomics (default) for reference, then the workspace form over layouts and shapes
        t = [timed(fwd, sets, 30) for _ in range(2)]
        say('time %s forward (atomics, default shape): %6.2f / %6.2f us  -> %.3f of 8 TB/s' % (name, t[0], t[1], frac(min(t), 1)))
        t = [timed(ws, sets, 30, split=True) for _ in range(2)]
        say('time %s forward (workspace, default shape): walk %6.2f / %6.2f us + reduce launch %5.2f / %5.2f us' % (name, t[0][0], t[1][0], t[0][1], t[1][1]))
        for window in (0, 1):                                 # the atomics form with contiguous trips, over its shapes
            for threads, per_cu, unroll in ((512, 1, 4), (512, 1, 2), (256, 2, 4), (256, 2, 2), (256, 4, 2), (1024, 1, 2)):
                _C.loss_tuning(FWD, fp32, threads, per_cu, unroll, 64)
                _C.loss_layout(FWD, fp32, 0, window, 1)
                say('shape %s forward (atomics): %6.2f us  window %d  threads %4d  per_cu %2d  unroll %d' % (name, timed(fwd, sets, 20), window, threads, per_cu, unroll))
        reset()
        shapes = []
        for per_wave in (0, 1):
            for window in (0, 1):
                for threads in (64, 256, 512):
                    for per_cu in (2, 4, 8, 16, 32):
                        if threads * per_cu > 8192 or (threads == 64 and per_cu < 8):
                            continue
                        for unroll in (1, 2, 4):
                            _C.loss_tuning(WS, fp32, threads, per_cu, unroll, 256)
                            _C.loss_layout(WS, fp32, per_wave, window, 1)
                            walk, red = timed(ws, sets, 15, split=True)
                            shapes.append((walk + red, per_wave, window, threads, per_cu, unroll, walk, red))
        for per_wave in (0, 1):
            for window in (0, 1):
                sub = sorted(r for r in shapes if r[1:3] == (per_wave, window))
                for r in sub[:3] + sub[-1:]:
                    say('shape %s forward ws (2 launches): %6.2f us  per_wave %d  window %d  threads %4d  per_cu %2d  unroll %d  = walk %6.2f + reduce %5.2f  -> %.3f' % ((name,) + r + (frac(r[0], 1),)))
        reset()
        del sets
    _C.profile_enable(False)
    say('ALL AGREE' if ok else 'DISAGREEMENT')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())

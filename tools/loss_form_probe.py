#!/usr/bin/env python
"""A/B of the two arithmetic forms of the loss kernels' classification walk (include/odtk_hip.h: odtk_debug_loss_form;
csrc/loss.hpp focal_term / focal_plain) at the training step's own sizes (BASELINE config 3 per GPU: 2 images of 800x1280,
five levels, 9 anchors x 80 classes).

1. agreement: form 1 against form 0 on the same inputs (sums relative, gradients relative to the largest one), against the
   torch expression in float64 on one small level, and BIT equality where every vector must take the element-by-element path
   (a logit above 64 in every vector; NaN / +-inf logits);
2. time: forward (atomics) and backward, fp32 and bf16 heads, channels_last and NCHW, event-timed by the library over calls that
   rotate through three input sets (loss_probe.py's method).
Every line is flushed as it is produced: a run cut short keeps what it measured."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
from odtk import _C
from odtk import loss as L

SIZES = [(100, 160), (50, 80), (25, 40), (13, 20), (7, 10)]
B, A, C, NB = 2, 9, 80, 4


def say(*a):
    print(*a, flush=True)


def make_set(dtype, seed, channels_last=True, sizes=SIZES):
    g = torch.Generator(device='cuda').manual_seed(seed)
    cls, box, depth, tgt = [], [], [], []
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    for h, w in sizes:
        c = (torch.randn(B, A * C, h, w, device='cuda', generator=g) * 1.5 - 4.0).to(dtype).contiguous(memory_format=fmt)
        b = (torch.randn(B, A * NB, h, w, device='cuda', generator=g) * 0.3).to(dtype).contiguous(memory_format=fmt)
        u = torch.rand(B, A, 1, h, w, device='cuda', generator=g)
        d = torch.zeros_like(u)
        d[u < 0.02] = -1.0
        fg = u > 0.995
        d[fg] = torch.randint(1, C + 1, (int(fg.sum()),), device='cuda', generator=g).float()
        cls.append(c); box.append(b); depth.append(d.contiguous())
        tgt.append((torch.randn(B, A, NB, h, w, device='cuda', generator=g) * 0.3).contiguous())
    return cls, box, depth, tgt


def both(s, gc, gb):
    fwd = _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, reproducible=True)
    bwd = _C.retina_loss_levels_backward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, gc, gb)
    return fwd, [t for pair in bwd for t in pair]


def timed(fn, sets, iters):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    _C.profile_collect()
    for i in range(iters):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    ms, n = _C.profile_collect()['retina_loss_kernel']
    assert n == iters, (n, iters)
    return ms * 1e3 / iters


def pmc_mode():
    """A short, fixed sequence for a rocprofv3 --pmc pass (tools/loss_pmc.sh): per dtype 3 warm-up + 10 measured launches of
    the forward (atomics form), the forward through the workspace and the backward, inputs rotated through three sets."""
    gc = torch.full((len(SIZES),), 0.37, device='cuda')
    gb = torch.full((len(SIZES),), -1.9, device='cuda')
    for dtype in (torch.float32, torch.bfloat16):
        sets = [make_set(dtype, 10 + i, True) for i in range(3)]
        for i in range(13):
            s = sets[i % 3]
            _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11)
        for i in range(13):
            s = sets[i % 3]
            _C.retina_loss_levels_backward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, gc, gb)
        torch.cuda.synchronize()
        del sets
    return 0


def main():
    if '--pmc' in sys.argv:
        return pmc_mode()
    gc = torch.full((len(SIZES),), 0.37, device='cuda')
    gb = torch.full((len(SIZES),), -1.9, device='cuda')
    ok = True
    # ---- 1. agreement ----
    for dtype, name in ((torch.float32, 'fp32'), (torch.bfloat16, 'bf16'), (torch.float16, 'fp16')):
        for cl in (True, False):
            s = make_set(dtype, 3, cl)
            _C.loss_form(0)
            f0, g0 = both(s, gc, gb)
            _C.loss_form(1)
            f1, g1 = both(s, gc, gb)
            rel = float(((f1 - f0).abs() / f0.abs().clamp_min(1e-300)).max())
            worst = max(float((a.float() - b.float()).abs().max()) / max(float(b.float().abs().max()), 1e-30) for a, b in zip(g1, g0))
            good = rel <= 5e-7 and worst <= (2e-6 if dtype == torch.float32 else 1e-2)
            ok &= good
            say('agree %s %s: sums rel %.2e, gradients %.2e of the largest  %s' % (name, 'nhwc' if cl else 'nchw', rel, worst, 'ok' if good else 'BAD'))
    # against float64 torch on one small level, both forms
    small = [(20, 32)]
    s = make_set(torch.float32, 5, True, small)
    cls64 = s[0][0].double().view(B, A, C, *small[0])
    dep = s[2][0]
    tgt = torch.zeros_like(cls64)
    idx = (dep.long() - 1).clamp_min(0)
    tgt.scatter_(2, idx.expand(B, A, 1, *small[0]), (dep > 0).double())
    want = float((L.FocalLoss(0.25, 2.0)(cls64, tgt) * (dep >= 0).double()).sum())
    for form in (0, 1):
        _C.loss_form(form)
        got = float(_C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, reproducible=True)[0, 0])
        good = abs(got - want) <= 1e-6 * abs(want)
        ok &= good
        say('float64 truth, form %d: %.10g vs %.10g  rel %.2e  %s' % (form, got, want, abs(got - want) / abs(want), 'ok' if good else 'BAD'))
    # every vector through the element path: bit equality between the forms; special values
    for cl in (True, False):
        s = make_set(torch.float32, 7, cl, [(25, 40)])
        flat = s[0][0].view(-1) if not cl else s[0][0].permute(0, 2, 3, 1).reshape(-1)
        flat[::4] = 70.0                                     # one logit beyond kPlainMax in every 16-byte vector
        _C.loss_form(0)
        f0, g0 = both(s, gc[:1], gb[:1])
        _C.loss_form(1)
        f1, g1 = both(s, gc[:1], gb[:1])
        good = torch.equal(f0, f1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
        ok &= good
        say('big logit in every vector (%s): forms bit-equal %s' % ('nhwc' if cl else 'nchw', 'ok' if good else 'BAD'))
        flat[1::16] = float('inf'); flat[2::32] = float('-inf'); flat[5::64] = float('nan')
        _C.loss_form(0)
        f0, g0 = both(s, gc[:1], gb[:1])
        _C.loss_form(1)
        f1, g1 = both(s, gc[:1], gb[:1])
        same = lambda a, b: torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float(), 1e30, 2e30, -2e30), torch.nan_to_num(b.float(), 1e30, 2e30, -2e30))
        # vectors with a NaN / +inf take the element path in both forms; -inf is a legal plain-path input (term 0, gradient 0)
        good = same(f0, f1) and all(same(a, b) for a, b in zip(g0, g1))
        ok &= good
        say('NaN / +-inf logits (%s): forms agree %s' % ('nhwc' if cl else 'nchw', 'ok' if good else 'BAD'))
    # ---- 2. time ----
    _C.profile_enable(True, ('retina_loss_kernel', 'loss_reduce_kernel'))   # (the workspace form's second launch has its own id since round 6)
    logits = sum(B * A * C * h * w for h, w in SIZES)
    for dtype, name, cl in ((torch.float32, 'fp32', True), (torch.bfloat16, 'bf16', True), (torch.float32, 'fp32', False)):
        sets = [make_set(dtype, 10 + i, cl) for i in range(3)]
        elem = 4 if dtype == torch.float32 else 2
        for which, label in ((0, 'forward'), (1, 'backward')):
            def call(s, which=which):
                if which == 1:
                    return _C.retina_loss_levels_backward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, gc, gb)
                return _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11)
            row = []
            for form in (0, 1, 0, 1):
                _C.loss_form(form)
                row.append(timed(call, sets, 30))
            alg = logits * elem * (2 if which else 1)
            say('time %s %s %-8s form 0: %6.2f / %6.2f us   form 1: %6.2f / %6.2f us   -> %.3f -> %.3f of 8 TB/s'
                % (name, 'nhwc' if cl else 'nchw', label, row[0], row[2], row[1], row[3],
                   alg / min(row[0], row[2]) / 1e3 / 8000, alg / min(row[1], row[3]) / 1e3 / 8000))
        del sets
    # ---- 3. where the fp32 forward's time goes: ablations of form 1 (wrong sums on purpose) and its launch shapes ----
    # (the ablation forms exist only in a -DODTK_LOSS_ABLATIONS library:  make -C retinanet-examples_amd/csrc ablations, then
    #  ODTK_HIP_LIBRARY=build_ablate/libodtk_hip.so python tools/loss_form_probe.py ...)
    if _C.library().odtk_debug_loss_form(2) != 0:
        say('ablations skipped: this library was built without -DODTK_LOSS_ABLATIONS')
        return
    sets = [make_set(torch.float32, 10 + i, True) for i in range(3)]
    fwd = lambda s: _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11)
    for form, what in ((1, 'form 1'), (2, 'no depth gather'), (3, 'no arithmetic'), (4, 'no index arithmetic, no depth gather'),
                       (6, 'no box-delta walk'), (7, 'no logit walk'), (1, 'form 1 again'), (6, 'no box-delta walk again'), (7, 'no logit walk again')):
        _C.loss_form(form)
        say('ablation fp32 nhwc forward, %-38s %6.2f us' % (what + ':', timed(fwd, sets, 30)))
    _C.loss_form(1)
    rows = []
    for threads in (256, 512, 1024):
        for per_cu in (1, 2, 4, 8):
            for unroll in (1, 2, 4):
                _C.loss_tuning(0, True, threads, per_cu, unroll, 64)
                rows.append((timed(fwd, sets, 20), threads, per_cu, unroll))
    _C.loss_tuning(0, True, 512, 1, 4, 64)
    rows.sort()
    for r in rows[:6] + rows[-2:]:
        say('shape fp32 nhwc forward form 1: %6.2f us  threads %4d  per_cu %d  unroll %d' % r)
    for box_blocks in (16, 64, 256, 1024, 4096):
        _C.loss_tuning(0, True, 512, 1, 4, box_blocks)
        say('box workgroups per level %4d: fp32 nhwc forward form 1 %6.2f us' % (box_blocks, timed(fwd, sets, 20)))
    _C.loss_tuning(0, True, 512, 1, 4, 64)
    ws = lambda s: _C.retina_loss_levels_forward(s[0], s[1], s[2], s[3], 0.25, 2.0, 0.11, reproducible=True)
    rows = []
    for threads in (256, 512):
        for per_cu in (2, 4, 8, 16):
            for unroll in (1, 2, 4):
                _C.loss_tuning(2, True, threads, per_cu, unroll, 256)
                for s_ in sets:
                    ws(s_)
                torch.cuda.synchronize()
                _C.profile_collect()
                for i in range(20):
                    ws(sets[i % 3])
                torch.cuda.synchronize()
                got = _C.profile_collect()
                rows.append(((got['retina_loss_kernel'][0] + got['loss_reduce_kernel'][0]) * 1e3 / 20, threads, per_cu, unroll))
    _C.loss_tuning(2, True, 256, 4, 2, 256)          # (the default since round 6)
    rows.sort()
    for r in rows[:4] + rows[-1:]:
        say('shape fp32 nhwc forward (workspace, 2 launches) form 1: %6.2f us  threads %4d  per_cu %2d  unroll %d' % r)
    _C.profile_enable(False)
    _C.loss_form(_C.LOSS_FORM_DEFAULT)
    say('ALL AGREE' if ok else 'DISAGREEMENT')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python
"""Which op of the inference engine is not run-to-run reproducible?  (at batch 1 two eager runs of
Model.forward differ in threshold-grazing detections.)  Run 1 records every engine convolution call (method, inputs,
output); each call is then replayed on the recorded inputs a few times and compared bit for bit with the recorded output.
Prints the calls that differ, with shape, kind (hipBLASLt GEMM / MIOpen conv) and the size of the difference.

    python tools/determinism_probe.py --batch 1
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch

torch.backends.cudnn.benchmark = True
from odtk import _C, fused
from odtk.model import Model

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--backbone', default='ResNet50FPN')
ap.add_argument('--replays', type=int, default=4)
ap.add_argument('--deterministic-flag', action='store_true', help='torch.backends.cudnn.deterministic = True (MIOpen)')
args = ap.parse_args()
if args.deterministic_flag:
    torch.backends.cudnn.deterministic = True

torch.manual_seed(0)
m = Model(args.backbone)
m.initialize(None)
m = m.cuda().to(memory_format=torch.channels_last).eval()
x = torch.randn(args.batch, 3, 800, 1280, generator=torch.Generator().manual_seed(0)).cuda().contiguous(memory_format=torch.channels_last)
eng = m.inference_engine(torch.bfloat16)
eng.level_streams = False
with torch.no_grad():
    for _ in range(3):
        eng.heads_without_last_bias(x)
torch.cuda.synchronize()

calls = []
names = {mod: name for name, mod in eng.named_modules()}


def wrap(mod, method):
    orig = getattr(mod, method)

    def rec(*a):
        out = orig(*a)
        calls.append((mod, method, tuple(t.clone() if t is not None else None for t in a), out.clone()))
        return out
    setattr(mod, method, rec)
    return orig


origs = []
for mod in eng.modules():
    if isinstance(mod, fused._Conv):
        for method in ('forward', 'conv_only', 'conv_then_pool'):
            origs.append((mod, method, wrap(mod, method)))
with torch.no_grad():
    heads1 = eng.heads_without_last_bias(x)
torch.cuda.synchronize()
for mod, method, orig in origs:
    setattr(mod, method, orig)

with torch.no_grad():
    heads2 = eng.heads_without_last_bias(x)
torch.cuda.synchronize()
same_heads = [bool(torch.equal(a, b)) for a, b in zip(heads1[0] + heads1[1], heads2[0] + heads2[1])]
print('batch %d: whole engine, run 1 vs run 2, cls x5 + box x5 equal: %s' % (args.batch, same_heads))

bad = 0
with torch.no_grad():
    for mod, method, a, out in calls:
        diffs = []
        for _ in range(args.replays):
            again = getattr(mod, method)(*[t.clone() if t is not None else None for t in a])
            if not torch.equal(again, out):
                diffs.append(float((again.float() - out.float()).abs().max()))
        if diffs:
            bad += 1
            kind = 'hipBLASLt GEMM' if (mod.pointwise and method == 'forward' and _C.gemm_available()) else 'MIOpen conv'
            print('  NOT reproducible: %-28s %-14s in %s w %s stride %s -> out %s | %d of %d replays differ, max |d| %.3g | %s'
                  % (names.get(mod, '?'), method, tuple(a[0].shape), tuple(mod.weight.shape), tuple(mod.stride), tuple(out.shape),
                     len(diffs), args.replays, max(diffs), kind))
print('batch %d: %d engine conv calls recorded, %d not bit-reproducible on identical inputs' % (args.batch, len(calls), bad))

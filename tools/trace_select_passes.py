#!/usr/bin/env python
"""Timeline of one decode call on synthetic sparse-realistic bf16 logits (debug trace, odtk_debug_set_trace): when
each selection pass of segment (P3, image 0) starts, what its phases cost, and the gaps between the launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
import torch
from odtk import _C, box, synthetic

kind = sys.argv[1] if len(sys.argv) > 1 else 'sparse'
g = torch.Generator(device='cuda').manual_seed(1234)
strides = [8, 16, 32, 64, 128]
cls, dl = [], []
for (h, w) in synthetic.level_shapes(800, 1280, strides):
    c = torch.randn((8, 720, h, w), generator=g, device='cuda') * synthetic.SIGMA[kind] + synthetic.LOGIT_PRIOR
    d = torch.randn((8, 36, h, w), generator=g, device='cuda') * 0.2
    cls.append(c.bfloat16().contiguous(memory_format=torch.channels_last))
    dl.append(d.bfloat16().contiguous(memory_format=torch.channels_last))
anchors = {s: box.generate_anchors(s, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]) for s in strides}
run = lambda: box.detect(cls, dl, strides, anchors, 0.05, 1000, 0.5, 100, False, logits=True)
for _ in range(3):
    run()
trace = torch.zeros(8192, dtype=torch.int64, device='cuda')
_C.library().odtk_debug_set_trace(trace.data_ptr())
run(); torch.cuda.synchronize()
_C.library().odtk_debug_set_trace(None)
t = trace.cpu()
us = lambda a, b: (int(b) - int(a)) / 100.0
for seg in (0, 8, 16):            # (P3, img 0), (P4, img 0), (P5, img 0)
    rows = [t[1024 + (p * 64 + seg) * 8:1024 + (p * 64 + seg) * 8 + 8] for p in range(3)]
    dec = t[seg * 8:seg * 8 + 8]
    t0 = int(rows[0][0]) or int(dec[0])
    print('segment %d: candidates %d, n_sort %d' % (seg, int(dec[5]), int(dec[6])))
    for p, r in enumerate(rows):
        if int(r[0]):
            print('  pass %d: starts at %7.2f us | counts %5.2f | state %5.2f | walk %5.2f | flush %5.2f | ends at %7.2f'
                  % (p, us(t0, r[0]), us(r[0], r[1]), us(r[1], r[2]), us(r[2], r[3]), us(r[3], r[4]), us(t0, r[4])))
    print('  select_decode: starts at %7.2f | gather %5.2f | lds narrow %5.2f | sort %5.2f | decode %5.2f | ends at %7.2f'
          % (us(t0, dec[0]), us(dec[0], dec[1]), us(dec[1], dec[2]), us(dec[2], dec[3]), us(dec[3], dec[4]), us(t0, dec[4])))

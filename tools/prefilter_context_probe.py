#!/usr/bin/env python
"""Why is the prefilter slower inside the end-to-end step (~61 us) than back to back (~48 us)?
Times `prefilter_scan_kernel` (library hipEvent hooks) on the same bf16 channels_last logits when the
launch is preceded by different kinds of work:

  A  nothing (back-to-back detect calls: the 245 MB input can sit in the 256 MiB Infinity Cache)
  B  a 512 MB device copy         (cache flushed, dirty lines left behind)
  C  ~5 ms of bf16 GEMMs          (MFMA load: power / clock state, small footprint)
  D  the head tensors re-written  (producer -> consumer, as after the head convolutions)

    python tools/prefilter_context_probe.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from odtk import _C, box, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(1234)
    strides = (8, 16, 32, 64, 128)
    batch, sigma = 8, 0.6225
    cls, dl = [], []
    for (h, w) in synthetic.level_shapes(800, 1280, strides):
        c = torch.randn((batch, 720, h, w), generator=g, device=dev) * sigma + synthetic.LOGIT_PRIOR
        d = torch.randn((batch, 36, h, w), generator=g, device=dev) * 0.2
        cls.append(c.bfloat16().contiguous(memory_format=torch.channels_last))
        dl.append(d.bfloat16().contiguous(memory_format=torch.channels_last))
    src = [c.clone() for c in cls]
    anchors = {s: box.generate_anchors(s, [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]) for s in strides}
    big_a = torch.empty(256 << 20, dtype=torch.bfloat16, device=dev)
    big_b = torch.zeros(256 << 20, dtype=torch.bfloat16, device=dev)
    ma = torch.randn(4096, 4096, device=dev).bfloat16()
    mb = torch.randn(4096, 4096, device=dev).bfloat16()

    def detect():
        return box.detect(cls, dl, list(strides), anchors, 0.05, 1000, 0.5, 100, False, logits=True)

    def copy512():
        big_a.copy_(big_b)

    def gemms():
        for _ in range(40):
            torch.mm(ma, mb)

    def rewrite():
        for c, s in zip(cls, src):
            c.copy_(s)

    def rewrite_small_last():
        for c, s in zip(cls[::-1], src[::-1]):
            c.copy_(s)

    scenarios = [('A back-to-back', None), ('B 512MB copy before', copy512), ('C 5ms GEMMs before', gemms),
                 ('D heads rewritten P3..P7', rewrite), ('E heads rewritten P7..P3', rewrite_small_last),
                 ('F GEMMs then heads rewritten', lambda: (gemms(), rewrite()))]
    res = {}
    for name, pre in scenarios:
        for _ in range(3):
            if pre:
                pre()
            detect()
        torch.cuda.synchronize()
        _C.profile_enable(True)
        _C.profile_collect()
        for _ in range(20):
            if pre:
                pre()
            detect()
        torch.cuda.synchronize()
        _C.profile_enable(False)
        prof = _C.profile_collect()
        res[name] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]}
        print(name, res[name], flush=True)
    print(json.dumps(res))


if __name__ == '__main__':
    main()

"""tests/golden/nms_trained_scenes_{unique,ties}.npz: the NMS stage on what a TRAINED detector hands it (build container only).

    python -m oracle.gen_golden_trained_nms gpurun_out/r6c2/trained_postproc_inputs.npz      # needs /root/reference

TEST INFRASTRUCTURE ONLY.  The input file is written on the GPU box by `tools/trained_ap.py --save-postproc-inputs`: the
decode_levels output (5 x 1000 candidates per image, 16 images) of the bf16 engine of a ResNet18FPN trained by the product's loop
on odtk/scenes.py -- clusters of dozens of overlapping same-class candidates around every object, the regime the synthetic
fixtures (independent random boxes) do not have: the NMS examines 50-100 % of its candidates here, and 16-bit scores tie in
the hundreds (1411 candidates on 314 distinct scores).

  *_unique  the scores nudged apart by whole ulps (synthetic.make_unique_scores), expected outputs = the REFERENCE's own
            odtk/box.py nms (CPU branch, box.py:312-367) through oracle/ref_loader.py: a reference-generated fixture like
            the other nms_*.npz (kind 'nms': picked up by the existing CPU and GPU fixture tests).
  *_ties    the scores as the engine produced them.  The reference's CPU branch sorts with an unstable torch.sort, so its
            output on ties is implementation-defined (it keeps other boxes than its own CUDA path's stable radix sort on 13
            of these 16 images); expected outputs = the canonical rule (score desc, position asc; oracle/box_oracle.py),
            which is what the reference's CUDA path produces (nms.cu:136-137) -- kind 'nms_ties'.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'retinanet-examples_amd'))

from oracle import box_oracle, ref_loader  # noqa: E402
from odtk import synthetic                 # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def main(path):
    d = np.load(path)
    scores, boxes, classes = (torch.from_numpy(d[k]) for k in ('scores', 'boxes', 'classes'))
    nms, det = float(d['nms']), int(d['detections'])
    # ties: canonical rule; the GPU's own detections of the same batch (saved beside the inputs) must already equal it
    tie_out = box_oracle.nms(scores, boxes, classes, nms, det)
    for got, key in zip(tie_out, ('det_scores', 'det_boxes', 'det_classes')):
        assert torch.equal(got, torch.from_numpy(d[key])), 'the engine run that saved %s disagrees with the oracle on %s' % (path, key)
    np.savez_compressed(os.path.join(GOLDEN, 'nms_trained_scenes_ties.npz'), kind='nms_ties', scores=d['scores'], boxes=d['boxes'],
                        classes=d['classes'], nms=np.float64(nms), detections=det, out_scores=tie_out[0].numpy(),
                        out_boxes=tie_out[1].numpy(), out_classes=tie_out[2].numpy())
    ref_tie = ref_loader.ref_nms(scores, boxes, classes, nms, det)
    differ = int(sum(not torch.equal(a[i], b[i]) for i in range(scores.shape[0]) for a, b in [(ref_tie[1], tie_out[1])]))
    print('ties: %d candidates on %d distinct scores in the densest image; reference CPU branch (unstable sort) differs from the '
          'canonical rule on %d of %d images' % (int((scores > 0).sum(1).max()), int(torch.unique(scores[(scores > 0).sum(1).argmax()]).numel()) - 1,
                                                differ, scores.shape[0]))
    # unique: the reference's own CPU nms
    unique = synthetic.make_unique_scores(scores, 0.0) * (scores > 0)      # (padding stays 0)
    for b in range(unique.shape[0]):
        v = unique[b][unique[b] > 0]
        assert v.unique().numel() == v.numel()
    out = ref_loader.ref_nms(unique, boxes, classes, nms, det)
    mine = box_oracle.nms(unique, boxes, classes, nms, det)
    assert all(torch.equal(a, b) for a, b in zip(out, mine)), 'oracle != reference on tie-free input'
    np.savez_compressed(os.path.join(GOLDEN, 'nms_trained_scenes_unique.npz'), kind='nms', scores=unique.numpy(), boxes=d['boxes'],
                        classes=d['classes'], nms=np.float64(nms), detections=det, out_scores=out[0].numpy(),
                        out_boxes=out[1].numpy(), out_classes=out[2].numpy())
    print('unique: kept per image', (out[0] > 0).sum(1).tolist(), 'of', (unique > 0).sum(1).tolist())


if __name__ == '__main__':
    main(sys.argv[1])

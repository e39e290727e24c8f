"""The `odtk` command line on an MI355X: what a user of the reference types, end to end on the HIP path --
`train` (HIP target assignment + fused loss, fp16 autocast + GradScaler as the reference's default) and `infer`
(uint8 upload, table normalisation on the device, BN-folded engine in bf16 and fp32, fused post-processing,
detections JSON, AP) on the committed five-image data set."""
import json
import os

import numpy as np
import pytest
import torch

from odtk import data as D
from odtk import main as cli
from odtk.model import Model

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data')
ANN = os.path.join(DATA, 'annotations.json')


def test_device_side_normalisation_is_bit_identical_to_the_host():
    ds = D.CocoDataset(DATA, resize=128, max_size=200, stride=32, annotations=ANN, training=False)
    packed, _, _ = ds.collate_fn([ds[i] for i in range(5)])
    host = D.normalise_batch(packed)
    for dtype in (torch.float32, torch.bfloat16):
        dev = D.normalise_batch(packed.cuda(), dtype=dtype)
        assert dev.is_cuda and dev.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(dev.cpu(), host.to(dtype))


def test_train_then_infer_through_the_command_line(tmp_path, capsys):
    path = str(tmp_path / 'tiny.pth')
    common = ['--annotations', ANN, '--images', DATA, '--backbone', 'ResNet18FPN', '--classes', '3', '--batch', '2',
              '--resize', '128', '--max-size', '160', '--jitter', '96', '128', '--warmup', '2', '--lr', '0.001', '--workers', '0']
    done, _ = cli.main(['train', path, '--iters', '4'] + common)                     # mixed precision (the default)
    out = capsys.readouterr().out
    assert done == 4 and '[4/4] focal loss:' in out and 'precision: mixed' in out
    done, _ = cli.main(['train', path, '--iters', '6', '--full-precision'] + common)   # resume in fp32
    assert done == 6 and Model.load(path)[1]['iteration'] == 6
    capsys.readouterr()

    model, _ = Model.load(path)
    with torch.no_grad():
        model.cls_head[-1].bias.fill_(0.0)                                            # lift the class prior: detections exist
    lifted = str(tmp_path / 'lifted.pth')
    model.save({'path': lifted})
    results = {}
    for tag, extra in (('mixed', []), ('full', ['--full-precision'])):
        out_file = str(tmp_path / (tag + '.json'))
        stats = cli.main(['infer', lifted, '--images', DATA, '--annotations', ANN, '--output', out_file, '--batch', '2',
                          '--resize', '128', '--max-size', '160', '--workers', '0'] + extra)
        text = capsys.readouterr().out
        assert 'device: 1 GPU' in text and 'Average Precision  (AP) @[ IoU=0.50:0.95' in text
        assert isinstance(stats, np.ndarray) and stats.shape == (12,)
        results[tag] = json.load(open(out_file))
    for doc in results.values():
        assert set(doc) == {'annotations', 'images', 'categories'}
        assert {d['image_id'] for d in doc['annotations']} <= {100, 103, 106, 109, 112}
        assert all(d['category_id'] in (7, 3, 11) and 0.05 <= d['score'] <= 1 and len(d['bbox']) == 4 for d in doc['annotations'])
        assert len(doc['annotations']) >= 5


def test_rotated_infer_reports_an_ap(tmp_path, capsys):
    """`odtk infer --rotated-bbox` with ground truth: the reference scores the corner polygons ('segm', infer.py:166); round 3
    provides that evaluation (exact polygon IoU, odtk/cocoeval.py) -- before, a rotated model got no AP at all."""
    torch.manual_seed(0)
    model = Model('ResNet18FPN', classes=3, rotated_bbox=True)
    model.initialize(None)
    with torch.no_grad():
        model.cls_head[-1].bias.fill_(0.0)                                            # detections exist
    path = str(tmp_path / 'rotated.pth')
    model.save({'path': path})
    out_file = str(tmp_path / 'rotated.json')
    stats = cli.main(['infer', path, '--images', DATA, '--annotations', os.path.join(DATA, 'annotations_rotated.json'),
                      '--output', out_file, '--batch', '2', '--resize', '128', '--max-size', '160', '--workers', '0', '--rotated-bbox'])
    text = capsys.readouterr().out
    assert 'exact polygon IoU' in text and 'Average Precision  (AP) @[ IoU=0.50:0.95' in text
    assert isinstance(stats, np.ndarray) and stats.shape == (12,) and np.all((stats >= 0) | (stats == -1.0))
    doc = json.load(open(out_file))
    assert len(doc['annotations']) >= 5
    assert all(len(d['bbox']) == 5 and len(d['segmentation']) == 1 and len(d['segmentation'][0]) == 8 for d in doc['annotations'])

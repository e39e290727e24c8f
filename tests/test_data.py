"""The input side of the path: odtk/data.py against what the REFERENCE's data.py produced for the committed
five-image data set (tests/golden/data/, made by oracle/gen_golden_data.py from the unmodified reference),
bit for bit -- pixels after the device-side table normalisation, ids, ratios, targets, batch padding."""
import os
import random

import numpy as np
import pytest
import torch

from odtk import data as D

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data')
ANN = os.path.join(HERE, 'annotations.json')
ANN_ROT = os.path.join(HERE, 'annotations_rotated.json')


@pytest.fixture(scope='module')
def expected():
    return np.load(os.path.join(HERE, 'expected.npz'))


def _normalised(pixels, stride):
    """One item the way the reference returns it: normalised CHW float, zero-padded to the stride."""
    h, w = pixels.shape[:2]
    up = lambda d: d + (stride - d % stride) % stride
    assert pixels.shape[2] == 4 and bool((pixels[..., 3] == 255).all())       # R, G, B, valid
    packed = torch.zeros(1, up(h), up(w), 4, dtype=torch.uint8)
    packed[0, :h, :w] = pixels
    return D.normalise_batch(packed)[0]


def test_normalisation_table_is_the_reference_arithmetic():
    table = D.normalisation_table()
    v = torch.arange(256, dtype=torch.uint8)
    for c, (m, s) in enumerate(zip(D.MEAN, D.STD)):
        want = v.float().div(255)
        want.sub_(m).div_(s)                                     # reference data.py:112-117
        assert torch.equal(table[c], want)


def test_inference_items_and_batch_equal_the_reference(expected):
    ds = D.CocoDataset(HERE, resize=128, max_size=200, stride=32, annotations=ANN, training=False)
    assert len(ds) == 5
    items = [ds[i] for i in range(5)]
    for i, (pixels, image_id, ratio) in enumerate(items):
        assert pixels.dtype == torch.uint8
        got = _normalised(pixels, 32)
        want = torch.from_numpy(expected['infer_pixels_%d' % i])
        assert got.shape == want.shape
        assert torch.equal(got.contiguous(), want), 'image %d' % i
        assert image_id == int(expected['infer_id_%d' % i])
        assert ratio == float(expected['infer_ratio_%d' % i])
    packed, ids, ratios = ds.collate_fn(items[:4])
    assert packed.dtype == torch.uint8 and packed.shape[-1] == 4
    batch = D.normalise_batch(packed)
    assert batch.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(batch.contiguous(), torch.from_numpy(expected['infer_batch']))
    assert not torch.signbit(batch).logical_and(batch == 0).any()                 # pad is +0.0, like F.pad
    assert ids.dtype == torch.int32 and torch.equal(ids, torch.from_numpy(expected['infer_batch_ids']))
    assert torch.equal(ratios, torch.from_numpy(expected['infer_batch_ratios']))


@pytest.mark.parametrize('tag,cls,ann,extra', [
    ('train', D.CocoDataset, ANN, {}),
    ('rtrain', D.RotatedCocoDataset, ANN_ROT, {}),
    ('rabs', D.RotatedCocoDataset, ANN_ROT, {'absolute_angle': True}),
])
def test_seeded_training_items_equal_the_reference(expected, tag, cls, ann, extra):
    ds = cls(HERE, resize=[96, 160], max_size=220, stride=32, annotations=ann, training=True, rotate_augment=True, **extra)
    random.seed(1234)
    items = [ds[i % len(ds)] for i in range(10)]
    for i, (pixels, target) in enumerate(items):
        want = torch.from_numpy(expected['%s_pixels_%d' % (tag, i)])
        assert torch.equal(_normalised(pixels, 32).contiguous(), want), 'item %d' % i
        assert torch.equal(target, torch.from_numpy(expected['%s_target_%d' % (tag, i)])), 'item %d' % i
    packed, targets = ds.collate_fn(items[:5])
    assert torch.equal(D.normalise_batch(packed).contiguous(), torch.from_numpy(expected['%s_batch' % tag]))
    assert torch.equal(targets, torch.from_numpy(expected['%s_batch_targets' % tag]))
    assert targets.shape[-1] == ds.box_fields + 1


def test_index_follows_file_order_and_skips_nothing():
    index = D.CocoIndex(ANN)
    assert index.getCatIds() == [7, 3, 11]                        # file order, not sorted
    assert list(index.imgs) == [100, 103, 106, 109, 112]
    assert index.getAnnIds(imgIds=106) == []                      # the image without annotations
    first = index.loadAnns(index.getAnnIds(imgIds=100))
    assert [a['image_id'] for a in first] == [100] * len(first) and len(first) >= 2
    res = index.loadRes([{'image_id': 100, 'category_id': 7, 'score': 0.5, 'bbox': [1.0, 2.0, 3.0, 4.0]}])
    (ann,) = res.loadAnns(res.getAnnIds(imgIds=100))
    assert ann['area'] == 12.0 and ann['id'] == 1 and ann['iscrowd'] == 0
    with pytest.raises(AssertionError):
        index.loadRes([{'image_id': 5, 'category_id': 7, 'score': 0.5, 'bbox': [1, 2, 3, 4]}])


def test_iterator_yields_channels_last_batches_on_the_device():
    it = D.DataIterator(HERE, 128, 200, 2, 32, 1, ANN, training=False, num_workers=0, device='cpu')
    assert len(it) == 3 and 'loader: pytorch' in repr(it)
    seen = []
    for images, ids, ratios in it:
        assert images.dtype == torch.float32 and images.shape[1] == 3
        assert images.shape[2] % 32 == 0 and images.shape[3] % 32 == 0
        assert images.is_contiguous(memory_format=torch.channels_last)
        assert ratios.shape == (images.shape[0], 1, 1)
        seen += ids.tolist()
    assert seen == [100, 103, 106, 109, 112]
    train = D.RotatedDataIterator(HERE, [96, 128], 200, 2, 32, 1, ANN_ROT, training=True, num_workers=0, device='cpu',
                                  absolute_angle=True)
    images, targets = next(iter(train))
    assert targets.shape[0] == 2 and targets.shape[2] == 6
    with pytest.raises(RuntimeError, match='multiple of the number of GPUs'):
        D.DataIterator(HERE, 128, 200, 3, 32, 2, ANN, num_workers=0, device='cpu', rank=0)


def test_sharded_iterators_cover_the_data_set_once():
    ids = []
    for rank in range(2):
        it = D.DataIterator(HERE, 128, 200, 2, 32, 2, ANN, num_workers=0, device='cpu', rank=rank)
        assert it.dataloader.batch_size == 1
        for _, batch_ids, _ in it:
            ids += batch_ids.tolist()
    assert sorted(set(ids)) == [100, 103, 106, 109, 112] and len(ids) == 6      # DistributedSampler pads to 2 x 3


def test_colour_jitter_runs_and_keeps_geometry():
    ds = D.CocoDataset(HERE, resize=96, max_size=200, stride=32, annotations=ANN, training=True,
                       augment_brightness=0.2, augment_contrast=0.2, augment_hue=0.05, augment_saturation=0.2)
    plain = D.CocoDataset(HERE, resize=96, max_size=200, stride=32, annotations=ANN, training=True)
    random.seed(7)
    a, ta = ds[1]
    random.seed(7)
    b, tb = plain[1]
    assert a.shape == b.shape and torch.equal(ta, tb)
    assert not torch.equal(a, b)


def test_worker_processes_deliver_the_same_batches():
    """Batches collated inside loader workers are born in shared memory (`_batch_buffer`); same bytes as in-process."""
    plain = D.DataIterator(HERE, 128, 200, 2, 32, 1, ANN, training=False, num_workers=0, device='cpu')
    worked = D.DataIterator(HERE, 128, 200, 2, 32, 1, ANN, training=False, num_workers=2, device='cpu')
    for (a, ia, ra), (b, ib, rb) in zip(plain, worked):
        assert torch.equal(a, b) and torch.equal(ia, ib) and torch.equal(ra, rb)
    train = D.DataIterator(HERE, [96, 128], 200, 2, 32, 1, ANN, training=True, num_workers=2, device='cpu')
    assert train.dataloader.persistent_workers
    for _ in range(2):                                            # two epochs on the same workers
        assert sum(images.shape[0] for images, _ in train) == 5

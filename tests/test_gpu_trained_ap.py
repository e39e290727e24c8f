"""COCO AP of a TRAINED detector through every inference path, against the true boxes.

North star: "COCO mAP within +-0.1 of the reference" (reference README.md:33 quotes 0.358 for ResNet50FPN; the reference
measures it with pycocotools' COCOeval in odtk/infer.py:160-172).  Neither COCO nor weights exist in this image, so rounds
2-5 scored a random-init network whose last layer was ridge-fitted to plant objects -- an ill-conditioned proxy on which the
timed bf16 engine "lost" up to 3.7 AP points that nobody could interpret.  This file replaces it with what the repository
can actually make: a detector trained by the product's own loop (odtk/train.py: SGD, warm-up, frozen BN, HIP target
assignment + fused focal / smooth-L1 loss) on a learnable synthetic set (odtk/scenes.py: textured coloured rectangles of six
classes on noise, 24..300 px, i.e. the anchor ranges of P3..P6), scored on held-out scenes against the TRUE boxes with
odtk/cocoeval.py through

    reference            fp32 eager nn.Module graph (what the reference's PyTorch inference runs) + the CPU oracle's decode x5 + nms
                         (oracle/box_oracle.py, pinned to the reference's odtk/box.py)
    engine_fp32/16/bf16  Model.forward on the GPU = BN-folded engine + odtk_detect (HIP); bf16 is the path bench.py times,
                         fp16 is what `odtk infer` runs by default (the reference's mixed precision)
    eager_autocast_*     the eager graph under autocast + the HIP post-processing (models without a fused engine)

The GPU test trains a short schedule (about a minute) and asserts that the fp32 / fp16 engines are the reference's detector
(|AP - AP_reference| <= 0.005) and reports bf16; tools/trained_ap.py runs the long schedule on two training seeds and writes
profiles/r06_trained_ap.txt.  The CPU tests cover the scene generator and that the loop learns at all (loss falls)."""
import math
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

from odtk import scenes  # noqa: E402
from odtk.model import Model  # noqa: E402

CLASSES = 6
HELD_OUT_START = 10_000_000          # scene indices the training never reaches


def train_detector(seed=0, iterations=600, batch=16, size=512, backbone='ResNet18FPN', lr=0.01, device='cuda', verbose=False,
                   log_interval=50, finetune=0.25):
    """RetinaNet trained from its random initialisation on scenes 0 .. iterations * batch - 1 of `seed` by odtk/train.py's
    loop, fp32 like BASELINE's config 3.  -> (model in eval mode on `device`, [(iteration, focal, box, s/step, lr), ...]).

    Two phases, as the reference's users have them.  (1) What the ImageNet checkpoint stands for (reference resnet.py:20-22; no
    weights exist here): the first (1 - finetune) of the iterations with LIVE batch norm (`frozen_bn=False`) -- a random ResNet
    whose BN layers are frozen at the identity does not learn at the reference's learning rate (tried: the focal loss stays on
    its initial plateau of 1.1 for 500 steps, lr 0.01 diverges) -- and He-initialised head towers (the reference's std-0.01
    towers attenuate the signal 100 x and need its 90 000-iteration schedule).  (2) The reference's own recipe on that
    "pretrained" model: BN frozen at the learned statistics (gamma, beta, mean, var all non-trivial: what the engine's BN fold
    has to get right), lr / 10, the remaining iterations."""
    from odtk import train
    torch.manual_seed(seed)
    model = Model(backbone, classes=CLASSES)
    model.initialize(None)
    for tower in (model.cls_head, model.box_head):
        for layer in list(tower)[:-1]:
            if isinstance(layer, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(layer.weight, nonlinearity='relu')
    device = torch.device(device)
    history = []
    n_pre = iterations - int(iterations * finetune)
    phases = [(n_pre, lr, False, 0), (iterations - n_pre, lr * 0.1, True, n_pre * batch)]
    done = 0
    for n_it, phase_lr, frozen, first_scene in phases:
        if n_it <= 0:
            continue
        batches = scenes.SceneBatches(batch, size, size, classes=CLASSES, seed=seed, device=device, length=n_it, start=first_scene)
        offset = done
        train.train_batches(model, {}, batches, n_it, device, lr=phase_lr, warmup=min(200, max(n_it // 3, 1)) if not frozen else 0,
                            milestones=[int(n_it * 0.8)] if not frozen else [], mixed_precision=False, verbose=verbose,
                            log_interval=log_interval, frozen_bn=frozen,
                            on_report=lambda it, *rest: history.append((offset + it,) + rest))
        done += n_it
    model.eval()
    if device.type == 'cuda':
        model = model.to(memory_format=torch.channels_last)
    return model, history


def reference_detections(model, images):
    """fp32 eager graph + the oracle's post-processing on the CPU (reference model.py:140-165 on reference box.py:255-367)."""
    from oracle import box_oracle
    with torch.no_grad():
        cls_heads, box_heads = model.heads(images.float())
    strides = [images.shape[-1] // c.shape[-1] for c in cls_heads]
    for s in strides:
        model.level_anchors(s)
    return box_oracle.postprocess([c.float().sigmoid().contiguous().cpu() for c in cls_heads],
                                  [b.float().contiguous().cpu() for b in box_heads], strides, model.anchors,
                                  model.threshold, model.top_n, model.nms, model.detections)


def gpu_paths(model, images, progress=None):
    """progress (debug): called with the path's name after a device synchronisation behind every path -- a fault then surfaces
    at the path that caused it (tools/trained_ap.py --progress)."""
    out = {}

    def done(name):
        if progress is not None:
            torch.cuda.synchronize()
            progress(name)

    with torch.no_grad():
        out['engine_fp32'] = model(images)
        done('engine_fp32')
        with torch.autocast('cuda', dtype=torch.float16):
            out['engine_fp16'] = model(images)
        done('engine_fp16')
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out['engine_bf16'] = model(images)
        done('engine_bf16')
        model.fused_graph = False
        try:
            out['eager_fp32_hip_postproc'] = model(images)
            done('eager_fp32_hip_postproc')
            with torch.autocast('cuda', dtype=torch.float16):
                out['eager_autocast_fp16'] = model(images)
            done('eager_autocast_fp16')
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out['eager_autocast_bf16'] = model(images)
            done('eager_autocast_bf16')
        finally:
            model.fused_graph = True
    return out


def evaluate_paths(model, seed=0, images=64, batch=16, size=512, device='cuda', with_gpu_paths=True, progress=None):
    """Held-out scenes of `seed` -> {path: COCOeval.stats (12 numbers, AP first)} + {'examined': ...}."""
    from odtk.cocoeval import COCOeval
    from odtk.data import CocoIndex
    from odtk.infer import detections_to_coco
    held_out = scenes.SceneBatches(batch, size, size, classes=CLASSES, seed=seed, device=device, start=HELD_OUT_START)
    all_targets, dets = [], {}
    for step in range(images // batch):
        x, targets = held_out.batch_at(step)
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        ids = torch.arange(step * batch, (step + 1) * batch)
        all_targets.append(targets.cpu())
        found = {'reference': reference_detections(model, x)}
        if progress is not None:
            progress('step %d reference' % step)
        if with_gpu_paths:
            found.update(gpu_paths(model, x, None if progress is None else (lambda name, step=step: progress('step %d %s' % (step, name)))))
        for name, (s, b, c) in found.items():
            dets.setdefault(name, []).extend(detections_to_coco(s.float().cpu(), b.float().cpu(), c.float().cpu(), ids, torch.ones(batch)))
    truth = CocoIndex(dataset=scenes.coco_ground_truth(torch.cat(all_targets), 0, CLASSES))
    stats = {}
    for name, found in dets.items():
        if not found:
            stats[name] = [0.0] * 12
            continue
        ev = COCOeval(truth, truth.loadRes(found), 'bbox')
        ev.evaluate()
        ev.accumulate()
        stats[name] = [float(v) for v in ev.summarize(out=lambda line: None)]
    stats['_detections'] = {name: len(found) for name, found in dets.items()}
    stats['_objects'] = len(truth.dataset['annotations'])
    return stats


# ---- CPU ------------------------------------------------------------------------------------------------------------------------
def test_scenes_are_a_function_of_seed_and_index():
    a = scenes.SceneBatches(4, 128, 160, classes=CLASSES, seed=3)
    b = scenes.SceneBatches(4, 128, 160, classes=CLASSES, seed=3)
    xa, ta = a.batch_at(5)
    xb, tb = b.batch_at(5)
    assert torch.equal(ta, tb) and torch.equal(xa, xb)
    assert xa.shape == (4, 3, 128, 160) and ta.shape == (4, 6, 5)
    other = scenes.SceneBatches(4, 128, 160, classes=CLASSES, seed=4).batch_at(5)[1]
    assert not torch.equal(ta, other)
    # two ranks of a global batch see disjoint halves of the same scenes
    r0 = scenes.SceneBatches(4, 128, 160, classes=CLASSES, seed=3, rank=0, world=2).batch_at(5)[1]
    r1 = scenes.SceneBatches(4, 128, 160, classes=CLASSES, seed=3, rank=1, world=2).batch_at(5)[1]
    assert torch.equal(torch.cat([r0, r1]), ta)


def test_scene_targets_are_in_the_reference_format_and_inside_the_image():
    s = scenes.SceneBatches(8, 256, 256, classes=CLASSES, seed=1)
    x, t = s.batch_at(0)
    assert torch.isfinite(x).all()
    for rows in t:
        real = rows[rows[:, 4] >= 0]
        assert 1 <= real.shape[0] <= 6 and bool((rows[real.shape[0]:] == -1).all())      # padding rows of -1 (data.py:154-161)
        assert bool((real[:, 0] >= 0).all()) and bool((real[:, 0] + real[:, 2] <= 256).all())
        assert bool((real[:, 1] >= 0).all()) and bool((real[:, 1] + real[:, 3] <= 256).all())
        assert bool((real[:, 2:4] >= 8).all()) and bool((real[:, 4] < CLASSES).all())
    # an object is painted where its box says: the pixels inside differ from the background noise by the class colour
    x0, y0, w, h, c = (int(v) for v in t[0, 0])
    patch = x[0, :, y0:y0 + h, x0:x0 + w]
    colour = torch.tensor(scenes.CLASS_COLOURS[c])
    later = t[0, 1:][t[0, 1:, 4] >= 0]
    if later.numel() == 0:                                       # (nothing drawn over it)
        assert float((patch.mean((1, 2)) - colour * 0.675).abs().max()) < 0.35
    doc = scenes.coco_ground_truth(t, 100, CLASSES)
    assert len(doc['images']) == 8 and doc['images'][0]['id'] == 100
    assert len(doc['annotations']) == int((t[:, :, 4] >= 0).sum()) and doc['annotations'][0]['bbox'] == t[0, 0, :4].tolist()


def test_the_training_loop_learns_the_scenes_on_the_cpu():
    """80 SGD steps at 128 x 128 (60 with live batch norm, 20 frozen): the focal loss falls well below its initial plateau of
    ~1.1 (a random-target set does not move it)."""
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    model, history = train_detector(seed=0, iterations=80, batch=4, size=128, device='cpu', log_interval=10, lr=0.005)
    focal = [h[1] for h in history]
    assert all(math.isfinite(v) for v in focal)
    assert focal[-1] < 0.7 * focal[0], focal


# ---- GPU ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_trained_detector_ap_per_path():
    torch.backends.cudnn.benchmark = True
    iterations = int(os.environ.get('ODTK_TRAINED_AP_ITERATIONS', '700'))
    t0 = time.time()
    model, history = train_detector(seed=0, iterations=iterations)
    t_train = time.time() - t0
    assert history and all(math.isfinite(h[1] + h[2]) for h in history)
    stats = evaluate_paths(model, seed=0, images=int(os.environ.get('ODTK_TRAINED_AP_IMAGES', '96')))
    ap = {k: v[0] for k, v in stats.items() if not k.startswith('_')}
    print('trained %d iterations in %.0f s (focal %.3f -> %.3f, box %.3f -> %.3f); %d held-out objects; COCO AP per path: %s'
          % (iterations, t_train, history[0][1], history[-1][1], history[0][2], history[-1][2], stats['_objects'],
             {k: round(v, 4) for k, v in ap.items()}))
    assert ap['reference'] >= 0.25, ap                                     # a detector, not noise (long schedule: ~0.8)
    # the engines ARE the reference's detector: the BN fold and the HIP post-processing change no detection that matters
    assert abs(ap['engine_fp32'] - ap['reference']) <= 0.002, ap
    assert abs(ap['eager_fp32_hip_postproc'] - ap['reference']) <= 0.002, ap
    # fp16 = what `odtk infer` runs by default (the reference's mixed precision): within half an AP point
    assert abs(ap['engine_fp16'] - ap['reference']) <= 0.005, ap
    # bf16 = BASELINE's headline dtype: reported; bounded at one AP point (measured: profiles/r06_trained_ap.txt)
    assert abs(ap['engine_bf16'] - ap['reference']) <= 0.01, ap

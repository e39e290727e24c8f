#!/usr/bin/env python
"""COCO AP of a detector TRAINED by the product's own loop on the learnable synthetic set (odtk/scenes.py), per inference path,
against the true boxes of held-out scenes -- the long form of tests/test_gpu_trained_ap.py, on several
training seeds; plus what the post-processing sees on a trained model: candidates per image and how many of them the NMS has
to examine (the figure the fused path's lazy sort / decode would live on).

    python tools/trained_ap.py --seeds 0 1 --iterations 2500 --images 256 > profiles/r06_trained_ap.txt
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import torch  # noqa: E402
import test_gpu_trained_ap as T  # noqa: E402
from odtk import _C, scenes  # noqa: E402

ap_ = argparse.ArgumentParser()
ap_.add_argument('--seeds', type=int, nargs='+', default=[0, 1])
ap_.add_argument('--iterations', type=int, default=2500)
ap_.add_argument('--batch', type=int, default=16)
ap_.add_argument('--size', type=int, default=512)
ap_.add_argument('--images', type=int, default=256)
ap_.add_argument('--backbone', default='ResNet18FPN')
ap_.add_argument('--json', default=None)
ap_.add_argument('--save-postproc-inputs', default=None, help='npz: what the post-processing sees on the first held-out batch of the '
                 'first seed -- decode_levels output (the NMS input) of the bf16 engine, for offline NMS studies and fixtures')
ap_.add_argument('--progress', action='store_true', help='debug: synchronise behind every inference path of every held-out batch and '
                 'say so on stderr (a device fault then names its path); the JSON is rewritten after every seed')
args = ap_.parse_args()
torch.backends.cudnn.benchmark = True

PATHS = ['reference', 'engine_fp32', 'eager_fp32_hip_postproc', 'engine_fp16', 'engine_bf16', 'eager_autocast_fp16', 'eager_autocast_bf16']
table, extra = {}, {}
for seed in args.seeds:
    t0 = time.time()
    model, history = T.train_detector(seed=seed, iterations=args.iterations, batch=args.batch, size=args.size, backbone=args.backbone,
                                      log_interval=max(50, args.iterations // 20))
    t_train = time.time() - t0
    say = (lambda msg, seed=seed: print('[seed %d] %s' % (seed, msg), file=sys.stderr, flush=True)) if args.progress else None
    if say:
        say('trained in %.0f s' % t_train)
    stats = T.evaluate_paths(model, seed=seed, images=args.images, batch=args.batch, size=args.size, progress=say)
    table[seed] = stats
    # what the NMS sees on a trained model (bf16 engine, the timed path): candidates with a positive score per image (K) and how
    # many of them it examines before 100 are kept or the list is exhausted
    held = scenes.SceneBatches(args.batch, args.size, args.size, classes=T.CLASSES, seed=seed, device='cuda', start=T.HELD_OUT_START)
    x = held.batch_at(0)[0].contiguous(memory_format=torch.channels_last)
    trace = torch.zeros(_C.TRACE_WORDS, dtype=torch.int64, device='cuda')
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        for _ in range(3):
            model(x)
        _C.debug_set_trace(trace)
        det = model(x)
        torch.cuda.synchronize()
        _C.debug_set_trace(None)
    rows = trace.cpu().view(-1, 8)[64 + args.batch:64 + 2 * args.batch]
    extra[seed] = {'train_s': round(t_train, 1), 'history': [(h[0], round(h[1], 4), round(h[2], 4)) for h in history],
                   'nms_examined': [int(r[5]) for r in rows], 'nms_candidates': [int(r[6]) for r in rows],
                   'nms_us': [round((int(r[4]) - int(r[0])) / 100.0, 1) for r in rows],
                   'kept': [int(v) for v in (det[0] > 0).sum(1).tolist()]}
    if args.save_postproc_inputs and seed == args.seeds[0]:
        import numpy as np
        from odtk import box as box_ops
        eng = model.inference_engine(torch.bfloat16)
        with torch.no_grad():
            cls_h, box_h = eng.heads(x)
        strides = [x.shape[-1] // c.shape[-1] for c in cls_h]
        s_, b_, c_ = box_ops.decode_levels(cls_h, box_h, strides, model.threshold, model.top_n, model.anchors, logits=True)
        np.savez_compressed(args.save_postproc_inputs, scores=s_.float().cpu().numpy(), boxes=b_.float().cpu().numpy(),
                            classes=c_.float().cpu().numpy(), det_scores=det[0].float().cpu().numpy(),
                            det_boxes=det[1].float().cpu().numpy(), det_classes=det[2].float().cpu().numpy(),
                            top_n=model.top_n, nms=model.nms, detections=model.detections, levels=len(strides))
    del model
    torch.cuda.empty_cache()
    if args.json:                                            # (after every seed: a later fault does not take the finished seeds with it)
        with open(args.json, 'w') as f:
            json.dump({'args': vars(args), 'table': {str(k): v for k, v in table.items()}, 'extra': {str(k): v for k, v in extra.items()}}, f, indent=1)

print('COCO AP (IoU 0.50:0.95 | 0.50 | 0.75, odtk/cocoeval.py) against the TRUE boxes of %d held-out scenes per seed; %s trained %d '
      'iterations x batch %d at %dx%d by odtk/train.py (fp32, HIP target assignment + fused loss; 75 %% live batch norm from the random '
      'init, 25 %% frozen at lr / 10), %d classes; one row per inference path' % (args.images, args.backbone, args.iterations, args.batch,
                                                                                 args.size, args.size, T.CLASSES))
print('%-26s' % 'path' + ''.join('%28s' % ('training seed %d' % s) for s in args.seeds) + '%14s' % 'max |dAP|')
for name in PATHS:
    cells, devs = [], []
    for s in args.seeds:
        v = table[s].get(name)
        if v is None:
            cells.append('%28s' % '-')
            continue
        cells.append('%28s' % ('%.4f | %.4f | %.4f' % (v[0], v[1], v[2])))
        devs.append(abs(v[0] - table[s]['reference'][0]))
    print('%-26s' % name + ''.join(cells) + '%14s' % ('%.4f' % max(devs) if devs else '-'))
print()
for s in args.seeds:
    e = extra[s]
    print('seed %d: trained in %.0f s; held-out objects %d; detections per path %s' % (s, e['train_s'], table[s]['_objects'], table[s]['_detections']))
    print('   loss (iteration, focal, box): ' + '  '.join('%d: %.3f / %.3f' % h for h in e['history'][::max(1, len(e['history']) // 8)] + e['history'][-1:]))
    print('   NMS on the first held-out batch (bf16 engine, in Model.forward): candidates per image %s' % e['nms_candidates'])
    print('                                                      examined by the NMS   %s' % e['nms_examined'])
    print('                                                      detections kept       %s' % e['kept'])
    print('                                                      us per image          %s' % e['nms_us'])
if args.json:
    with open(args.json, 'w') as f:
        json.dump({'args': vars(args), 'table': {str(k): v for k, v in table.items()}, 'extra': {str(k): v for k, v in extra.items()}}, f, indent=1)

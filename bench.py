#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: images/sec end-to-end (backbone + heads + sigmoid +
decode x5 + NMS), ResNet50FPN, 800x1280, batch 8 per GPU, bf16 autocast, channels_last.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --mode train ...            # BASELINE config 3: RN50FPN fp32 training, 2 images per GPU

One process per GPU.  Inference is embarrassingly parallel over images, so N>1 runs N
replicas with NO data-path collective (weak scaling); the only collectives are the barrier and the
MAX-reduction of the timed region.  A "step" is one call of the drop-in `Model.forward` (eval) on one
device-resident synthetic batch -- since round 2 that call IS the BN-folded engine + the HIP
post-processing (odtk/model.py); `--no-fuse` times the eager nn.Module graph instead, and the default run
reports that number too (`eager`).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- the dominant kernel of the hand-written path (prefilter_scan_kernel, HBM-bound):
                  algorithmic bytes per launch (sizeof(dtype) x every score of the batch)
                  divided by its average launch duration measured with hipEvents on the launch stream
                  (include/odtk_hip.h odtk_profile_*), inside the timed region.
  latency_bound-- the two latency-bound post-processing launches (select_decode, nms) against their
                  serial-chain lower bounds.
  epilogue_roofline -- the engine's own HBM-bound epilogue kernels (bias_act, the stem's pool pass, the FPN upsampling):
                  algorithmic bytes per step / their time per step, dispatch timestamps, three untimed steps.
  conv_roofline-- whole-pipeline view: conv FLOP/s achieved vs the dense bf16 MFMA peak.
  cpu_baseline -- the reference's pure-PyTorch CPU path on the host cores (rank 0, N=1 only, bounded
                  sample): `postproc` = decode x5 + nms of the pinned oracle restatement of odtk/box.py on
                  the head tensors captured from the timed path (the hot path itself), `value` = the whole
                  pipeline (same model on the CPU + that post-processing).  oracle/ is imported ONLY here.
"""
import argparse
import copy
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'retinanet-examples_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

# multi-process GPU work on this image needs dmabuf IPC (the host driver has no legacy IPC: RCCL otherwise fails with
# `hipIpcGetMemHandle: invalid argument`); the launcher's environment normally carries it already
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np                # noqa: E402
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak
CLOCK_GHZ = 2.4
# SURVEY.md 8(d) "sparse-realistic": logits ~ N(-ln 99, 0.573^2) -> this fraction of the scores is >= 0.05
# (23 224 / 5 646 / 1 456 / 391 / 93 candidates per image on P3..P7 at 800x1280)
SPEC_FRACTION = 23224 / 11520000.0
SPEC_CANDIDATES = [23224, 5646, 1456, 391, 93]
# Serial-chain lower bounds: greedy NMS resolves its `detections` kept boxes one after the
# other -- per kept box at least one LDS read (64 clk issue->use), the dependent IoU arithmetic (~25 VALU ops x 4 clk)
# and a ballot + readlane resolve (~30 clk) ~ 190 clk; a top-1000 selection needs at least one pass over the
# candidate keys out of L2 (~1 us at these sizes) + one 1024-key sort (55 dependent compare-exchange stages ~ 3 us)
NMS_CLK_PER_KEPT = 190


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def conv_flops_per_image(model, x):
    """2 x MACs of every Conv2d in one forward (bias / BN / ReLU / upsample ignored)."""
    total = [0]
    hooks = []

    def hook(m, inp, out):
        k = m.kernel_size[0] * m.kernel_size[1] * (m.in_channels // m.groups)
        total[0] += 2 * k * out.numel() // out.shape[0]

    for m in model.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        model.heads(x)
    for h in hooks:
        h.remove()
    return total[0]


def calibrate_cls_head(model, heads, x, fraction, threshold):
    """Random-init heads score ~0.01 everywhere (class prior) = zero detections, so decode/NMS would have
    nothing to do.  Rescale the LAST classification conv so that the specified FRACTION of the scores of the
    path being timed clears the threshold (0.2016 %).  Matching sigma alone is not enough: the
    logits of a random network are not Gaussian and their tail is lighter (round 1 matched sigma = 0.573 and
    got 0.046 %).  The scale is exact: the (1 - fraction) quantile q of the centred logits, found by bisection
    on a count, must land on logit(threshold) - bias."""
    import math
    with torch.no_grad():
        cls_heads, _ = heads(x)
        bias = model.cls_head[-1].bias
        prior = float(bias.float().mean())
        centred = torch.cat([(c.float() - bias.view(1, -1, 1, 1)).flatten() for c in cls_heads])
        want = int(round(fraction * centred.numel()))
        lo, hi = 0.0, float(centred.max())
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            if int((centred >= mid).sum()) > want:
                lo = mid
            else:
                hi = mid
        q = 0.5 * (lo + hi)
        target = math.log(threshold / (1.0 - threshold)) - prior
        sigma_before = centred.std().item()
        scale = target / max(q, 1e-12)
        model.cls_head[-1].weight.mul_(scale)               # bumps the version counter: the engine is re-folded
    return sigma_before, sigma_before * scale


def run_train(args, rank, local_rank, world, dev):
    """BASELINE config 3: ResNet50FPN fp32 training (FocalLoss + SmoothL1), 2 images per GPU, DDP over RCCL.
    A step = forward + loss + backward (+ bucketed gradient all-reduce overlapped with it) + SGD update.
    `exposed_allreduce_ms` = step time with DDP's all-reduce minus the time of the same step on the bare module (same
    compute, no communication): the part of the all-reduce that backward does not hide."""
    from odtk import parallel, train as T
    from odtk.model import Model
    torch.manual_seed(0)
    model = Model(backbones=args.backbone, classes=80, rotated_bbox=args.rotated_bbox)
    model.initialize(None)
    if args.rotated_bbox and args.unit_rotation:
        # the reference's init puts the -4.6 class prior on all six box outputs (model.py:121-122): the box loss starts at ~28 and
        # SGD diverges within a few dozen steps in the reference's own arithmetic (tools/rotated_train_probe.py --bench-like,
        # profiles/r05_rotated_train_trajectory.txt: non-finite at step 6 fused / 7 torch); (0, 0, 0, 0, sin 0, cos 1) is what a trained rotated model emits
        with torch.no_grad():
            bias = model.box_head[-1].bias.view(model.num_anchors, 6)
            bias.zero_()
            bias[:, 5] = 1.0
    per_gpu = args.batch
    amp_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': None}[args.dtype]
    on_gpu = dev.type == 'cuda'
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    model, net, optimizer, scheduler = T.prepare(model, dev, lr=0.01, world=world, rank=rank, warmup=1000)
    model.fused_loss = not args.no_fused_loss
    scaler = torch.amp.GradScaler('cuda', enabled=amp_dtype == torch.float16) if amp_dtype == torch.float16 else None
    source = T.SyntheticBatches(per_gpu * world, args.height, args.width, classes=80, max_boxes=20, seed=0, rank=rank,
                                world=world, device='cpu', rotated=args.rotated_bbox)
    batches = []
    for _ in range(4):                                       # a small device-resident pool (no PCIe in the timed region)
        d, t = source.batch()
        batches.append((d.to(dev).contiguous(memory_format=torch.channels_last), t.to(dev)))
    it = [0]
    losses = []

    def step():
        d, t = batches[it[0] % len(batches)]
        it[0] += 1
        c, b = T.train_step(net, optimizer, scheduler, scaler, d, t, amp_dtype)
        losses.append((c, b))
        return c

    for _ in range(args.warmup):
        step()
    sync()
    elapsed, _ = parallel.timed_steps(step, args.steps, sync, dev)
    both = T.reduce_losses(losses[-1][0], losses[-1][1], world)
    # the hand-written training-side kernels, timed on a few extra steps outside the timed region.  EVERY rank runs the
    # steps (each one all-reduces gradients under DDP); only rank 0 records
    hip_kernels = None
    if on_gpu:
        from odtk import _C
    if rank == 0 and on_gpu:
        _C.profile_enable(True, ('retina_loss_kernel', 'loss_reduce_kernel', 'snap_to_anchors_kernel'))
        _C.profile_collect()
    for _ in range(5 if on_gpu else 0):
        step()
    sync()
    if rank == 0 and on_gpu:
        _C.profile_enable(False)
        prof = _C.profile_collect()
        hip_kernels = {k: {'us_per_step': round(v[0] / 5 * 1e3, 1), 'launches_per_step': v[1] // 5} for k, v in prof.items() if v[1]}
        if 'retina_loss_kernel' in hip_kernels:
            # forward reads every logit once, backward reads it again and writes its gradient: 3 x sizeof(dtype) per logit
            from odtk import synthetic
            esize = 4 if amp_dtype is None else 2
            logits = per_gpu * sum(model.num_anchors * model.classes * h * w for h, w in synthetic.level_shapes(args.height, args.width))
            alg = 3 * esize * logits
            # (the forward goes through a workspace since round 6: its second, tiny launch -- loss_reduce_kernel -- is part of
            # the time the bytes are divided by)
            t = (hip_kernels['retina_loss_kernel']['us_per_step'] + hip_kernels.get('loss_reduce_kernel', {}).get('us_per_step', 0.0)) * 1e-6
            hip_kernels['retina_loss_kernel'].update({'alg_bytes_per_step': alg, 'achieved_GBps': round(alg / t / 1e9, 1),
                                                      'frac_of_hbm_peak': round(alg / t / 1e9 / HBM_PEAK_GBS, 4)})
    exposed = None
    if world > 1:
        # the same step on the bare module: identical compute, no gradient all-reduce (replicas drift apart from here on,
        # which no longer matters: nothing is timed with DDP after this)
        def quiet_step():
            d, t = batches[it[0] % len(batches)]
            it[0] += 1
            return T.train_step(model, optimizer, scheduler, scaler, d, t, amp_dtype)[0]
        for _ in range(2 if on_gpu else 1):
            quiet_step()
        n_quiet = max(args.steps // 2, 5) if on_gpu else max(args.steps // 2, 1)
        quiet, _ = parallel.timed_steps(quiet_step, n_quiet, sync, dev)
        exposed = round((elapsed / args.steps - quiet / n_quiet) * 1e3, 3)
    line = None
    if rank == 0:
        images = per_gpu * world * args.steps
        grad_bytes = sum(p.numel() * 4 for p in model.parameters() if p.requires_grad)
        line = {
            'metric': 'images/sec training (fwd + FocalLoss/SmoothL1 + bwd + SGD), %s%s %dpx, %d img/GPU' % (
                short_name(args.backbone), ' --rotated-bbox' if args.rotated_bbox else '', args.height, per_gpu),
            'value': round(images / elapsed, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic randn images + synthetic targets (1..20 boxes per image, SURVEY 8d config 3), random-init weights',
            'config': {'workload': '%s %s training, %d images per GPU at %dx%d, target assignment + losses: %s'
                                   % (args.backbone, args.dtype, per_gpu, args.height, args.width,
                                      'fused HIP' if (model.fused_loss and on_gpu) else 'torch'),
                       'global_batch': per_gpu * world,
                       'parallelism': 'ddp x%d (%s all-reduce, 25 MB buckets, overlapped)' % (world, 'RCCL' if args.backend == 'nccl' else args.backend),
                       'gradient_bytes_per_step': grad_bytes, 'device': dev.type, 'backend': args.backend if world > 1 else None},
            'exposed_allreduce_ms': exposed, 'hip_kernels': hip_kernels,
            # a throughput measured on non-finite tensors is not a measurement of the configuration: `finite` says which it is
            'loss': {'focal': round(float(both[0]), 5), 'box': round(float(both[1]), 5), 'first_step': [round(float(v), 5) for v in losses[0]],
                     'finite': bool(math.isfinite(float(both[0])) and math.isfinite(float(both[1]))), 'sgd_steps': len(losses)},
        }
        if not line['loss']['finite']:
            line['error'] = 'loss went non-finite within %d SGD steps: the throughput below is not a valid measurement' % len(losses)
    return line


def kernel_src_hash():
    """sha256 (16 hex) over the HIP sources of the C-ABI library: ties a figure quoted from profiles/ to the kernels it was
    measured on (tools/pmc_traffic.py records the same hash)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'retinanet-examples_amd', 'csrc', '*.hpp')) +
                    glob.glob(os.path.join(ROOT, 'retinanet-examples_amd', 'csrc', '*.hip'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def default_workload(args):
    """The configuration the committed profiles were taken on (BASELINE configs[1])."""
    return (args.backbone == 'ResNet50FPN' and args.batch == 8 and (args.height, args.width) == (800, 1280) and args.dtype == 'bf16'
            and not args.rotated_bbox and not args.no_fuse and args.postproc == 'fused')


def short_name(backbone):
    return backbone.replace('ResNet', 'RN').replace('ResNeXt', 'RNX')


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'])
    ap.add_argument('--backbone', default='ResNet50FPN')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (default: 8 infer, 2 train)')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--dtype', default=None, choices=['bf16', 'fp16', 'fp32'], help='default: bf16 infer, fp32 train')
    ap.add_argument('--fraction', type=float, default=SPEC_FRACTION,
                    help='fraction of the scores calibrated to clear the threshold (SURVEY 8d sparse-realistic)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the cpu_baseline leg (0 = skip)')
    ap.add_argument('--no-miopen-find', action='store_true',
                    help='torch.backends.cudnn.benchmark = False (MIOpen immediate mode instead of find mode)')
    ap.add_argument('--rotated-bbox', action='store_true', help='BASELINE config 5: 27 anchors, 6 box parameters, '
                                                                'rotated decode + polygon-IoU NMS')
    ap.add_argument('--unit-rotation', action='store_true',
                    help="with --rotated-bbox: bias the box head's (sin, cos) outputs to (0, 1) and the deltas to 0, as a trained rotated "
                         "model emits them; the reference's random init puts the -4.6 class prior on all six outputs "
                         "(model.py:121-122), i.e. 6.5x inflated, mutually overlapping quads")
    ap.add_argument('--no-fuse', action='store_true',
                    help='time the eager nn.Module graph under autocast (Model.fused_graph = False) instead of the '
                         'BN-folded engine Model.forward uses by default')
    ap.add_argument('--no-conv-library', action='store_true',
                    help='A/B: every k x k convolution as MIOpen convolution + odtk_bias_act (round 4\'s graph) instead of the '
                         'per-layer plan that may route it to the convolution library\'s fused epilogue (csrc/conv_ck.cpp)')
    ap.add_argument('--no-eager-leg', action='store_true', help='skip the extra (untimed-region) eager-graph measurement')
    ap.add_argument('--no-level-streams', action='store_true',
                    help='run the head towers of all pyramid levels on one stream (default: small levels on side streams)')
    ap.add_argument('--tower-plan', type=int, default=0, help='assignment of pyramid levels to HIP streams (odtk/fused.py)')
    ap.add_argument('--postproc', default='fused', choices=['fused', 'reference'],
                    help="fused: sigmoid+decode+nms read the bf16 channels_last head tensors in place (3 launches); "
                         "reference: the reference's op sequence (sigmoid, .contiguous(), .float(), decode x5, cat, nms)")
    ap.add_argument('--no-fused-loss', action='store_true', help='train mode: torch losses instead of the HIP focal/smooth-L1 kernel')
    ap.add_argument('--no-other-configs', action='store_true',
                    help="default 1-GPU run: skip the legs for BASELINE.json's other configurations (`other_configs` on the line)")
    ap.add_argument('--other-steps', type=int, default=10, help='timed steps of each other_configs leg')
    ap.add_argument('--leg-budget-s', type=float, default=330.0,
                    help='wall-clock budget of the other_configs legs together: a leg only starts while budget is left (its typical '
                         'cost included); the rest are reported as skipped.  0 = no limit (tools/profile_round.sh)')
    ap.add_argument('--device', default='cuda', choices=['cuda', 'cpu'],
                    help='cpu: the plumbing run of the CPU tests (gloo, the reference CPU branch of odtk/box.py; no HIP kernel runs, '
                         'no roofline) -- never a measurement')
    ap.add_argument('--backend', default=None, choices=['nccl', 'gloo'], help='default: nccl (= RCCL) on cuda, gloo on cpu')
    ap.add_argument('--tiny', action='store_true', help='ResNet18FPN on one 128x128 image per rank: the size the CPU tests run')
    ap.add_argument('--detail-out', default=os.path.join(ROOT, 'gpurun_out', 'bench_detail_latest.json'),
                    help='where the full record (every leg, every sub-object) is written; the stdout line is the compact headline')
    return ap


def sanitize(obj):
    """Strict-JSON form: non-finite floats -> None, tensors / numpy scalars -> python numbers, tuples -> lists."""
    if isinstance(obj, dict):
        return {str(k): sanitize(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [sanitize(v) for v in obj]
    if isinstance(obj, bool) or obj is None or isinstance(obj, (int, str)):
        return obj
    if isinstance(obj, float):
        return obj if math.isfinite(obj) else None
    if hasattr(obj, 'item'):                                 # torch / numpy scalar
        try:
            return sanitize(obj.item())
        except Exception:                                    # noqa: BLE001
            return str(obj)
    return str(obj)


def strict_json(obj):
    return json.dumps(sanitize(obj), allow_nan=False, separators=(', ', ': '))


def pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if d else None


HEADLINE_LIMIT = 4096


def headline(full):
    """The compact object that goes on stdout (< HEADLINE_LIMIT bytes, strict JSON): the contract's keys + `roofline` +
    `cpu_baseline`, the latency-bound ratios, the conv roofline, and ONE number per other configuration.  Everything else
    (per-leg records, epilogue figures, figures quoted from committed profiles) lives in the detail file."""
    h = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                  'scaling', 'vs_baseline', 'dtype', 'data')}
    cfg = full.get('config') or {}
    h['config'] = pick(cfg, 'workload', 'global_batch', 'parallelism', 'entry', 'postproc', 'device', 'backend')
    if full.get('roofline'):
        h['roofline'] = pick(full['roofline'], 'kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic',
                             'alg_bytes_per_launch', 'avg_us', 'launches')
    else:
        h['roofline'] = None
    lb = full.get('latency_bound') or {}
    if lb:
        h['latency_bound'] = {k: pick(v, 'us_per_step', 'launches_per_step', 'lower_bound_us', 'ratio') for k, v in lb.items()}
    if full.get('conv_roofline'):
        h['conv_roofline'] = pick(full['conv_roofline'], 'bound', 'achieved', 'peak', 'unit', 'frac')
    if full.get('kernels'):
        h['kernels_avg_us'] = {k: v.get('avg_us') for k, v in full['kernels'].items()}
        h['postproc_us_per_step'] = full.get('postproc_us_per_step')
    if full.get('eager'):
        h['eager'] = pick(full['eager'], 'value', 'ms_per_step')
    if full.get('conv_epilogue'):
        h['conv_epilogue'] = pick(full['conv_epilogue'], 'layers_routed_to_library', 'layers_measured', 'us_saved_per_step', 'plan_hash')
    cb = full.get('cpu_baseline')
    if cb:
        h['cpu_baseline'] = pick(cb, 'value', 'unit', 'cores', 'kind', 'sample')
        if cb.get('postproc'):
            h['cpu_baseline']['postproc'] = pick(cb['postproc'], 'value', 'unit', 'ms_per_image', 'cores', 'kind',
                                                 'gpu_us_per_image', 'gpu_vs_cpu')
    else:
        h['cpu_baseline'] = None
    for k in ('parity', 'exposed_allreduce_ms', 'loss', 'quoted'):
        if full.get(k) is not None:
            h[k] = full[k]
    if full.get('hip_kernels'):
        h['hip_kernels'] = {k: pick(v, 'us_per_step', 'frac_of_hbm_peak') for k, v in full['hip_kernels'].items()}
    legs = full.get('other_configs')
    if legs:
        h['other_configs'] = {}
        for leg in legs:
            e = pick(leg, 'value', 'unit', 'ms_per_step', 'dtype', 'error', 'skipped')
            if leg.get('roofline'):
                e['roofline_frac'] = leg['roofline'].get('frac')
            if isinstance(leg.get('latency_bound'), dict):
                e['latency_ratio'] = {k: v.get('ratio') for k, v in leg['latency_bound'].items()}
            if isinstance(leg.get('loss'), dict):
                e['loss'] = pick(leg['loss'], 'focal', 'box', 'finite')
            if 'exposed_allreduce_ms' in leg:                # training legs: null at N = 1 (no collective to expose)
                e['exposed_allreduce_ms'] = leg['exposed_allreduce_ms']
            if isinstance(leg.get('hip_kernels'), dict):     # ... and the HIP loss / target kernels against the HBM peak
                e['hip_kernels_frac'] = {k: v.get('frac_of_hbm_peak') for k, v in leg['hip_kernels'].items()
                                         if isinstance(v, dict) and v.get('frac_of_hbm_peak') is not None}
            h['other_configs'][leg.get('key', leg.get('leg', '?'))] = e
    h['detail'] = full.get('detail')
    return h


def headline_line(full):
    """-> the ONE stdout line.  Never raises, never exceeds HEADLINE_LIMIT: optional objects are dropped in order (least
    important first) until it fits."""
    h = headline(full)
    line = strict_json(h)
    for drop in ('quoted', 'eager', 'kernels_avg_us', 'hip_kernels', 'loss', 'parity', 'other_configs', 'conv_roofline',
                 'latency_bound', 'data'):
        if len(line) < HEADLINE_LIMIT:
            break
        h.pop(drop, None)
        h['dropped_to_fit'] = h.get('dropped_to_fit', []) + [drop]
        line = strict_json(h)
    if len(line) >= HEADLINE_LIMIT and isinstance(h.get('cpu_baseline'), dict):
        h['cpu_baseline']['sample'] = str(h['cpu_baseline'].get('sample'))[:120]
        line = strict_json(h)
    return line


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, the reference's own habit --
    odtk/main.py:155-171,246-250 spawns one worker per GPU and initialises NCCL from MASTER_ADDR/PORT), wait for them, and
    hand back the first non-zero exit code.  Rank 0 inherits stdout, so its one JSON line is this command's line."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   ODTK_BENCH_SELF_LAUNCHED='1')
        if args.device == 'cpu':
            env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 2) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait())
    except BaseException:                                    # Ctrl-C / a driver timeout: take the ranks we started down with us
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise
    return next((c for c in codes if c), 0)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = build_parser()
    args = ap.parse_args(argv)
    default_run = all(getattr(args, k) == ap.get_default(k) for k in
                      ('mode', 'backbone', 'batch', 'height', 'width', 'dtype', 'rotated_bbox', 'no_fuse', 'postproc', 'fraction',
                       'device', 'tiny'))
    if args.tiny:
        args.backbone, args.height, args.width = 'ResNet18FPN', 128, 128
        args.batch = args.batch or 1
        args.cpu_seconds, args.no_eager_leg, args.no_other_configs = 0.0, True, True
    if args.batch is None:
        args.batch = 8 if args.mode == 'infer' else 2
    if args.dtype is None:
        args.dtype = 'fp32' if (args.mode == 'train' or args.device == 'cpu') else 'bf16'
    if args.backend is None:
        args.backend = 'nccl' if args.device == 'cuda' else 'gloo'    # "nccl" IS RCCL on ROCm

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, argv))

    # stdout carries ONE line.  Libraries write there too (gloo's "[Gloo] Rank 0 is connected ..." banner, NCCL_DEBUG=INFO):
    # point fd 1 at stderr for the life of the process and keep a private handle on the real stdout for the headline
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    from odtk import parallel
    rank, local_rank, world = parallel.init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d was started with WORLD_SIZE=%d: launch one rank per GPU (or drop the launcher: '
                         'bench.py spawns its own ranks)' % (args.gpus, world))
    if args.device == 'cuda':
        assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP post-processing is what it measures)'
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    else:
        dev = torch.device('cpu')
    torch.backends.cudnn.benchmark = not args.no_miopen_find
    if args.no_conv_library:
        from odtk import fused
        fused._Conv.use_conv_library = False
    t_start = time.perf_counter()
    if args.mode == 'train':
        line = run_train(args, rank, local_rank, world, dev)
    else:
        line = run_infer(args, rank, world, dev)
    if rank == 0:
        # the headline is safe from here on: printed FIRST to stderr (a leg that kills the process cannot take it down),
        # then the legs, then the one stdout line
        log('[bench] headline %.1f s: %s' % (time.perf_counter() - t_start, headline_line(line)))
        # BASELINE.json's other configurations behind the headline (1 GPU, default invocation only): RN101FPN bs 16 (config 4),
        # --rotated-bbox bs 8 (config 5), the per-GPU share of config 3 (fp32 training, 2 images) and the batch-1 latency the
        # reference publishes for its TensorRT engines (README.md:26-34; timing loop extras/cppapi/infer.cpp:69-77)
        if world == 1 and default_run and args.mode == 'infer' and not args.no_other_configs:
            try:
                line['other_configs'] = other_configs(args, rank, local_rank, world, dev)
            except Exception as e:                           # noqa: BLE001 -- never the headline's problem
                line['other_configs'] = [{'key': 'legs', 'error': '%s: %s' % (type(e).__name__, e)}]
        line['wall_s'] = round(time.perf_counter() - t_start, 1)
        try:
            os.makedirs(os.path.dirname(args.detail_out), exist_ok=True)
            with open(args.detail_out, 'w') as f:
                f.write(json.dumps(sanitize(line), allow_nan=False, indent=1))
            line['detail'] = os.path.relpath(args.detail_out, ROOT)
        except OSError as e:
            line['detail'] = 'not written: %s' % e
        real_stdout.write(headline_line(line) + '\n')
        real_stdout.flush()
    if world > 1:
        dist.destroy_process_group()


def leg_args(args, **over):
    leg = copy.copy(args)
    leg.steps, leg.warmup, leg.cpu_seconds, leg.no_eager_leg = args.other_steps, min(args.warmup, 5), 0.0, True
    for k, v in over.items():
        setattr(leg, k, v)
    return leg


# (key, name, typical wall seconds on a fresh box [MIOpen find + model build dominate; the leg_wall_s of the last three detail
# files of round 5 and round 4's full run, profiles/bench_r05_detail_final{,2,3}.json, bench_r04_default_with_legs.json], builder of
# the leg).  Order = priority (BASELINE.json's configurations first: 4, 5, 3; then the fp16 twin of config 2, the batch-1 latency and
# the variants); the legs run while --leg-budget-s lasts -- the default budget takes all of them (sum of the typical costs 302 s).
def leg_table(args, rank, local_rank, world, dev):
    return [
        ('cfg4_rn101_bs16', 'config 4: ResNet101FPN bf16 inference bs 16', 43,
         lambda: run_infer(leg_args(args, backbone='ResNet101FPN', batch=16), rank, world, dev)),
        ('cfg5_rotated_bs8', 'config 5: ResNet50FPN --rotated-bbox bf16 inference bs 8', 45,
         lambda: run_infer(leg_args(args, rotated_bbox=True), rank, world, dev)),
        ('cfg3_train_fp32_2img', 'config 3 (per-GPU share): ResNet50FPN fp32 training, 2 images per GPU', 127,
         lambda: run_train(leg_args(args, mode='train', batch=2, dtype='fp32'), rank, local_rank, world, dev)),
        # the precision `odtk infer` runs by default (fp16 autocast, as the reference's mixed precision)
        ('cfg2_fp16', 'config 2 in fp16: ResNet50FPN fp16 inference bs 8', 22,
         lambda: run_infer(leg_args(args, dtype='fp16'), rank, world, dev)),
        ('bs1_latency_ms', 'batch-1 latency: ResNet50FPN bf16', 13, lambda: run_latency(leg_args(args, batch=1), dev)),
        # (unit (sin, cos) head bias: with the reference's own initialisation -- the -4.6 class prior on all six box outputs,
        # model.py:121-122 -- SGD diverges within 6-7 steps in the reference's arithmetic exactly as in the fused kernels:
        # profiles/r05_rotated_train_trajectory.txt, tools/rotated_train_probe.py --bench-like)
        ('cfg3_train_rotated', 'config 3 with --rotated-bbox, unit (sin, cos) head bias (per-GPU share): fused rotated target assignment', 47,
         lambda: run_train(leg_args(args, mode='train', batch=2, dtype='fp32', rotated_bbox=True, unit_rotation=True), rank, local_rank, world, dev)),
        ('cfg5_rotated_unit', 'config 5 with a unit (sin, cos) head bias', 5,
         lambda: run_infer(leg_args(args, rotated_bbox=True, unit_rotation=True), rank, world, dev)),
    ]


def other_configs(args, rank, local_rank, world, dev):
    legs = []
    t_legs = time.perf_counter()
    for key, name, typical_s, fn in leg_table(args, rank, local_rank, world, dev):
        left = args.leg_budget_s - (time.perf_counter() - t_legs)
        if args.leg_budget_s > 0 and left < typical_s:
            legs.append({'key': key, 'leg': name, 'skipped': 'leg budget (%.0f s left of --leg-budget-s %.0f, typical cost %d s)'
                                                            % (max(left, 0), args.leg_budget_s, typical_s)})
            log('[bench] leg %s: skipped (budget)' % key)
            continue
        t0 = time.perf_counter()
        try:
            line = fn()
        except Exception as e:                                # noqa: BLE001 -- a failing leg is reported, never hidden, and never takes the headline down
            line = {'error': '%s: %s' % (type(e).__name__, e)}
        line['key'], line['leg'] = key, name
        line['leg_wall_s'] = round(time.perf_counter() - t0, 1)
        legs.append(line)
        if dev.type == 'cuda':
            torch.cuda.empty_cache()
        log('[bench] leg ' + strict_json(pick(line, 'key', 'value', 'unit', 'ms_per_step', 'error', 'leg_wall_s')))
    return legs


def run_latency(args, dev):
    """Batch-1 latency of Model.forward, the figure the reference publishes for its TensorRT engines (README.md:26-34: RN50FPN
    11 ms on A100, 18 ms on V100, fp16, bs 1, post-processing included).  Timed like its loop (extras/cppapi/infer.cpp:69-77):
    one synchronous call after the other, wall time / count.  `graph`: the same call replayed as ONE hipGraph
    (Model.forward(x, graph=True))."""
    from odtk.model import Model
    amp_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': None}[args.dtype]
    torch.manual_seed(0)
    model = Model(backbones=args.backbone, classes=80, rotated_bbox=args.rotated_bbox)
    model.initialize(None)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn(1, 3, args.height, args.width, generator=torch.Generator().manual_seed(0)).to(dev).contiguous(memory_format=torch.channels_last)
    calibrate_cls_head(model, lambda t: model.inference_engine(amp_dtype or torch.float32).heads(t), x, args.fraction, model.threshold)

    def step(graph):
        with torch.no_grad(), torch.autocast('cuda', dtype=amp_dtype, enabled=amp_dtype is not None):
            return model(x, graph=graph)

    out = {}
    count = 100
    for name, graph in (('eager', False), ('graph', True)):
        try:
            for _ in range(10):
                det = step(graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(count):
                step(graph)
                torch.cuda.synchronize()                     # one inference at a time, like engine->infer()
            sync_ms = (time.perf_counter() - t0) / count * 1e3
            t0 = time.perf_counter()
            for _ in range(count):
                step(graph)
            torch.cuda.synchronize()
            out[name] = {'latency_ms': round(sync_ms, 3), 'back_to_back_ms': round((time.perf_counter() - t0) / count * 1e3, 3),
                         'detections': int((det[0] > 0).sum())}
        except Exception as e:
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
    best = min((v['latency_ms'] for v in out.values() if 'latency_ms' in v), default=None)
    return {'metric': 'batch-1 latency end-to-end (incl. decode+NMS), %s %dpx' % (short_name(args.backbone), args.height),
            'latency_bs1_ms': best, 'value': best, 'unit': 'ms', 'higher_is_better': False, 'dtype': args.dtype, 'iterations': count,
            'eager': out.get('eager'), 'graph': out.get('graph'),
            'reference_published_ms': {'A100 TensorRT fp16': 11, 'V100 TensorRT fp16': 18, 'source': 'reference README.md:33 (other hardware, not a baseline)'},
            'config': {'workload': '%s %s inference, bs=1 at %dx%d, Model.forward (eval), one synchronous call at a time'
                                   % (args.backbone, args.dtype, args.height, args.width)}}


def run_infer_cpu_plumbing(args, rank, world, dev, model, amp_dtype):
    """--device cpu: the same launch / sharding / timing bracket / line as the GPU run, with Model.forward on the reference's
    CPU branch of odtk/box.py (BASELINE configs[0]'s plumbing).  No HIP kernel runs, so there is no roofline: the CPU tests
    use this to run `bench.py --gpus 2` end to end (gloo)."""
    from odtk import parallel
    x = torch.randn(args.batch, 3, args.height, args.width, generator=torch.Generator().manual_seed(rank)).contiguous(
        memory_format=torch.channels_last)

    def step():
        with torch.no_grad():
            return model(x)

    out = None
    for _ in range(max(args.warmup, 1)):
        out = step()
    elapsed, out = parallel.timed_steps(step, args.steps, None, dev)
    if rank != 0:
        return None
    return {'metric': 'images/sec end-to-end (incl. decode+NMS), %s %dpx bs=%d' % (short_name(args.backbone), args.height, args.batch),
            'value': round(args.batch * world * args.steps / elapsed, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic randn images, random-init weights; CPU plumbing run -- NOT a measurement of the HIP path',
            'config': {'workload': '%s fp32 inference on the host cores, bs=%d per rank at %dx%d, pure-torch decode x5 + nms'
                                   % (args.backbone, args.batch, args.height, args.width),
                       'global_batch': args.batch * world, 'parallelism': 'replicas x%d (no data-path collective)' % world,
                       'entry': 'Model.forward (eval)', 'postproc': 'odtk/box.py CPU branch', 'device': 'cpu',
                       'backend': args.backend if world > 1 else None},
            'roofline': None, 'cpu_baseline': None, 'detections_shape': list(out[0].shape)}


def run_infer(args, rank, world, dev):
    from odtk import _C, parallel
    from odtk.model import Model
    miopen_find = not args.no_miopen_find

    amp_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': None}[args.dtype]
    torch.manual_seed(0)
    model = Model(backbones=args.backbone, classes=80, rotated_bbox=args.rotated_bbox)
    model.initialize(None)
    if args.rotated_bbox and args.unit_rotation:
        with torch.no_grad():
            bias = model.box_head[-1].bias.view(model.num_anchors, 6)
            bias.zero_()
            bias[:, 5] = 1.0
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    on_gpu = dev.type == 'cuda'
    model.fused_postprocess = args.postproc == 'fused'
    fuse_graph = not args.no_fuse and args.postproc == 'fused' and on_gpu
    model.fused_graph = fuse_graph
    if not on_gpu:
        return run_infer_cpu_plumbing(args, rank, world, dev, model, amp_dtype)

    g = torch.Generator(device='cpu').manual_seed(rank)
    x = torch.randn(args.batch, 3, args.height, args.width, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    flops_img = conv_flops_per_image(model, x[:1])

    def autocast():
        return torch.autocast('cuda', dtype=amp_dtype, enabled=amp_dtype is not None)

    def engine():
        e = model.inference_engine(amp_dtype or torch.float32)
        e.level_streams = not args.no_level_streams
        e.tower_plan = args.tower_plan
        return e

    def timed_heads(inp):
        """Head tensors (bias applied) of the path being timed."""
        if fuse_graph:
            return engine().heads(inp)
        with autocast():
            return model.heads(inp)

    sigma0, sigma1 = calibrate_cls_head(model, timed_heads, x, args.fraction, model.threshold)
    if fuse_graph:
        engine()                                             # re-fold once, outside the timed region

    def step():                                              # THE drop-in call: Model.forward in eval mode
        with torch.no_grad(), autocast():
            return model(x)

    out = None
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    n_det = int((out[0] > 0).sum().item())

    # time only the three post-processing launches inside the timed region (6 event records per step);
    # the ~110 epilogue launches per step are timed in a separate, untimed pass below
    post = ('prefilter_scan_kernel', 'select_decode_kernel', 'nms_kernel',
            'nms_first_round_kernel', 'rotated_sup_matrix_kernel')   # (the last two: rotated boxes only, five launches)
    post = tuple(k for k in post if k in _C.KERNEL_NAMES)
    _C.profile_enable(True, post)
    _C.profile_collect()
    # EXACTLY `steps` steps between barrier + device-sync brackets, MAX over ranks (tested on CPU with
    # gloo, world_size 2: tests/test_parallel_gloo.py)
    elapsed, out = parallel.timed_steps(step, args.steps, torch.cuda.synchronize, dev)
    _C.profile_enable(False)
    prof = _C.profile_collect()
    eager = None
    epilogue_roofline, marker_timed, quoted_epilogue_obj = None, None, None
    conv_epilogue = conv_epilogue_record(engine()) if (fuse_graph and rank == 0) else None
    if rank == 0:
        # The engine's own epilogue kernels.  Their algorithmic bytes (every element read once and written once, + the skip
        # input) are counted live over three untimed steps.  Their TIME cannot be taken live: an event pair handed to a launch
        # is stamped when the packet reaches the head of its queue, and an epilogue sits behind a long convolution -- the pair
        # then includes the predecessor's tail and cache write-back (13.6-17.8 us per bias_act launch where rocprofv3's kernel
        # trace of the same command measures 9.15; the post-processing launches, which follow short kernels, agree with it to
        # 0.5 %).  So the time comes from the committed rocprofv3 summary of THIS command (tools/profile_round.sh ->
        # profiles/r04_bench_steady_kernel_stats.csv), quoted only when the workload is the profiled one -- like `traffic`.
        epi = ('bias_act_kernel', 'bias_act_maxpool_kernel', 'upsample_nearest2x_kernel')
        _C.traffic_bytes.clear()
        _C.traffic_count = True
        _C.profile_enable(True, ('gemm_bias_act',))
        _C.profile_collect()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        _C.profile_enable(False)
        _C.traffic_count = False
        extra = _C.profile_collect()
        # (times of these kernels are never taken from a file into this object: what a committed rocprofv3 summary of this
        # command says about them goes under `quoted.epilogue_kernels`, with the file's name -- VERDICT r04 weak #11)
        profiled = {}
        stats_name = next((n for n in ('r06_bench_steady_kernel_stats.csv', 'r05_bench_steady_kernel_stats.csv') if os.path.isfile(os.path.join(ROOT, 'profiles', n))), None)
        stats = os.path.join(ROOT, 'profiles', stats_name or 'none')
        if default_workload(args) and stats_name:
            import csv
            rows = list(csv.reader(open(stats)))[1:]
            steps_profiled = next((int(r[1]) for r in rows if 'prefilter_scan_kernel' in r[0]), 0)
            for r in rows:
                for k in epi:
                    if k in r[0] and steps_profiled:
                        calls, total_ns = profiled.get(k, (0, 0))
                        profiled[k] = (calls + int(r[1]), total_ns + int(r[2]))
            profiled = {k: (c / steps_profiled, t / steps_profiled * 1e-3) for k, (c, t) in profiled.items()}
        epilogue_roofline, quoted_epilogue = {}, {}
        for k in epi:
            nbytes = _C.traffic_bytes.get(k, 0) / 3.0
            if not nbytes:
                continue
            epilogue_roofline[k] = {'alg_bytes_per_step': int(nbytes), 'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                    'measured': 'bytes counted live over three untimed steps; no time is measured in this run '
                                                '(an event pair behind a long convolution includes its tail): see quoted.epilogue_kernels'}
            if k in profiled:
                calls, us = profiled[k]
                # (launches and time of the PROFILED run only: the plan pass may route the k x k convolutions differently on
                #  another box, so today's byte count must not be divided by that run's time)
                quoted_epilogue[k] = {'launches_per_step': round(calls, 1), 'us_per_step': round(us, 1)}
        if quoted_epilogue:
            quoted_epilogue_obj = {'source': 'profiles/%s (rocprofv3 --kernel-trace --stats of this command, another run)' % stats_name,
                                   'kernels': quoted_epilogue}
        # hipBLASLt launches its own kernels: the library can only put marker packets around the call, and a marker pair
        # includes the dispatch latency on both sides -- NOT comparable with the dispatch-timestamp figures in `kernels`
        ms_g, n_g = extra['gemm_bias_act']
        if n_g:
            marker_timed = {'gemm_bias_act': {'avg_us': round(ms_g / n_g * 1e3, 2), 'launches': n_g,
                                              'note': 'event markers around the hipBLASLt call (dispatch latency included); '
                                                      "the kernels' own times are in profiles/ (rocprofv3 --kernel-trace)"}}
        if fuse_graph and world == 1 and not args.no_eager_leg:
            # the same Model.forward with fused_graph = False (eager nn.Module graph under autocast + the same HIP
            # post-processing): the A/B of what routing eval through the engine buys
            model.fused_graph = False
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            n_eager = max(10, args.steps // 5)
            t0 = time.perf_counter()
            for _ in range(n_eager):
                step()
            torch.cuda.synchronize()
            t_eager = time.perf_counter() - t0
            eager = {'value': round(args.batch * n_eager / t_eager, 2), 'unit': 'images/s',
                     'ms_per_step': round(t_eager / n_eager * 1e3, 3), 'steps': n_eager,
                     'graph': 'eager nn.Module under autocast (Model.fused_graph = False), same post-processing'}
            model.fused_graph = True

    images = args.batch * world * args.steps
    value = images / elapsed

    # ---- roofline of the dominant hand-written kernel ----
    with torch.no_grad():
        cls_heads, box_heads = timed_heads(x)                # the tensors the timed post-processing reads
        scores_per_batch = sum(c.numel() for c in cls_heads)
        candidates = [int((c.float().sigmoid() >= model.threshold).sum().item()) // args.batch for c in cls_heads]
    # every score is read exactly once, in the dtype the kernel consumes: the head's own dtype on the
    # fused path, fp32 on the reference-sequence path (after torch's .float())
    bytes_per_score = cls_heads[0].element_size() if model.fused_postprocess else 4
    alg_bytes = bytes_per_score * scores_per_batch
    ms, n = prof['prefilter_scan_kernel']
    roofline = None
    # HBM traffic per launch comes from PMC passes (cannot be collected live next to the timing):
    # profiles/r03_pmc_traffic.json (tools/profile_round.sh), quoted only when the workload matches the profiled one
    # It goes on the line only when the file was taken on THESE kernel sources (hash recorded by tools/pmc_traffic.py);
    # an older file is named under `quoted` with its mismatch and `roofline.traffic` stays null.
    traffic, traffic_src, quoted = None, None, None
    src_hash = kernel_src_hash()
    for name in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json'):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
            key = 'bf16_logits_channels_last' if (model.fused_postprocess and bytes_per_score == 2) else 'fp32_scores_nchw'
            if pmc[key]['scores_per_launch'] == scores_per_batch and (bytes_per_score == 2) == (key[0] == 'b'):
                same = pmc.get('kernel_src_sha16') == src_hash
                quoted = {'prefilter_traffic_bytes': pmc[key]['traffic_bytes'],
                          'ratio_to_algorithmic': round(pmc[key]['traffic_bytes'] / float(bytes_per_score * scores_per_batch), 4),
                          'source': 'profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, same workload)' % name,
                          'same_kernel_sources': same}
                if same:
                    traffic, traffic_src = pmc[key]['traffic_bytes'], 'quoted: ' + quoted['source']
                break
        except Exception:
            continue
    if quoted_epilogue_obj:
        quoted = dict(quoted or {})
        quoted['epilogue_kernels'] = quoted_epilogue_obj
    if n:
        avg_ms = ms / n
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        roofline = {'kernel': 'prefilter_scan_kernel', 'bound': 'hbm', 'achieved': round(achieved, 1),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                    'traffic': traffic, 'traffic_source': traffic_src, 'alg_bytes_per_launch': alg_bytes,
                    'bytes_per_score': bytes_per_score, 'avg_ms': round(avg_ms, 5), 'avg_us': round(avg_ms * 1e3, 2), 'launches': n,
                    'timing': 'hipEvent pairs on the launch stream inside the timed region (include/odtk_hip.h odtk_profile_*)'}
    kernels = {k: {'avg_us': round(v[0] / v[1] * 1e3, 2), 'launches': v[1]} for k, v in prof.items() if v[1]}
    # the latency-bound launches against their serial-chain lower bounds (one workgroup per image / per segment:
    # the launch time IS the per-image time)
    latency_bound = {}
    if 'nms_kernel' in kernels:
        lb = model.detections * NMS_CLK_PER_KEPT / (CLOCK_GHZ * 1e3)
        nms_parts = [k for k in ('nms_first_round_kernel', 'rotated_sup_matrix_kernel', 'nms_kernel') if k in kernels]
        nms_us = round(sum(kernels[k]['avg_us'] * kernels[k]['launches'] for k in nms_parts) / max(n, 1), 2)   # per step (rotated: 3 or 5 launches)
        latency_bound['nms_kernel'] = {'us_per_launch': nms_us, 'kernels': nms_parts,
                                       'launches_per_step': round(sum(kernels[k]['launches'] for k in nms_parts) / max(n, 1), 2),
                                       'us_per_image_throughput': round(nms_us / args.batch, 2),
                                       'lower_bound_us': round(lb, 2),
                                       'model': '%d kept boxes x %d clk (LDS read + dependent IoU chain + ballot/readlane) at %.1f GHz'
                                                % (model.detections, NMS_CLK_PER_KEPT, CLOCK_GHZ),
                                       'ratio': round(nms_us / lb, 1)}
    sel = [k for k in ('select_decode_kernel',) if k in kernels]
    if sel:
        total = sum(kernels[k]['avg_us'] * kernels[k]['launches'] for k in sel) / kernels['select_decode_kernel']['launches']
        latency_bound['select'] = {'us_per_step': round(total, 2), 'kernels': sel, 'lower_bound_us': 4.0,
                                   'model': 'one pass over the candidate keys out of L2 (~1 us) + one sort of the 1024 selected '
                                            'keys (~3 us); the same bound as in rounds 2-3, when the selection was three launches',
                                   'ratio': round(total / 4.0, 1)}
    conv_tflops = flops_img * (value / world) / 1e12
    conv_roofline = {'bound': 'mfma', 'achieved': round(conv_tflops, 1), 'peak': MFMA_BF16_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': round(conv_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                     'gflop_per_image': round(flops_img / 1e9, 1), 'per': 'gpu'}

    # ---- CPU baseline (rank 0, N=1): the hot path = decode x5 + nms of the reference's CPU algorithm on the
    # captured head tensors; and the whole pipeline (same model on the host cores + that post-processing) ----
    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and not args.rotated_bbox:
        from oracle import box_oracle      # the checker, timed as the reference's CPU path ("port")
        strides = [x.shape[-1] // c.shape[-1] for c in cls_heads]
        anchors = {s: box_oracle.generate_anchors(s, model.ratios, model.scales) for s in strides}
        # exactly what the reference's op receives (model.py:140,160; box.py:263): fp32 NCHW post-sigmoid scores
        cap_cls = [c.sigmoid().float().contiguous().cpu() for c in cls_heads]
        cap_box = [b.float().contiguous().cpu() for b in box_heads]
        t_budget0 = time.perf_counter()
        # torch's CPU ops on these sizes do not scale to every core of a big host (128 threads measured 6x SLOWER than 8 on
        # the GPU box): give the CPU its best thread count
        all_threads = torch.get_num_threads()
        trials = {}
        for nt in sorted({min(8, all_threads), min(32, all_threads), all_threads}):
            torch.set_num_threads(nt)
            best_nt = None
            for _ in range(2):
                t0 = time.perf_counter()
                box_oracle.postprocess([c[:1] for c in cap_cls], [b[:1] for b in cap_box], strides, anchors,
                                       model.threshold, model.top_n, model.nms, model.detections)
                dt = time.perf_counter() - t0
                best_nt = dt if best_nt is None else min(best_nt, dt)
            trials[nt] = best_nt
        cores = min(trials, key=trials.get)
        torch.set_num_threads(cores)
        per_image, oracle_out = [], []
        for i in range(args.batch):                          # every captured image once ...
            t0 = time.perf_counter()
            oracle_out.append(box_oracle.postprocess([c[i:i + 1] for c in cap_cls], [b[i:i + 1] for b in cap_box], strides, anchors,
                                                     model.threshold, model.top_n, model.nms, model.detections))
            per_image.append(time.perf_counter() - t0)
        # the checker doing its other job: the HIP op on the SAME captured head tensors (bf16 channels_last logits in place,
        # sigmoid inside the kernel -- the form the step runs) against what the CPU path just produced, all images
        try:
            from odtk import box as hip_box
            from oracle import box_check
            got = hip_box.detect(cls_heads, box_heads, strides, {s: a.to(dev) for s, a in anchors.items()}, model.threshold,
                                 model.top_n, model.nms, model.detections, logits=model.fused_postprocess) \
                if model.fused_postprocess else hip_box.detect([c.to(dev) for c in cap_cls], [b.to(dev) for b in cap_box], strides,
                                                               {s: a.to(dev) for s, a in anchors.items()}, model.threshold,
                                                               model.top_n, model.nms, model.detections)
            got = [t.float().cpu() for t in got]
            ref = [torch.cat([o[k] for o in oracle_out]) for k in range(3)]
            diff = (got[1] - ref[1]).abs()
            # what oracle/box_check.py's exp-rounding escape would have to excuse on these boxes: coordinates beyond 1e-4,
            # split into those beyond one fp32 ulp of the coordinate (counted against its bound of 2 per call) and the
            # one-ulp deviations at >= 1024 px (one ulp IS 1.2e-4 there: proven, not counted) -- VERDICT r05 weak #1
            over = diff > box_check.NORTH_STAR_ATOL
            one_ulp = torch.from_numpy(np.spacing(ref[1].abs().numpy())).to(diff.dtype)
            parity = {'images': args.batch, 'scores_bit_exact': bool(torch.equal(got[0], ref[0])),
                      'classes_bit_exact': bool(torch.equal(got[2], ref[2])), 'max_box_abs_diff': float(diff.max()),
                      'box_coords_beyond_1e-4': int(over.sum()), 'box_coords': int(diff.numel()),
                      'proven': {'beyond_one_ulp': int((over & (diff > one_ulp * (1 + 1e-6))).sum()),
                                 'one_ulp_only': int((over & ~(diff > one_ulp * (1 + 1e-6))).sum())},
                      'checker': 'oracle/box_oracle.py (restatement of reference odtk/box.py, pinned) on the captured heads'}
        except Exception as e:                               # noqa: BLE001 -- the check reports, it never takes the line down
            parity = {'error': '%s: %s' % (type(e).__name__, e)}
        best = min(per_image)
        for _ in range(4):                                   # ... and image 0 four more times: best of 5
            t0 = time.perf_counter()
            box_oracle.postprocess([c[:1] for c in cap_cls], [b[:1] for b in cap_box], strides, anchors,
                                   model.threshold, model.top_n, model.nms, model.detections)
            best = min(best, time.perf_counter() - t0)
        mean = sum(per_image) / len(per_image)
        gpu_post_us = sum(kernels[k]['avg_us'] * kernels[k]['launches'] for k in post if k in kernels) / max(n, 1)
        postproc = {'value': round(1.0 / mean, 2), 'unit': 'images/s', 'ms_per_image': round(mean * 1e3, 2),
                    'best_ms_per_image': round(best * 1e3, 2), 'cores': cores, 'kind': 'port',
                    'threads_tried_ms': {str(k): round(v * 1e3, 1) for k, v in trials.items()},
                    'sample': 'oracle decode x5 + nms (restatement of reference odtk/box.py:255-367, pinned to it) on the %d '
                              'captured images of the timed batch, fp32 NCHW post-sigmoid scores; best = best of 5 on image 0'
                              % args.batch,
                    'gpu_us_per_image': round(gpu_post_us / args.batch, 2),
                    'gpu_vs_cpu': round(mean * 1e6 / max(gpu_post_us / args.batch, 1e-9), 1)}
        # the whole pipeline on the CPU: convolutions want many threads, torch's decode / nms ops few (above) -- try a
        # middle and the full count once each (after a warm-up call that creates the oneDNN primitives), keep the better
        model.__dict__['_engine_cache'].clear()
        cpu_model = copy.deepcopy(model).float().cpu().eval()
        xc = x[:1].float().cpu().contiguous(memory_format=torch.channels_last)

        def cpu_image():
            t0 = time.perf_counter()
            with torch.no_grad():
                cpu_model(xc)                                # eager graph + the CPU branch of odtk/box.py
            return time.perf_counter() - t0

        pipeline_trials = {}
        for nt in sorted({min(32, all_threads), all_threads}):
            torch.set_num_threads(nt)
            cpu_image()
            pipeline_trials[nt] = cpu_image()
        pipeline_threads = min(pipeline_trials, key=pipeline_trials.get)
        torch.set_num_threads(pipeline_threads)
        times = [pipeline_trials[pipeline_threads]]
        while time.perf_counter() - t_budget0 < args.cpu_seconds and len(times) < 8:
            times.append(cpu_image())
        torch.set_num_threads(all_threads)
        cpu_baseline = {'value': round(len(times) / sum(times), 4), 'unit': 'images/s', 'cores': pipeline_threads, 'kind': 'port',
                        'sample': '%d x (1 image %dx%d: %s fp32 forward on the host cores + pure-torch decode x5 + nms), %.1f s; '
                                  'threads tried (s per image): %s'
                                  % (len(times), args.height, args.width, args.backbone, sum(times),
                                     {k: round(v, 2) for k, v in pipeline_trials.items()}),
                        'postproc': postproc}

    line = None
    if rank == 0:
        line = {
            'metric': 'images/sec end-to-end (incl. decode+NMS), %s%s %dpx bs=%d' % (
                short_name(args.backbone), ' --rotated-bbox' if args.rotated_bbox else '', args.height, args.batch),
            'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic randn images, random-init weights; last cls conv rescaled so that %.4f %% of the scores '
                    'are >= %.2f (SURVEY 8d sparse-realistic; logit sigma %.4f -> %.4f); %d detections in the last batch'
                    % (100 * args.fraction, model.threshold, sigma0, sigma1, n_det),
            'config': {'workload': '%s %s inference%s, bs=%d per GPU at %dx%d, HIP decode x5 + NMS'
                                   % (args.backbone, args.dtype, (' --rotated-bbox' + (' (unit (sin, cos) head bias)' if args.unit_rotation else ''))
                                      if args.rotated_bbox else '',
                                      args.batch, args.height, args.width),
                       'global_batch': args.batch * world, 'image': [args.height, args.width],
                       'parallelism': 'replicas x%d (no data-path collective)' % world,
                       'memory_format': 'channels_last', 'miopen_find': miopen_find, 'postproc': args.postproc,
                       'entry': 'Model.forward (eval)',
                       'graph': 'BN folded into conv weights + HIP bias/skip/ReLU epilogue + 1x1 convs as fused GEMMs' if fuse_graph
                                else 'eager nn.Module under autocast'},
            'roofline': roofline, 'latency_bound': latency_bound, 'conv_roofline': conv_roofline, 'kernels': kernels,
            'postproc_us_per_step': round(sum(kernels[k]['avg_us'] * kernels[k]['launches'] for k in post if k in kernels) / max(n, 1), 2),
            'quoted': quoted, 'parity': parity, 'kernel_src_sha16': src_hash,
            'epilogue_roofline': epilogue_roofline, 'marker_timed': marker_timed, 'conv_epilogue': conv_epilogue,
            'candidates_per_image_per_level': candidates,
            'spec_candidates_per_image_per_level': SPEC_CANDIDATES if (args.height, args.width) == (800, 1280) and not args.rotated_bbox else None,
            'eager': eager,
            'cpu_baseline': cpu_baseline,
        }
    model.__dict__['_engine_cache'].clear()
    return line


def conv_epilogue_record(engine):
    """What the engine's plan pass decided for its k x k convolutions (odtk/fused.py: `_Conv.route`): per layer and input
    shape the time of the ONE-launch form (composable_kernel convolution with bias + ReLU in its epilogue, csrc/conv_ck.cpp)
    and of the TWO-launch form (MIOpen convolution + odtk_bias_act), both measured back to back on this GPU during the
    plan pass, and which one the step runs."""
    routes = engine.conv_routes()
    if not routes:
        return None
    layers, one, two, saved, n_lib, n_timed = {}, 0.0, 0.0, 0.0, 0, 0
    for name, per_shape in sorted(routes.items()):
        for shape, (use, t_one, t_two) in per_shape.items():
            key = '%s %s' % (name, ('conv only ' if shape[0] == 'only' else '') + 'x'.join(str(v) for v in shape if v != 'only'))
            n_lib += bool(use)
            if t_one is None or t_two is None:               # pinned by a loaded plan / ODTK_CONV_ROUTE: nothing was timed
                layers[key] = {'library': bool(use), 'pinned': True}
                continue
            layers[key] = {'library': bool(use), 'us_one_launch': None if t_one == float('inf') else round(t_one, 1),
                           'us_two_launches': round(t_two, 1)}
            n_timed += 1
            two += t_two
            one += min(t_one, t_two)
            saved += max(t_two - t_one, 0.0)
    from odtk import fused
    state = engine.plan_state()
    return {'layers_routed_to_library': n_lib, 'layers_measured': len(layers), 'layers_timed_here': n_timed,
            'us_per_step_two_launch_form': round(two, 1), 'us_per_step_as_routed': round(one, 1), 'us_saved_per_step': round(saved, 1),
            # the plan the step ran: routes + the instance / solution each library problem runs on (odtk/fused.py: plan_state);
            # ODTK_CONV_PLAN=<file> replays it on another box bit for bit where the libraries still offer those kernels
            'plan_hash': engine.plan_hash(state), 'plan_source': ('file ' + os.environ['ODTK_CONV_PLAN']) if os.environ.get('ODTK_CONV_PLAN')
            else 'route mode ' + fused._Conv.route_mode, 'library_lines': len(state['libraries']),
            'note': 'per-layer A/B of the plan pass (median of 5, back to back, caller stream only): shared tower layers run once '
                    'per pyramid level and are listed per input shape', 'layers': layers, 'plan': state}


if __name__ == '__main__':
    main()

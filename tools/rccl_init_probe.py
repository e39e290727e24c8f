#!/usr/bin/env python
"""Does RCCL initialise on this driver stack with the environment bench.py / odtk main set (HSA_ENABLE_IPC_MODE_LEGACY=0)?
The GPU boxes of this project have ONE GPU, so the N > 1 data path has only ever run on gloo (tests/test_*_gloo.py); this
probe takes the part that CAN run on one GPU through the real backend: process-group initialisation over `nccl` (= RCCL on
ROCm) with world size 1, an all_reduce / all_gather / barrier on device tensors, and one DDP training step of the tiny model
through `odtk.train.prepare(world=...)`'s DistributedDataParallel wrapper (forced on at world size 1).

    python tools/rccl_init_probe.py > profiles/r05_rccl_init_probe.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29517')
import torch
import torch.distributed as dist

t0 = time.perf_counter()
dist.init_process_group('nccl', rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.ones(1 << 20, device='cuda')
dist.all_reduce(x)
out = [torch.empty(4, device='cuda')]
dist.all_gather(out, torch.arange(4.0, device='cuda'))
dist.barrier()
torch.cuda.synchronize()
print('backend %s, world %d, HSA_ENABLE_IPC_MODE_LEGACY=%s' % (dist.get_backend(), dist.get_world_size(), os.environ['HSA_ENABLE_IPC_MODE_LEGACY']))
print('nccl (RCCL) version', torch.cuda.nccl.version())
print('all_reduce of ones -> %.1f, all_gather -> %s, init + collectives %.2f s' % (float(x[0]), out[0].tolist(), time.perf_counter() - t0))

from torch.nn.parallel import DistributedDataParallel
from odtk import train as T
from odtk.model import Model
torch.manual_seed(0)
model = Model('ResNet18FPN', classes=8)
model.initialize(None)
model, net, opt, sched = T.prepare(model, torch.device('cuda', 0), lr=0.01, world=1, rank=0, warmup=10)
net = DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True, static_graph=True, bucket_cap_mb=25)
src = T.SyntheticBatches(2, 256, 320, classes=8, max_boxes=6, seed=0, device='cuda')
for step in range(3):
    d, t = src.batch()
    c, b = T.train_step(net, opt, sched, None, d.contiguous(memory_format=torch.channels_last), t, None)
    print('DDP step %d over RCCL (world 1): focal %.4f box %.4f' % (step, float(c), float(b)))
dist.destroy_process_group()
print('ok')

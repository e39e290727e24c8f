"""Multi-GPU plumbing of the inference path: one process per GPU, images sharded across ranks,
NO collective in the data path (reference odtk/main.py:155-195 spawns one worker per
GPU, odtk/infer.py:95-102 gathers the final detections once, at the end).

Backend is "nccl" (= RCCL over xGMI on ROCm) on GPUs and "gloo" in the CPU tests; nothing here
depends on which.
"""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process = 1 GPU)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init_from_env(backend):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend, init_method='env://', rank=rank, world_size=world)
    return rank, local_rank, world


def barrier(device=None):
    """Rendezvous of all ranks.  With a device-side backend (RCCL) this is an explicit 1-element
    all_reduce on `device` (never relies on a 'current device' guess) followed by a device sync."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    if device is not None and torch.device(device).type == 'cuda':
        t = torch.zeros(1, device=device)
        dist.all_reduce(t)
        torch.cuda.synchronize(device)
    else:
        dist.barrier()


def timed_steps(step, steps, device_sync=None, device=None):
    """Run `step()` exactly `steps` times between barrier + device-sync brackets and return the
    MAX over ranks of the elapsed seconds (every rank gets the same number)."""
    sync = device_sync or (lambda: None)
    barrier(device)
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    barrier(device)
    elapsed = time.perf_counter() - t0
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def shard_batch(global_batch, rank, world):
    """Per-rank batch of a data-parallel run; the reference insists on divisibility (main.py:170-171)."""
    if global_batch % world != 0:
        raise RuntimeError('Batch size should be a multiple of the number of GPUs')
    per = global_batch // world
    return per, range(rank * per, (rank + 1) * per)


def is_master():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def pack_detections(scores, boxes, classes, ids, ratios):
    """[N, D], [N, D, nb], [N, D], [N], [N] -> one [N, D*(nb+2) + 2] float32 tensor, so the final
    hand-off is ONE all_gather instead of the reference's five (infer.py:98-102).  The image id rides as
    the BIT PATTERN of an int32 (ids above 2^24 would not survive a float conversion)."""
    n, d = scores.shape
    id_bits = ids.reshape(n, 1).to(torch.int32).view(torch.float32)
    return torch.cat([scores.reshape(n, -1), boxes.reshape(n, -1), classes.reshape(n, -1),
                      id_bits.to(scores.device), ratios.reshape(n, 1).to(scores.dtype)], 1)


def unpack_detections(packed, detections, nb=4):
    n = packed.shape[0]
    d = detections
    scores = packed[:, :d]
    boxes = packed[:, d:d + d * nb].reshape(n, d, nb)
    classes = packed[:, d + d * nb:d + d * nb + d]
    ids = packed[:, -2].contiguous().view(torch.int32).long()
    ratios = packed[:, -1]
    return scores, boxes, classes, ids, ratios


def gather_packed(packed):
    """Rank-major concatenation of every rank's packed rows (equal per-rank counts, as DistributedSampler
    guarantees): ONE all_gather.  world == 1: the input."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return packed
    packed = packed.contiguous()
    out = [torch.empty_like(packed) for _ in range(dist.get_world_size())]
    dist.all_gather(out, packed)
    return torch.cat(out, 0)


def gather_detections(scores, boxes, classes, ids, ratios):
    """All ranks' detections on every rank, rank-major.  world == 1: returns the inputs."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return scores, boxes, classes, ids, ratios
    packed = gather_packed(pack_detections(scores, boxes, classes, ids, ratios))
    return unpack_detections(packed, scores.shape[1], boxes.shape[-1])

"""Config 3 (data-parallel training) on CPU with gloo, world_size 2: the DDP-wrapped RetinaNet step
(frozen BN, FocalLoss + SmoothL1, target assignment) produces, on every rank, the AVERAGE of the
per-rank gradients -- i.e. the all-reduce path is wired correctly for this model (unused `fc`
parameters, buffers not broadcast, static graph), and one optimiser step keeps the replicas identical."""
import copy
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from odtk import train as T
from odtk.model import Model


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build():
    torch.manual_seed(0)
    m = Model('ResNet18FPN', classes=4)
    m.initialize(None)
    return m


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cpu')
    batches = [T.SyntheticBatches(2, 128, 128, classes=4, max_boxes=4, seed=3, rank=r, world=world).batch()
               for r in range(world)]

    # expected: mean over ranks of the single-process gradients
    expected = None
    for r in range(world):
        ref, _, _, _ = T.prepare(_build(), dev, lr=0.01, world=1)
        cls_loss, box_loss = ref([batches[r][0], batches[r][1]])
        (cls_loss + box_loss).backward()
        grads = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        expected = grads if expected is None else {n: expected[n] + g for n, g in grads.items()}
    expected = {n: g / world for n, g in expected.items()}

    model, net, opt, sched = T.prepare(_build(), dev, lr=0.01, world=world, rank=rank, warmup=10)
    assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
    data, target = batches[rank]
    cls_loss, box_loss = net([data, target])
    (cls_loss + box_loss).backward()
    worst = 0.0
    for n, p in model.named_parameters():
        if p.grad is None:
            assert 'fc' in n or not p.requires_grad, n
            continue
        scale = expected[n].abs().max().item() + 1e-12
        worst = max(worst, (p.grad - expected[n]).abs().max().item() / scale)
    # a full optimisation step through the public helper, then replicas must still agree
    c, b = T.train_step(net, opt, sched, None, data, target)
    both = T.reduce_losses(c, b, world)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, worst, bool(torch.equal(gathered[0], gathered[1])), both.tolist(), sched.get_last_lr()[0]))
    dist.destroy_process_group()


def test_ddp_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, worst, same, losses, lr in res:
        assert worst < 1e-4, worst            # DDP gradient == mean of per-rank gradients (fp32 noise)
        assert same                           # replicas identical after the step
        assert all(l == l and l > 0 for l in losses)
        assert abs(lr - (0.9 * 1 / 10 + 0.1) * 0.01) < 1e-9
    assert res[0][3] == res[1][3]             # the reduced losses are the same on both ranks


def test_lr_schedule_and_synthetic_batches():
    f = T.lr_schedule(100, [200, 300], 0.1)
    assert abs(f(0) - 0.1) < 1e-12 and abs(f(50) - 0.55) < 1e-12 and f(100) == 1.0
    assert abs(f(250) - 0.1) < 1e-12 and abs(f(300) - 0.01) < 1e-12
    a = T.SyntheticBatches(4, 64, 96, seed=1, rank=0, world=2).batch()
    b = T.SyntheticBatches(4, 64, 96, seed=1, rank=1, world=2).batch()
    assert a[0].shape == (2, 3, 64, 96) and a[1].shape == (2, 20, 5) and not torch.equal(a[0], b[0])
    valid = a[1][a[1][:, :, 4] >= 0]
    assert (valid[:, 0] + valid[:, 2] <= 96 + 1e-3).all() and (valid[:, 1] + valid[:, 3] <= 64 + 1e-3).all()

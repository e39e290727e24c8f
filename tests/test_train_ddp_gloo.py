"""Config 3 (data-parallel training) on CPU with gloo, world_size 2: the DDP-wrapped RetinaNet step
(frozen BN, FocalLoss + SmoothL1, target assignment) produces, on every rank, the AVERAGE of the
per-rank gradients -- i.e. the all-reduce path is wired correctly for this model (unused `fc`
parameters, buffers not broadcast, static graph), and one optimiser step keeps the replicas identical."""
import copy
import os
import socket

import pytest
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


def _quiet_worker(rank, world, port, q):
    """bench.py run_train's `exposed_allreduce_ms`: steps through the DDP wrapper, then the SAME step on the bare module (same
    parameters and optimizer, no gradient all-reduce).  DDP's gradient hooks stay registered on the parameters; outside a DDP
    forward they must do nothing -- and the bare step must be exactly the step DDP's no_sync() would take."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cpu')
    data, target = T.SyntheticBatches(2, 128, 128, classes=4, max_boxes=4, seed=3, rank=rank, world=world).batch()
    out = []
    for mode in ('bare', 'no_sync'):
        model, net, opt, sched = T.prepare(_build(), dev, lr=0.001, world=world, rank=rank, warmup=10)
        for _ in range(2):
            T.train_step(net, opt, sched, None, data, target)
        if mode == 'bare':
            for _ in range(2):
                c, b = T.train_step(model, opt, sched, None, data, target)
        else:
            with net.no_sync():
                for _ in range(2):
                    c, b = T.train_step(net, opt, sched, None, data, target)
        out.append((float(c), float(b), torch.cat([p.detach().flatten() for p in model.parameters()])))
    # (run-to-run noise of the CPU convolution backward is ~1e-20 on a few hundred parameters: compare to 1e-12, not bit for bit)
    q.put((rank, out[0][:2], out[1][:2], float((out[0][2] - out[1][2]).abs().max())))
    dist.destroy_process_group()


def test_bare_module_steps_after_ddp_steps_equal_no_sync_steps():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_quiet_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, bare, no_sync, worst in res:
        assert all(abs(x - y) <= 1e-6 * abs(y) for x, y in zip(bare, no_sync)) and all(v == v and v > 0 for v in bare)
        assert worst <= 1e-12                 # the same parameters after the quiet steps either way
    assert res[0][1] != res[1][1]             # no all-reduce in the quiet steps: the replicas have drifted apart


def test_lr_schedule_and_synthetic_batches():
    f = T.lr_schedule(100, [200, 300], 0.1)
    assert abs(f(0) - 0.1) < 1e-12 and abs(f(50) - 0.55) < 1e-12 and f(100) == 1.0
    assert abs(f(250) - 0.1) < 1e-12 and abs(f(300) - 0.01) < 1e-12
    a = T.SyntheticBatches(4, 64, 96, seed=1, rank=0, world=2).batch()
    b = T.SyntheticBatches(4, 64, 96, seed=1, rank=1, world=2).batch()
    assert a[0].shape == (2, 3, 64, 96) and a[1].shape == (2, 20, 5) and not torch.equal(a[0], b[0])
    valid = a[1][a[1][:, :, 4] >= 0]
    assert (valid[:, 0] + valid[:, 2] <= 96 + 1e-3).all() and (valid[:, 1] + valid[:, 3] <= 64 + 1e-3).all()


# ---------------------------------------------------------------------------------------------------
# train(): the whole loop on two ranks (the logging collective must be entered by every rank
# at the same iteration whatever each rank's own clock says; epochs must wrap; checkpoints must carry
# the reference's optimizer layout = ALL parameters)
# ---------------------------------------------------------------------------------------------------
class _SlowFiniteSource:
    """3 batches per epoch; rank 1 is slower than rank 0, so wall-clock based decisions would diverge."""

    def __init__(self, rank, world):
        self.src = T.SyntheticBatches(2, 128, 128, classes=4, max_boxes=4, seed=5, rank=rank, world=world)
        self.items = [self.src.batch() for _ in range(3)]
        self.rank = rank
        self.epochs = 0

    def __iter__(self):
        import time
        self.epochs += 1
        for item in self.items:
            time.sleep(0.05 * self.rank)
            yield item


def _train_worker(rank, world, port, q, tmp):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _build()
    source = _SlowFiniteSource(rank, world)
    path = os.path.join(tmp, 'ckpt.pth')
    # log_every far below one step: a per-rank clock would fire on different iterations on the two ranks
    done = T.train_batches(model, {}, source, 8, torch.device('cpu'), lr=0.001, warmup=4, world=world, rank=rank,
                   mixed_precision=False, log_every=0.01, save_path=path, verbose=False)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, done, source.epochs, bool(torch.equal(gathered[0], gathered[1])), bool(torch.isfinite(flat).all())))
    dist.destroy_process_group()


def test_train_loop_two_ranks_gloo(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, done, epochs, same, finite in res:
        assert done == 8                      # all iterations ran although an epoch holds only 3 batches
        assert epochs == 3                    # 3 + 3 + 2
        assert same and finite                # replicas identical after 8 steps: no collective was mispaired
    # the checkpoint resumes: optimizer state covers EVERY parameter (reference layout, train.py:34)
    model, state = Model.load(os.path.join(str(tmp_path), 'ckpt.pth'))
    assert state['iteration'] == 8
    m2, net, opt, sched = T.prepare(model, torch.device('cpu'), lr=0.001, warmup=4, state=state)   # load_state_dict ok
    n_params = len(list(m2.parameters()))                    # frozen-BN conversion turns BN affines into buffers
    assert any(not p.requires_grad for p in m2.parameters())  # the unused `fc` is frozen, yet part of the layout
    assert len(state['optimizer']['param_groups'][0]['params']) == n_params
    assert sched.last_epoch == 8


def _trajectory_worker(rank, world, port, q, steps, every_level):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cpu')
    data, target = _trajectory_batches(steps, every_level)
    model, net, opt, sched = T.prepare(_build(), dev, lr=0.01, world=world, rank=rank, warmup=2)
    losses = []
    for s in range(steps):
        c, b = T.train_step(net, opt, sched, None, data[s][rank:rank + 1], target[s][rank:rank + 1])
        losses.append(T.reduce_losses(c, b, world).tolist())
    # numpy, not a tensor: a tensor travels as a file descriptor the parent must fetch while this process is alive
    q.put((rank, losses, torch.cat([p.detach().flatten() for p in model.parameters()]).numpy()))
    dist.destroy_process_group()


def _trajectory_batches(steps, every_level=True):
    """Global batches of 2 images.

    every_level=True: both images carry the SAME five boxes, one sitting exactly on an anchor of each
    pyramid level.  The reference normalises the loss by the foreground count of the rank's OWN batch,
    clamped to >= 1 PER LEVEL (model.py:196, :206-209), so a data-parallel run equals the single-process
    bs-2 run exactly when the per-rank foreground counts agree and no level is empty -- which this
    construction guarantees.  every_level=False: random boxes (levels without foreground exist)."""
    g = torch.Generator().manual_seed(11)
    data, target = [], []
    for _ in range(steps):
        d = torch.randn(2, 3, 128, 128, generator=g)
        if every_level:
            rows = [[-1.5 * s, -1.5 * s, 4.0 * s + 1, 4.0 * s + 1, float(i % 4)] for i, s in enumerate((8, 16, 32, 64, 128))]
            t = torch.tensor(rows).unsqueeze(0).repeat(2, 1, 1)
        else:
            t = torch.full((2, 4, 5), -1.0)
            for i in range(2):
                n = int(torch.randint(1, 4, (1,), generator=g))
                wh = torch.rand(n, 2, generator=g) * 60 + 20
                xy = torch.rand(n, 2, generator=g) * (128 - wh)
                t[i, :n] = torch.cat([xy, wh, torch.randint(0, 4, (n, 1), generator=g).float()], 1)
        data.append(d)
        target.append(t)
    return data, target


def _single_process_trajectory(steps, every_level):
    threads = torch.get_num_threads()
    torch.set_num_threads(2)                 # like the workers: the CPU convolutions' summation order depends on it
    try:
        return _single_process_trajectory_impl(steps, every_level)
    finally:
        torch.set_num_threads(threads)


def _single_process_trajectory_impl(steps, every_level):
    data, target = _trajectory_batches(steps, every_level)
    model, net, opt, sched = T.prepare(_build(), torch.device('cpu'), lr=0.01, world=1, warmup=2)
    single = []
    for s in range(steps):
        if every_level:
            c, b = T.train_step(net, opt, sched, None, data[s], target[s])        # ONE bs-2 forward
        else:
            # two micro-batches of one image, gradients averaged: what data parallelism computes by definition
            opt.zero_grad(set_to_none=True)
            parts = [net([data[s][i:i + 1], target[s][i:i + 1]]) for i in range(2)]
            c = (parts[0][0] + parts[1][0]) / 2
            b = (parts[0][1] + parts[1][1]) / 2
            (c + b).backward()
            opt.step()
            sched.step()
        single.append([float(c.detach()), float(b.detach())])
    return single, torch.cat([p.detach().flatten() for p in model.parameters()])


@pytest.mark.parametrize('every_level', [True, False], ids=['vs-one-bs2-forward', 'vs-accumulated-microbatches'])
def test_two_ranks_reproduce_one_rank_trajectory(every_level):
    """2 ranks x 1 image == 1 rank x 2 images, loss for loss, for 3 steps."""
    steps, world, port = 3, 2, _free_port()
    single, single_params = _single_process_trajectory(steps, every_level)

    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_trajectory_worker, args=(r, world, port, q, steps, every_level)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]
    for got, want in zip(res[0][1], single):
        for g_, w_ in zip(got, want):
            assert abs(g_ - w_) <= 1e-4 * max(1.0, abs(w_)), (res[0][1], single)     # fp32 noise of different conv batch shapes
    scale = single_params.abs().max().item()
    assert (torch.from_numpy(res[0][2]) - single_params).abs().max().item() <= 1e-4 * scale

"""world_size-2 CPU test (gloo) of the N>1 path bench.py / inference use: image sharding, the
barrier + MAX-over-ranks timing bracket, and the single packed all_gather of the detections
(no collective in the data path)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from odtk import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, lr, w = parallel.init_from_env('gloo')
    assert (r, w) == (rank, world)
    per, idx = parallel.shard_batch(16, rank, world)

    calls = []

    def step():                       # rank 1 is the slow one
        calls.append(1)
        time.sleep(0.02 * (rank + 1))
        return rank

    elapsed, last = parallel.timed_steps(step, 5)
    # fake per-rank detections, tagged by global image id
    d = 7
    ids = torch.tensor(list(idx), dtype=torch.int32)
    scores = torch.rand(per, d) + rank
    boxes = torch.rand(per, d, 4) * 100
    classes = torch.randint(0, 80, (per, d)).float()
    ratios = torch.full((per,), 0.5 + rank)
    g = parallel.gather_detections(scores, boxes, classes, ids, ratios)
    ok = (g[0].shape == (16, d) and g[1].shape == (16, d, 4) and g[3].tolist() == list(range(16))
          and torch.equal(g[0][rank * per:(rank + 1) * per], scores)
          and torch.equal(g[1][rank * per:(rank + 1) * per], boxes)
          and torch.equal(g[4][rank * per:(rank + 1) * per], ratios))
    q.put((rank, per, list(idx), len(calls), elapsed, last, bool(ok)))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, per0, idx0, n0, t0, last0, ok0), (r1, per1, idx1, n1, t1, last1, ok1) = res
    assert per0 == per1 == 8 and idx0 == list(range(8)) and idx1 == list(range(8, 16))
    assert n0 == n1 == 5 and (last0, last1) == (0, 1)
    assert t0 == t1                       # MAX over ranks, identical everywhere
    assert t0 >= 5 * 0.04 * 0.9           # ... and it is the slow rank's time
    assert ok0 and ok1


def test_single_process_is_a_noop():
    e, out = parallel.timed_steps(lambda: 3, 4)
    assert out == 3 and e >= 0
    s, b, c = torch.rand(2, 3), torch.rand(2, 3, 6), torch.rand(2, 3)
    i, r = torch.tensor([4, 9]), torch.tensor([1.0, 2.0])
    packed = parallel.pack_detections(s, b, c, i, r)
    assert packed.shape == (2, 3 * 8 + 2)
    u = parallel.unpack_detections(packed, 3, 6)
    assert torch.equal(u[0], s) and torch.equal(u[1], b) and torch.equal(u[2], c) and u[3].tolist() == [4, 9]
    with pytest.raises(RuntimeError):
        parallel.shard_batch(10, 0, 4)


class _Stub:
    """Detections that encode (image id's batch position, rank) so the gathered order can be checked."""
    def __init__(self, rank):
        self.rank = rank

    def __call__(self, images):
        n = images.shape[0]
        scores = torch.tensor([[0.9, 0.5, 0.0]]).repeat(n, 1) + 0.01 * self.rank
        scores[:, 2] = 0
        boxes = torch.zeros(n, 3, 4)
        boxes[:, :, 2:] = 10 + self.rank
        return scores, boxes, torch.full((n, 3), float(self.rank))


def _infer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from odtk import infer
    parallel.init_from_env('gloo')
    # rank r owns images 100*r + {0, 1, 2, 3} in two batches of two
    batches = [(torch.zeros(2, 3, 4, 4), torch.tensor([100 * rank + 2 * k, 100 * rank + 2 * k + 1]), torch.ones(2)) for k in range(2)]
    dets = infer.infer_batches(_Stub(rank), batches)
    q.put((rank, None if dets is None else [(d['image_id'], d['category_id'], d['bbox'][2]) for d in dets]))
    dist.destroy_process_group()


def test_infer_driver_two_ranks_one_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_infer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[1] is None                                    # only the master converts
    ids = [d[0] for d in res[0]]
    assert ids == [i for r in range(2) for k in range(4) for i in [100 * r + k] * 2]   # rank-major, 2 detections each
    assert all(cat == (0 if i < 100 else 1) and w == (11.0 if i < 100 else 12.0) for i, cat, w in res[0])

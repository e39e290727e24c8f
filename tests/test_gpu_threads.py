"""The boundary's threading promise (include/odtk_hip.h): calls only ENQUEUE on the caller's stream and may be
made concurrently from several host threads on distinct streams and workspaces.  Four host threads, each with its own HIP
stream and its own inputs, run `detect` (prefilter -> select_decode -> nms) and the fused loss pair (all-level forward through
the workspace form + backward) 200 times -- through the ctypes binding and through the compiled `_C_ext` module -- and every
result must equal, bit for bit, what the same call returned serially.  Process-wide state this exercises: the profiler's
event pool (enabled in half of the threads' iterations), hipFuncSetAttribute bookkeeping, the binding's workspace cache
(a lock since round 4), the loss launch-shape table (one snapshot per call since round 4)."""
import threading

import pytest
import torch

from odtk import _C, box, synthetic

pytestmark = pytest.mark.gpu
RATIOS, SCALES = [1.0, 2.0, 0.5], [4 * 2 ** (i / 3) for i in range(3)]
N_THREADS, ITERS = 4, 200


def _inputs(seed):
    cls, dl, strides = synthetic.pyramid(2, 9, 20, 192, 256, 'clustered', seed, unique=False)
    lg = [torch.logit(c.clamp(1e-6, 1 - 1e-6)).cuda().bfloat16().contiguous(memory_format=torch.channels_last) for c in cls]
    db = [d.cuda().bfloat16().contiguous(memory_format=torch.channels_last) for d in dl]
    anchors = {s: box.generate_anchors(s, RATIOS, SCALES) for s in strides}
    g = torch.Generator().manual_seed(seed)
    depths = [torch.randint(-1, 21, (2, 9, 1) + tuple(c.shape[-2:]), generator=g).float().cuda() for c in cls]
    targets = [torch.randn((2, 9, 4) + tuple(c.shape[-2:]), generator=g).cuda() * 0.3 for c in cls]
    return lg, db, strides, anchors, depths, targets


def _work(inp, ext):
    lg, db, strides, anchors, depths, targets = inp
    if ext:
        from odtk import _C_ext
        det = _C_ext.detect(lg, db, [anchors[s].reshape(-1).tolist() for s in strides], strides, 0.05, 1000, 0.5, 100, False, True)
    else:
        det = box.detect(lg, db, strides, anchors, 0.05, 1000, 0.5, 100, logits=True)
    sums = _C.retina_loss_levels_forward(lg, db, depths, targets, 0.25, 2.0, 0.11, reproducible=True)
    gc = torch.full((len(lg),), 0.5, device='cuda')
    dcls, dbox = _C.retina_loss_levels_backward(lg, db, depths, targets, 0.25, 2.0, 0.11, gc, gc)
    return list(det) + [sums] + list(dcls) + list(dbox)


@pytest.mark.parametrize('ext', [False, True], ids=['ctypes', 'compiled'])
def test_concurrent_calls_on_distinct_streams_equal_serial_results(ext):
    inputs = [_inputs(1000 + t) for t in range(N_THREADS)]
    serial = [_work(inp, ext) for inp in inputs]
    torch.cuda.synchronize()
    assert all(int((s[0] > 0).sum()) > 20 for s in serial)
    streams = [torch.cuda.Stream() for _ in range(N_THREADS)]
    errors = []
    start = threading.Barrier(N_THREADS)

    def run(t):
        try:
            torch.cuda.set_device(0)
            start.wait()
            with torch.cuda.stream(streams[t]):
                for it in range(ITERS):
                    if t % 2 == 0 and it % 50 == 0:
                        _C.profile_enable(it % 100 == 0)             # the profiler's pool is process-wide state too
                    out = _work(inputs[t], ext)
                    if it % 20 == 19 or it == ITERS - 1:
                        streams[t].synchronize()
                        for k, (a, b) in enumerate(zip(out, serial[t])):
                            if not torch.equal(a, b):
                                errors.append((t, it, k))
                                return
        except Exception as e:                                       # noqa: BLE001 -- reported below, with the thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(N_THREADS)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    _C.profile_enable(False)
    _C.profile_collect()
    torch.cuda.synchronize()
    assert not errors, errors

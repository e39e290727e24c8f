"""The engine's plan is reproducible (odtk/fused.py: plan_state / load_plan / plan_hash, ODTK_CONV_PLAN,
ODTK_CONV_ROUTE; include/odtk_conv.h + include/odtk_hip.h: odtk_{conv,gemm}_plan_{export,import}).

The reference runs ONE deterministic PyTorch graph (odtk/model.py:125-165); the engine chooses routes, convolution instances
and hipBLASLt solutions by stopwatch, so the same checkpoint gave different bits on different boxes.  A plan names every
choice; a loaded plan is never re-measured.  CPU part: the state's format and round trip.  GPU part: two engines with the
same plan give bit-identical head tensors -- in one process, and across processes through a plan file."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(dtype=torch.bfloat16, classes=6, backbone='ResNet18FPN', device='cpu'):
    from odtk import fused
    from odtk.model import Model
    torch.manual_seed(0)
    model = Model(backbone, classes=classes).eval()
    model.initialize(None)
    model = model.to(device)
    return model, fused.FusedRetinaNet(model, dtype)


def test_route_keys_round_trip():
    from odtk import fused
    for key in [(8, 256, 100, 160), ('only', 2, 64, 32, 40)]:
        assert fused._str_key(fused._key_str(key)) == key
    assert fused._key_str((8, 256, 100, 160)) == 'act:8x256x100x160'
    assert fused._key_str(('only', 2, 64, 32, 40)) == 'only:2x64x32x40'


def test_plan_state_round_trips_and_pins_its_geometries(monkeypatch):
    from odtk import _C, fused
    taken = {}
    monkeypatch.setattr(_C, 'library_plans_export', lambda: 'conv 1 2 64 32 40 64 3 3 1 1 1 1 1 1 17 Some<Instance, 1>\ngemm 2560 64 256 1 1 0 7\n')
    monkeypatch.setattr(_C, 'library_plans_import', lambda text: taken.setdefault('text', text) and (1, 1))
    _, e1 = _engine()
    # what a plan pass would have recorded
    e1.layers[0][0].convs[0].route[(2, 64, 32, 40)] = (True, 10.0, 12.0)
    e1.layers[0][0].convs[0].learned['act'] = True
    e1.cls_head[-1].route[('only', 2, 256, 16, 20)] = (False, float('inf'), 9.0)
    e1.stem_s2d[((2, 3, 128, 160), torch.float32, True)] = (True, 30.0, 50.0)
    e1.stem_learned = True
    e1._planned.add(((2, 3, 128, 160), torch.device('cpu')))
    state = e1.plan_state()
    assert state['format'] == 'odtk-conv-plan-1' and state['dtype'] == 'bfloat16'
    assert state['layers']['layers.0.0.convs.0'] == {'act:2x64x32x40': 1}
    assert state['layers']['cls_head.%d' % (len(e1.cls_head) - 1)] == {'only:2x256x16x20': 0}
    assert state['stem'] == {'2x3x128x160:float32:1': 1} and state['geometries'] == [[2, 3, 128, 160]]
    assert len(state['libraries']) == 2
    text = json.dumps(state, sort_keys=True)                  # JSON-able, and the hash is a function of the content only
    _, e2 = _engine()
    e2.load_plan(json.loads(text))
    assert 'conv 1 2 64' in taken['text'] and 'gemm 2560' in taken['text']
    assert e2.layers[0][0].convs[0].route[(2, 64, 32, 40)][0] is True
    assert e2.cls_head[-1].route[('only', 2, 256, 16, 20)][0] is False
    assert e2.stem_s2d[((2, 3, 128, 160), torch.float32, True)][0] is True and e2.stem_learned
    assert (2, 3, 128, 160) in e2._loaded_geometries
    assert e2.plan_hash() == e1.plan_hash(state) == e1.plan_hash()
    # another route -> another hash
    e2.layers[0][0].convs[0].route[(2, 64, 32, 40)] = (False, None, None)
    assert e2.plan_hash() != e1.plan_hash()


def test_load_plan_refuses_what_it_cannot_apply():
    _, e = _engine()
    with pytest.raises(ValueError):
        e.load_plan({'format': 'something else'})
    with pytest.raises(ValueError):
        e.load_plan({'format': 'odtk-conv-plan-1', 'dtype': 'float16'})
    with pytest.raises(ValueError):
        e.load_plan({'format': 'odtk-conv-plan-1', 'dtype': 'bfloat16', 'layers': {'no.such.layer': {'act:1x8x8x8': 1}}})


def test_forced_route_modes_measure_nothing(monkeypatch):
    from odtk import fused
    _, e = _engine()
    conv = e.layers[0][0].convs[0]
    monkeypatch.setattr(fused._Conv, 'route_mode', 'library')
    assert conv._routed((2, 64, 32, 40), 'act', None, None, None) is True and conv.route[(2, 64, 32, 40)] == (True, None, None)
    monkeypatch.setattr(fused._Conv, 'route_mode', 'two_pass')
    assert conv._routed((2, 64, 16, 20), 'act', None, None, None) is False


def test_planning_flag_is_thread_local():
    import threading
    from odtk import fused
    fused._TLS.planning = True
    seen = []
    t = threading.Thread(target=lambda: seen.append(fused._planning()))
    t.start()
    t.join()
    fused._TLS.planning = False
    assert seen == [False]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_same_plan_same_bits_in_one_process():
    from odtk import _C, fused
    if not _C.conv_available():
        pytest.skip('libodtk_conv.so not built')
    # (the layers the plan leaves on MIOpen must themselves reproduce: find mode may pick kernels that do not, DESIGN section 5 --
    #  `cudnn.deterministic` restricts it to those that do, as tests/test_gpu_graph.py does)
    saved = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        _same_plan_same_bits()
    finally:
        torch.backends.cudnn.deterministic = saved


def _same_plan_same_bits():
    from odtk import fused
    model, e1 = _engine(device='cuda')
    x = torch.randn(2, 3, 256, 320, device='cuda')
    with torch.no_grad():
        e1.plan(x)                                               # the stopwatch decides (route mode auto)
        c1, b1 = e1.heads(x)
        again = e1.heads(x)
        for a, b in zip(c1 + b1, again[0] + again[1]):           # (the premise: one engine, one input, the same bits twice)
            assert torch.equal(a, b)
        state = json.loads(json.dumps(e1.plan_state()))
        assert state['layers'] and state['libraries']
        e2 = fused.FusedRetinaNet(model, torch.bfloat16)
        e2.load_plan(state)
        e2.plan(x)                                               # a no-op: the geometry is pinned
        c2, b2 = e2.heads(x)
    measured = [v for mod in e2.modules() if isinstance(mod, fused._Conv) for v in mod.route.values() if v[1] is not None]
    assert not measured, 'a loaded plan must not be re-measured'
    assert e2.plan_hash() == e1.plan_hash()
    for a, b in zip(c1 + b1, c2 + b2):
        assert torch.equal(a, b)
    # the opposite routes give other bits somewhere (the library epilogue adds its bias in bf16): the plan is what pins them
    flipped = json.loads(json.dumps(state))
    for routes in flipped['layers'].values():
        for k in routes:
            routes[k] = 1 - routes[k]
    e3 = fused.FusedRetinaNet(model, torch.bfloat16)
    e3.load_plan(flipped)
    with torch.no_grad():
        c3, b3 = e3.heads(x)
    assert e3.plan_hash() != e1.plan_hash()
    assert any(not torch.equal(a, b) for a, b in zip(c1 + b1, c3 + b3))


_CHILD = r'''
import hashlib, json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'retinanet-examples_amd'))
import torch
from odtk import fused
from odtk.model import Model
torch.backends.cudnn.deterministic = True   # the layers the plan leaves on MIOpen (every convolution with a skip input) must reproduce themselves
torch.manual_seed(0)
model = Model('ResNet18FPN', classes=6).eval()
model.initialize(None)
model = model.cuda()
e = fused.FusedRetinaNet(model, torch.bfloat16)
x = torch.randn(2, 3, 256, 320, generator=torch.Generator().manual_seed(3)).cuda()
with torch.no_grad():
    e.plan(x)
    cls, box = e.heads(x)
h = hashlib.sha256()
for t in cls + box:
    h.update(t.float().cpu().numpy().tobytes())
timed = sum(1 for line in e.plan_state()['libraries'] if line)
print(json.dumps({'digest': h.hexdigest(), 'plan_hash': e.plan_hash(), 'taken': e.libraries_taken, 'library_lines': timed}))
'''


@pytest.mark.gpu
def test_plan_file_replays_across_processes(tmp_path):
    """Every k x k layer the library supports goes through it (ODTK_CONV_ROUTE=library: no stopwatch between routes), first process
    writes the plan file, second one loads it: same plan hash, same head-tensor digest, and the libraries took their lines.
    The convolutions with a skip input stay on MIOpen + odtk_bias_act (a ResNet18 block's second convolution): the children run
    under cudnn.deterministic like the one-process test above -- MIOpen's find mode picks, per process and per box, kernels that
    need not reproduce their own bits (round 6, GPU calls 28-29: on one box every process, and every call
    inside a process, gave other head tensors from layers.1 on, with the stem and the plan hash equal: tools/plan_replay_probe.py)."""
    from odtk import _C
    if not _C.conv_available():
        pytest.skip('libodtk_conv.so not built')
    plan = str(tmp_path / 'plan.json')
    env = dict(os.environ, ODTK_CONV_PLAN=plan, ODTK_CONV_ROUTE='library')
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, '-c', _CHILD % {'root': ROOT}], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert os.path.isfile(plan)
    first, second = outs
    assert first['taken'] is None and second['taken'] is not None and second['taken'][0] > 0 and second['taken'][1] > 0
    assert first['plan_hash'] == second['plan_hash']
    assert first['digest'] == second['digest']

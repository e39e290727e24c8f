#!/usr/bin/env python
"""Does one plan file give the same head tensors in every process?  (tests/test_conv_plan.py::test_plan_file_replays_across_processes
failed once in GPU call 28 -- same plan hash, another digest -- after passing in every earlier call.)  Runs the test's child N times
on ONE plan file (the first run writes it) and prints per-tensor digests, digests of the engine's early activations and the GEMM
layer's pin misses, so that a difference names the layer it starts at.

    python tools/plan_replay_probe.py [runs]          (PROBE_DETERMINISTIC=1: the children set cudnn.deterministic, as the test does)
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'retinanet-examples_amd'))
import torch
from odtk import fused, _C
from odtk.model import Model
torch.backends.cudnn.deterministic = os.environ.get('PROBE_DETERMINISTIC', '0') == '1'
torch.manual_seed(0)
model = Model('ResNet18FPN', classes=6).eval()
model.initialize(None)
model = model.cuda()
e = fused.FusedRetinaNet(model, torch.bfloat16)
x = torch.randn(2, 3, 256, 320, generator=torch.Generator().manual_seed(3)).cuda()
acts = {}
def hook(name):
    def f(mod, inp, out):
        t = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(t) and name not in acts:
            acts[name] = hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:10]
    return f
for name, mod in e.named_modules():
    if name and name.count('.') <= 2:
        mod.register_forward_hook(hook(name))
with torch.no_grad():
    e.plan(x)
    acts.clear()
    cls, box = e.heads(x)
    first = [hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:10] for t in cls + box]
    cls, box = e.heads(x)
    second = [hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:10] for t in cls + box]
    packed = e._stem_packed(x) if hasattr(e, '_stem_packed') else None
sp = hashlib.sha256(packed.float().cpu().numpy().tobytes()).hexdigest()[:10] if torch.is_tensor(packed) else None
print(json.dumps({'heads': first, 'again': second, 'acts': acts, 'stem': sp, 'plan_hash': e.plan_hash(), 'taken': e.libraries_taken,
                  'pin_misses': _C.gemm_plan_pin_misses()}))
'''


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    plan = os.path.join(tempfile.mkdtemp(), 'plan.json')
    env = dict(os.environ, ODTK_CONV_PLAN=plan, ODTK_CONV_ROUTE='library')
    outs = []
    for i in range(runs):
        r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}], env=env, capture_output=True, text=True, timeout=900)
        if r.returncode:
            print('run %d failed: %s' % (i, r.stderr[-1500:]))
            return 1
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
        o = outs[-1]
        print('run %d: heads %s | same twice in the process: %s | stem %s | plan %s | taken %s | pin misses %s' %
              (i, ' '.join(o['heads']), o['heads'] == o['again'], o['stem'], o['plan_hash'][:10], o['taken'], o['pin_misses']), flush=True)
    ref = outs[0]
    bad = 0
    for i, o in enumerate(outs[1:], 1):
        if o['heads'] != ref['heads']:
            bad += 1
            names = [k for k in ref['acts'] if o['acts'].get(k) != ref['acts'][k]]
            print('run %d differs from run 0: head tensors %s; first differing activations: %s' %
                  (i, [j for j, (a, b) in enumerate(zip(o['heads'], ref['heads'])) if a != b], names[:12]))
    print('%d of %d runs differ from the first' % (bad, runs - 1))
    return 0


if __name__ == '__main__':
    sys.exit(main())

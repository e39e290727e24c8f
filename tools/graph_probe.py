import sys, os, time
sys.path[:0] = ['/root/repo', '/root/repo/retinanet-examples_amd']
import torch
torch.backends.cudnn.benchmark = True
from odtk.model import Model
from odtk.fused import FusedRetinaNet
P = lambda *a: print(*a, flush=True)
torch.manual_seed(0)
m = Model('ResNet50FPN'); m.initialize(None)
m = m.cuda().to(memory_format=torch.channels_last).eval()
x = torch.randn(8, 3, 800, 1280, device='cuda').contiguous(memory_format=torch.channels_last)
eng = FusedRetinaNet(m).cuda()
def step(): return eng(x)
for _ in range(8): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
P('eager fused: %.3f ms/step' % ((time.perf_counter() - t0) * 50))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
P('side-stream warmup done')
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
P('captured')
g.replay(); torch.cuda.synchronize()
P('first replay ok')
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
P('hipGraph replay fused: %.3f ms/step' % ((time.perf_counter() - t0) * 50))
snap = [o.clone() for o in out]
ref = step(); torch.cuda.synchronize()
P('graph output equals eager:', all(torch.equal(a, b) for a, b in zip(snap, ref)))

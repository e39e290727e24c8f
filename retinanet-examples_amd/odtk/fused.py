"""Inference-time fusion of the RetinaNet graph for MI355X.

The convolutions stay on PyTorch-ROCm / MIOpen (MFMA); what changes is everything BETWEEN them.
In the reference graph (torchvision blocks, FixedBatchNorm2d reference odtk/backbones/layers.py:5-16,
heads reference odtk/model.py:57-62) conv-bias, frozen batch-norm, residual add and ReLU are each a
full read+write pass over the activation, and under autocast every weight is re-cast every step.
Measured on MI355X those passes cost more than the convolutions between them.  Here

  * frozen BN is folded into the convolution: w' = w * gamma/sqrt(var+eps) (the scale), and the
    shift becomes a per-channel bias;
  * weights are stored once in the inference dtype (bf16), channels_last;
  * ONE hand-written HIP epilogue (`odtk_bias_act`, csrc/epilogue.hpp) applies bias (+ skip) (+ ReLU)
    in place on the convolution output;
  * 1x1 stride-1 convolutions (two of the three convs of every bottleneck, the FPN laterals) are a
    plain GEMM on the channels_last activation: they run as ONE hipBLASLt call with bias, skip and ReLU
    in the GEMM epilogue (`odtk_gemm_bias_act`, csrc/gemm_lt.hpp) -- no epilogue pass at all;
  * the post-processing reads the raw head tensors in place (odtk.box.detect(..., logits=True)); in
    `forward` even the bias of the heads' LAST convolutions is not applied by a pass of its own: it is
    handed to the post-processing kernels (`cls_bias` / `box_bias`), which add it in fp32 to the few values
    they actually look at -- the largest activation of the network (245 MB at bs 8) is written once by
    the convolution and read once by the prefilter.

`FusedRetinaNet(model)` is a drop-in for `model.eval()` inference: same outputs up to the rounding of
the folded weights (tests/test_gpu_fused_model.py).
"""
import hashlib
import json
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _C
from . import box as box_ops
from .backbones.resnet import BasicBlock, Bottleneck


_PLAN_LOCK = threading.Lock()     # one plan pass at a time per process (the A/B timings must not run beside each other)
_TLS = threading.local()          # .planning: THIS thread is inside a plan pass -> undecided k x k shapes are measured (A/B).  Thread-
                                  # local: a forward() on another thread never times anything in its hot path


def _planning():
    return getattr(_TLS, 'planning', False)


def _key_str(key):
    """Route key -> the string a plan file holds: tuple(x.shape) -> 'act:8x256x100x160', ('only',) + shape -> 'only:...'."""
    kind, shape = ('only', key[1:]) if key and key[0] == 'only' else ('act', key)
    return kind + ':' + 'x'.join(str(int(v)) for v in shape)


def _str_key(text):
    kind, shape = text.split(':')
    shape = tuple(int(v) for v in shape.split('x'))
    return (('only',) + shape) if kind == 'only' else shape


def fold_conv_bn(conv, bn=None):
    """(weight, bias) in float32 such that conv2d(x, weight) + bias == bn(conv(x)) in eval mode."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None:
        eps = getattr(bn, 'eps', 1e-5)
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
    return w, b


def stem_space_to_depth_weights(w):
    """[K, 3, 7, 7] weights of a 7x7 / stride-2 / pad-3 convolution -> [K, 16, 4, 4] weights of the 4x4 / stride-1 convolution with
    padding (2 before, 1 after) over the 2x2 space-to-depth image xs[n, (dy*2+dx)*3+c, y, x] = x[n, c, 2y+dy, 2x+dx] (channels 12..15
    zero) that computes the same outputs: tap (R, S) of sub-pixel (dy, dx) is tap (2R+dy-1, 2S+dx-1) of the 7x7 window, taps outside
    it are zero (include/odtk_hip.h: odtk_stem_pack; tests/test_stem_space_to_depth.py checks the identity in plain torch)."""
    w4 = torch.zeros(w.shape[0], 16, 4, 4, dtype=w.dtype, device=w.device)
    for dy in range(2):
        for dx in range(2):
            for r4 in range(4):
                for s4 in range(4):
                    r, s_ = 2 * r4 + dy - 1, 2 * s4 + dx - 1
                    if 0 <= r < 7 and 0 <= s_ < 7:
                        w4[:, (dy * 2 + dx) * 3:(dy * 2 + dx) * 3 + 3, r4, s4] = w[:, :, r, s_]
    return w4


def space_to_depth_pack(x):
    """Reference (plain torch) of odtk_stem_pack without the cast: [B, 3, H, W] -> [B, 16, H/2, W/2]."""
    b, _, h, w = x.shape
    xs = x.view(b, 3, h // 2, 2, w // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(b, 12, h // 2, w // 2)
    return torch.cat([xs, torch.zeros(b, 4, h // 2, w // 2, dtype=x.dtype, device=x.device)], 1)


class _Conv(nn.Module):
    """Convolution + bias (+ residual) (+ ReLU).  Three native routes, by kernel shape:
      * 1x1: ONE hipBLASLt GEMM with the whole epilogue (`odtk_gemm_bias_act`);
      * k x k without a skip input, 16-bit: ONE composable_kernel implicit-GEMM convolution with bias + ReLU in its own
        epilogue (`odtk_conv_bias_act`, csrc/conv_ck.cpp) -- where the engine's plan pass measured it faster than
      * the MIOpen convolution followed by the HIP epilogue pass `odtk_bias_act` (in place)."""

    use_conv_library = True   # set False to A/B the whole engine against round 4's graph
    # 'auto': every undecided k x k shape is measured both ways by the plan pass (a stopwatch: the routes -- and with them the
    # engine's bits, the library epilogue adds its bias in the activation dtype -- can differ from box to box); 'library' /
    # 'two_pass': no measurement, every layer the library supports / no layer goes through it (deterministic; ODTK_CONV_ROUTE).
    # A loaded plan (FusedRetinaNet.load_plan, ODTK_CONV_PLAN) pins the routes of its geometries whatever the mode.
    route_mode = os.environ.get('ODTK_CONV_ROUTE', 'auto')

    def __init__(self, conv, bn=None, relu=False, dtype=torch.bfloat16):
        super().__init__()
        w, b = fold_conv_bn(conv, bn)
        self.register_buffer('weight', w.to(dtype).contiguous(memory_format=torch.channels_last))
        self.register_buffer('bias', b.contiguous())
        # the convolution library's epilogue reads its bias in the activation dtype (the instance lists are built that way)
        self.register_buffer('bias_lp', b.to(dtype).contiguous())
        self.stride, self.padding, self.relu, self.groups = conv.stride, conv.padding, relu, conv.groups
        # pointwise: a GEMM over [N*H*W, Cin] with the whole epilogue fused (set False to A/B against MIOpen)
        # (a strided 1x1 convolution -- the downsample branch -- is the same GEMM on the subsampled pixels)
        self.pointwise = (tuple(conv.kernel_size) == (1, 1) and tuple(conv.padding) == (0, 0) and conv.groups == 1
                          and tuple(conv.dilation) == (1, 1))
        # (which channel counts the instances take is the library's business: it says "unsupported" and the A/B records inf)
        self.library_ok = (not self.pointwise and conv.groups == 1 and tuple(conv.dilation) == (1, 1)
                           and dtype in (torch.bfloat16, torch.float16) and w.shape[1] % 8 == 0)
        self.strided_ok = (self.pointwise and tuple(conv.stride) != (1, 1) and dtype in (torch.bfloat16, torch.float16)
                           and w.shape[1] % 8 == 0)
        self.zero_bias = None
        self.learned = {}                                               # kind ('act' | 'only') -> the last measured decision: unplanned geometries follow it
        self.route = {}                                                 # input shape -> (use the library, us library, us miopen + epilogue)

    def conv_only(self, x, out=None):
        """The convolution without its epilogue (the caller owns the bias: the heads' last convolutions, whose bias the
        post-processing kernels add).  Routed like `forward`: the library's instance list is a second find space for the same
        contraction (a zero bias, no clamp), taken where the plan pass measured it faster than MIOpen's pick."""
        if (self.library_ok and _Conv.use_conv_library and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last)
                and _C.conv_available()):
            key = ('only',) + tuple(x.shape)
            if self.zero_bias is None:
                self.zero_bias = torch.zeros_like(self.bias_lp)
            if self._routed(key, 'only', x, self._miopen_only, self._library_only):
                try:
                    return self._library_only(x, out)
                except RuntimeError:
                    self.route[key] = (False, float('inf'), 0.0)
        y = self._miopen_only(x)
        return y if out is None else out.copy_(y)                       # (MIOpen owns its output: an arena then costs a copy)

    def _miopen_only(self, x):
        y = F.conv2d(x, self.weight, None, self.stride, self.padding, groups=self.groups)
        return y if y.is_contiguous(memory_format=torch.channels_last) else y.contiguous(memory_format=torch.channels_last)

    def _library_only(self, x, out=None):
        return _C.conv_bias_act(x, self.weight, self.zero_bias, self.stride, self.padding, False, out=out)

    def _two_pass(self, x, residual=None):
        y = F.conv2d(x, self.weight, None, self.stride, self.padding, groups=self.groups)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        return _C.bias_act_(y, self.bias, residual, self.relu)

    def _copy_then_gemm(self, x, residual=None):
        x = x[:, :, ::self.stride[0], ::self.stride[1]].contiguous(memory_format=torch.channels_last)
        return _C.gemm_bias_act(x, self.weight, self.bias, residual, self.relu)

    def _one_pass(self, x):
        return _C.conv_bias_act(x, self.weight, self.bias_lp, self.stride, self.padding, self.relu)

    def _measure(self, x, two=None, one=None):
        """A/B of the two k x k routes on this input (plan pass only): median of 5 after a warm-up, events on the current stream."""
        two, one = two or self._two_pass, one or self._one_pass

        def timed(fn):
            fn(x)
            times = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(x)
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3)
            return sorted(times)[2]
        t_two = timed(two)
        try:
            t_one = timed(one)
        except RuntimeError:                                            # no instance of the library takes the problem
            t_one = float('inf')
        return (t_one < t_two, t_one, t_two)

    def forward(self, x, residual=None):
        if self.pointwise and _C.gemm_available():
            if tuple(self.stride) != (1, 1):
                # a strided 1x1 convolution (the downsample branches): the GEMM wants the subsampled pixels as a dense matrix -- a
                # copy pass of its own (29 us each at bs 8) -- the convolution library reads them in place; routed by the plan pass
                if (residual is None and self.strided_ok and _Conv.use_conv_library and x.is_cuda
                        and x.is_contiguous(memory_format=torch.channels_last) and _C.conv_available()):
                    if self._routed(tuple(x.shape), 'act', x, self._copy_then_gemm, self._one_pass):
                        try:
                            return self._one_pass(x)
                        except RuntimeError:
                            self.route[tuple(x.shape)] = (False, float('inf'), 0.0)
                return self._copy_then_gemm(x, residual)
            return _C.gemm_bias_act(x, self.weight, self.bias, residual, self.relu)
        if (residual is None and self.library_ok and _Conv.use_conv_library and x.is_cuda
                and x.is_contiguous(memory_format=torch.channels_last) and _C.conv_available()):
            if self._routed(tuple(x.shape), 'act', x, self._two_pass, self._one_pass):
                try:
                    return self._one_pass(x)
                except RuntimeError:                                    # an unplanned geometry no instance takes: remember, fall through
                    self.route[tuple(x.shape)] = (False, float('inf'), 0.0)
        return self._two_pass(x, residual)

    def _routed(self, key, kind, x, two, one):
        """Does this input go through the convolution library?  Measured during a plan pass; a geometry no plan pass has seen (the
        engine plans the first few only: a data set's batches come in dozens of padded sizes) follows the layer's last measured
        decision."""
        route = self.route.get(key)
        if route is None and _Conv.route_mode != 'auto':
            route = self.route[key] = (_Conv.route_mode == 'library', None, None)
        if route is None and _planning() and not torch.cuda.is_current_stream_capturing():
            route = self.route[key] = self._measure(x, two, one)
            self.learned[kind] = route[0]
        if route is None:
            return self.learned.get(kind, False)
        return route[0]

    def conv_then_pool(self, x):
        """conv -> bias -> ReLU -> maxpool 3x3/s2 with the epilogue folded into the pooling pass."""
        y = F.conv2d(x, self.weight, None, self.stride, self.padding, groups=self.groups)
        if y.dtype in (torch.bfloat16, torch.float16) and y.shape[1] % 8 == 0:
            return _C.bias_act_maxpool(y.contiguous(memory_format=torch.channels_last), self.bias, self.relu)
        return F.max_pool2d(_C.bias_act_(y.contiguous(memory_format=torch.channels_last), self.bias, None, self.relu), 3, 2, 1)


class _Block(nn.Module):
    def __init__(self, block, dtype):
        super().__init__()
        if isinstance(block, Bottleneck):
            self.convs = nn.ModuleList([_Conv(block.conv1, block.bn1, True, dtype), _Conv(block.conv2, block.bn2, True, dtype)])
            self.last = _Conv(block.conv3, block.bn3, True, dtype)        # ReLU after the skip add
        elif isinstance(block, BasicBlock):
            self.convs = nn.ModuleList([_Conv(block.conv1, block.bn1, True, dtype)])
            self.last = _Conv(block.conv2, block.bn2, True, dtype)
        else:
            raise TypeError('unsupported block %r' % type(block))
        self.down = None if block.downsample is None else _Conv(block.downsample[0], block.downsample[1], False, dtype)

    def forward(self, x):
        skip = x if self.down is None else self.down(x)
        y = x
        for c in self.convs:
            y = c(y)
        return self.last(y, skip)                                       # relu(conv + shift + skip), one pass


class FusedRetinaNet(nn.Module):
    @staticmethod
    def supports(model):
        """One ResNet-FPN backbone made of Bottleneck / BasicBlock stages taking C3..C5."""
        if len(model.backbones) != 1:
            return False
        fpn = next(iter(model.backbones.values()))
        net = getattr(fpn, 'features', None)
        if net is None or not all(hasattr(net, n) for n in ('conv1', 'bn1', 'layer1', 'layer2', 'layer3', 'layer4')):
            return False
        if list(getattr(net, 'outputs', [])) != [3, 4, 5]:
            return False
        return all(isinstance(b, (Bottleneck, BasicBlock)) for layer in (net.layer1, net.layer2, net.layer3, net.layer4)
                   for b in layer)

    def __init__(self, model, dtype=torch.bfloat16):
        super().__init__()
        if not self.supports(model):
            raise ValueError('FusedRetinaNet supports a single ResNet-FPN backbone')
        fpn = next(iter(model.backbones.values()))
        net = fpn.features
        self.dtype = dtype
        self.model = [model]                                            # not a submodule: shares anchors / config
        self.stem = _Conv(net.conv1, net.bn1, True, dtype)
        # the stem as a 4x4 / stride-1 convolution over the 2x2 space-to-depth image of the input (16-byte channel vectors instead
        # of 3-element ones; include/odtk_hip.h: odtk_stem_pack): same products, re-indexed weights
        self.stem_s2d = None
        c1 = net.conv1
        if (tuple(c1.kernel_size), tuple(c1.stride), tuple(c1.padding), c1.in_channels, c1.groups) == ((7, 7), (2, 2), (3, 3), 3, 1) \
                and dtype in (torch.bfloat16, torch.float16):
            w = self.stem.weight.float()
            w4 = stem_space_to_depth_weights(w)
            self.register_buffer('stem_w4', w4.to(dtype).contiguous(memory_format=torch.channels_last))
            self.register_buffer('stem_zero_bias', torch.zeros(w.shape[0], dtype=dtype, device=w.device))
            self.stem_s2d = {}                                          # input (shape, dtype, layout) -> (use it, us s2d, us direct)
        self.stem_learned = False                                       # the last measured stem decision (unplanned geometries follow it)
        self.layers = nn.ModuleList([nn.ModuleList([_Block(b, dtype) for b in layer])
                                     for layer in (net.layer1, net.layer2, net.layer3, net.layer4)])
        self.outputs = list(net.outputs)
        self.lateral = nn.ModuleList([_Conv(fpn.lateral3, None, False, dtype), _Conv(fpn.lateral4, None, False, dtype),
                                      _Conv(fpn.lateral5, None, False, dtype)])
        self.smooth = nn.ModuleList([_Conv(fpn.smooth3, None, False, dtype), _Conv(fpn.smooth4, None, False, dtype),
                                     _Conv(fpn.smooth5, None, False, dtype)])
        self.pyramid6 = _Conv(fpn.pyramid6, None, False, dtype)
        self.pyramid7 = _Conv(fpn.pyramid7, None, False, dtype)

        def head(seq):
            convs = [m for m in seq if isinstance(m, nn.Conv2d)]
            return nn.ModuleList([_Conv(c, None, i + 1 < len(convs), dtype) for i, c in enumerate(convs)])

        self.cls_head = head(model.cls_head)
        self.box_head = head(model.box_head)
        self.level_streams = True                                       # small pyramid levels on side HIP streams
        self._streams = None
        self.tower_plan = 0
        self._planned = set()                                           # input geometries whose k x k convolutions were routed (plan pass)
        self.cls_arena = bool(os.environ.get('ODTK_CLS_ARENA'))           # experiment: cls head tensors in one 2 MiB-aligned buffer (_cls_arena)
        self._arenas = {}
        self._loaded_geometries = set()                                 # input shapes routed by a loaded plan (load_plan): never measured
        self._plan_file_seen = False                                    # ODTK_CONV_PLAN was looked at
        self.libraries_taken = None                                     # (gemm, conv) lines the libraries took from the last loaded plan
        self.max_plans = 4                                              # ... at most so many: a data set's batches come in dozens of padded sizes
        self._graphs = {}                                               # input geometry + bias state -> (hipGraph, static input, outputs, tables kept alive)
        self._thresholds = {}                                           # score threshold -> the prefilter's table for cls_head[-1].bias
        self.max_graphs = 8

    def _stem_direct(self, x):
        x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
        return self.stem.conv_then_pool(x)                               # conv1 -> (bias + ReLU + maxpool, one pass)

    def _stem_packed(self, x):
        xs = _C.stem_pack(x, self.dtype)                                 # cast + space-to-depth, one pass over the input
        y = _C.conv_bias_act(xs, self.stem_w4, self.stem_zero_bias, 1, ((2, 1), (2, 1)), False)
        return _C.bias_act_maxpool(y, self.stem.bias, self.stem.relu)    # the stem's bias + ReLU ride in the pooling pass

    def _stem(self, x):
        """conv1 -> bn1 -> ReLU -> maxpool on the RAW input (any float dtype, NCHW or channels_last)."""
        if (self.stem_s2d is not None and _Conv.use_conv_library and x.is_cuda and x.dim() == 4 and x.shape[1] == 3
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.dtype in _C._DTYPES and _C.conv_available()
                and (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last))):
            key = (tuple(x.shape), x.dtype, x.is_contiguous())
            route = self.stem_s2d.get(key)
            if route is None and _Conv.route_mode != 'auto':
                route = self.stem_s2d[key] = (_Conv.route_mode == 'library', None, None)
            if route is None and _planning() and not torch.cuda.is_current_stream_capturing():
                route = self.stem_s2d[key] = self.stem._measure(x, self._stem_direct, self._stem_packed)
                self.stem_learned = route[0]
            if route[0] if route is not None else self.stem_learned:
                try:
                    return self._stem_packed(x)
                except RuntimeError:
                    self.stem_s2d[key] = (False, float('inf'), 0.0)
        return self._stem_direct(x)

    def features(self, x):
        """x: the network input as the caller has it (the stem owns the cast to the engine's dtype and layout)."""
        x = self._stem(x)
        feats = []
        for level, layer in enumerate(self.layers, start=2):
            for block in layer:
                x = block(x)
            if level in self.outputs:
                feats.append(x)
        c3, c4, c5 = feats
        p5 = self.lateral[2](c5)
        p4 = self.lateral[1](c4, self._upsample(p5))                    # lateral + upsampled, one pass
        p3 = self.lateral[0](c3, self._upsample(p4))
        p6 = self.pyramid6(c5)
        p7 = self.pyramid7(F.relu(p6))
        return [self.smooth[0](p3), self.smooth[1](p4), self.smooth[2](p5), p6, p7]

    @staticmethod
    def _upsample(t):
        """Nearest 2x (reference fpn.py:45-61).  On the GPU one HIP stream kernel (csrc/epilogue.hpp: torch's upsample kernel +
        the channels_last copy behind it cost 125 us per step, 4x what the bytes need)."""
        if (t.is_cuda and t.dtype in _C._DTYPES and t.is_contiguous(memory_format=torch.channels_last)
                and (t.shape[1] * t.element_size()) % 16 == 0):
            return _C.upsample2x(t)
        return F.interpolate(t, scale_factor=2).contiguous(memory_format=torch.channels_last)

    @staticmethod
    def _run(seq, t):
        for c in seq:
            t = c(t)
        return t

    def _towers(self, feats, last_bias, level_streams=None):
        """Both head towers on every pyramid level.  The levels are independent and the small ones (P5-P7:
        8000 / 2080 / 560 pixels at bs 8) cannot fill 256 CUs on their own, so they run on side HIP streams
        next to P3's convolutions: [P3] on the caller's stream, [P4] and [P5, P6, P7] on two others."""
        arena = self._cls_arena(feats) if (self.cls_arena and not last_bias and feats[0].is_cuda and not _planning()) else None

        def level(t, i):
            if last_bias:
                return self._run(self.cls_head, t), self._run(self.box_head, t)
            return (self.cls_head[-1].conv_only(self._run(self.cls_head[:-1], t), None if arena is None else arena[i]),
                    self.box_head[-1].conv_only(self._run(self.box_head[:-1], t)))

        if level_streams is None:
            level_streams = self.level_streams
        if not level_streams or not feats[0].is_cuda or len(feats) < 3:
            out = [level(t, i) for i, t in enumerate(feats)]
            return [o[0] for o in out], [o[1] for o in out]
        main = torch.cuda.current_stream(feats[0].device)
        if self._streams is None or self._streams[0].device != feats[0].device:
            self._streams = [torch.cuda.Stream(feats[0].device) for _ in range(3)]
        n = len(feats)
        # (levels on `main`, levels of each side stream).  One A/B run (bench.py --tower-plan): 7.52 / 7.37 / 7.34 ms
        # per step for plans 0 / 1 / 2 in that order on one box -- within its drift; 0 is the tested default
        mine, groups = {0: ([0], [[1], list(range(2, n))]),
                        1: (list(range(2, n)) + [1], [[0]]),
                        2: ([0], [[1], [2], list(range(3, n))])}[self.tower_plan]
        ready = torch.cuda.Event()
        ready.record(main)
        out = [None] * len(feats)
        done = []
        for stream, group in zip(self._streams, groups):
            stream.wait_event(ready)                                    # the pyramid is complete
            with torch.cuda.stream(stream):
                for i in group:
                    out[i] = level(feats[i], i)
                    for t in out[i]:
                        t.record_stream(main)                           # consumed by the post-processing on `main`
                e = torch.cuda.Event()
                e.record(stream)
                done.append(e)
        for i in mine:
            out[i] = level(feats[i], i)
        for e in done:
            main.wait_event(e)
        return [o[0] for o in out], [o[1] for o in out]

    def _cls_arena(self, feats):
        """Experiment (ODTK_CLS_ARENA=1): the five cls head tensors -- the 245 MB the prefilter streams -- in ONE
        engine-owned buffer, every level on a 2 MiB boundary, allocated once per geometry and kept (the same virtual pages every
        step, aligned to the largest page-table fragment).  Views in channels_last layout; the convolution library writes into
        them (`conv_only(x, out)`), MIOpen-routed levels are copied.  Measured: profiles/r06_prefilter_tlb.txt."""
        channels = self.cls_head[-1].weight.shape[0]
        key = tuple((t.shape[0], t.shape[2], t.shape[3]) for t in feats) + (feats[0].device,)
        hit = self._arenas.get(key)
        if hit is None:
            step = 2 << 20
            elt = torch.empty(0, dtype=self.dtype).element_size()
            sizes = [t.shape[0] * t.shape[2] * t.shape[3] * channels * elt for t in feats]
            total = sum((n + step - 1) // step * step for n in sizes) + step
            buf = torch.empty(total, dtype=torch.uint8, device=feats[0].device)
            off = (-buf.data_ptr()) % step
            views = []
            for t, n in zip(feats, sizes):
                v = buf[off:off + n].view(self.dtype).view(t.shape[0], t.shape[2], t.shape[3], channels).permute(0, 3, 1, 2)
                views.append(v)
                off += (n + step - 1) // step * step
            self._arenas.clear()
            hit = self._arenas[key] = (buf, views)
        return hit[1]

    # The engine owns its dtypes (weights are stored in `self.dtype`, every op runs in it): an enclosing
    # torch.autocast region -- which is how Model.forward picks the engine's dtype -- must not re-cast anything.
    def heads(self, x):
        with torch.autocast(x.device.type, enabled=False):
            return self._towers(self.features(x), True)

    def heads_without_last_bias(self, x):
        """Head tensors as the last convolutions wrote them (bias NOT added) + the two bias vectors."""
        with torch.autocast(x.device.type, enabled=False):
            cls, box = self._towers(self.features(x), False)
        return cls, box, self.cls_head[-1].bias, self.box_head[-1].bias

    @torch.no_grad()
    def replay(self, x):
        """`forward(x)` as ONE hipGraph (torch.cuda.CUDAGraph = hipGraph on ROCm): captured once per input geometry, then
        only the input copy and the graph launch remain on the host -- the ~300 launches of a step are what a batch-1 call
        costs.  Every buffer the captured kernels touch (activations, the binding's scratch, outputs) is allocated during the
        capture and therefore owned by the graph; nothing cached outside it is referenced (odtk/_C.py:_workspace).  The
        returned tensors are copies (the graph's own output buffers are overwritten by the next replay)."""
        m = self.model[0]                                                # (post-processing parameters are baked into the launches)
        bias = self.cls_head[-1].bias                                    # its state is baked in too: the prefilter's threshold table
        key = (tuple(x.shape), x.dtype, x.device, x.is_contiguous(memory_format=torch.channels_last),
               m.threshold, m.top_n, m.nms, m.detections, self.level_streams, self.tower_plan, bias.data_ptr(), bias._version)
        entry = self._graphs.get(key)
        if entry is None:
            static_x = torch.empty_like(x)
            static_x.copy_(x)
            # warm-up on a side stream (MIOpen find results, hipBLASLt plans, lazily created streams) -- capture must find
            # nothing left to initialise
            side = torch.cuda.Stream(x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.forward(static_x)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward(static_x)
            while len(self._graphs) >= self.max_graphs:                 # a graph pins its activations: keep a handful of geometries
                self._graphs.pop(next(iter(self._graphs)))              # (oldest first: dicts keep insertion order)
            # the capture baked the address of the threshold table its warm-up passes made into the prefilter's launch: the
            # graph entry co-owns that table, so a later `_thresholds.clear()` (new threshold, bias updated in place) cannot
            # free memory a cached graph still reads
            entry = self._graphs[key] = (graph, static_x, out, tuple(self._thresholds.values()))
        graph, static_x, out = entry[:3]
        static_x.copy_(x)
        graph.replay()
        return tuple(o.clone() for o in out)

    def plan(self, x):
        """Routes every k x k convolution for this input geometry: one pass over the graph on the caller's stream (no side
        streams: nothing runs beside a convolution while it is timed) in which each such convolution is run both ways -- the
        convolution library's fused epilogue vs MIOpen + `odtk_bias_act` -- and keeps the faster (`_Conv.route`).
        ODTK_CONV_PLAN=<file>: the file's plan is loaded before anything is measured (its geometries are then never timed,
        neither here nor inside the two kernel libraries); what a plan pass adds is written back to it."""
        path = os.environ.get('ODTK_CONV_PLAN')
        if path and not self._plan_file_seen:
            self._plan_file_seen = True
            if os.path.isfile(path):
                with open(path) as f:
                    self.load_plan(json.load(f))
        key = (tuple(x.shape), x.device)
        if key in self._planned or not x.is_cuda or not _C.conv_available() or not _Conv.use_conv_library:
            return
        if tuple(x.shape) in self._loaded_geometries:
            self._planned.add(key)                                      # routed by a loaded plan: nothing to measure
            return
        if len(self._planned) >= self.max_plans:
            return                                                      # later geometries follow the layers' last measured decisions
        if torch.cuda.is_current_stream_capturing():
            return                                                      # (replay() warms up eagerly first: planned by then)
        with _PLAN_LOCK:
            _TLS.planning = True
            try:
                with torch.autocast(x.device.type, enabled=False):
                    self._towers(self.features(x), False, level_streams=False)
            finally:
                _TLS.planning = False
            self._planned.add(key)
        if path:
            self.save_plan(path)

    # ---- reproducible plans: the engine's bits depend on three stopwatches -- this file's A/B per layer, the
    # convolution library's instance per problem, hipBLASLt's solution per problem -- none of which picks the same on every box
    # (68 vs 70 of 75 layers routed to the library on two boxes of round 5), and the two routes of a layer round differently (the
    # library's epilogue takes its bias in the activation dtype, `odtk_bias_act` adds it in fp32).  A plan names all three.
    def plan_state(self):
        """Everything the plan passes of this engine (and the libraries under it, process-wide) decided so far, as a JSON-able
        dict: layer -> input geometry -> route, the stem's form, and the libraries' own lines."""
        layers, learned = {}, {}
        for name, mod in self.named_modules():
            if isinstance(mod, _Conv):
                if mod.route:
                    layers[name] = {_key_str(k): int(bool(v[0])) for k, v in sorted(mod.route.items(), key=lambda kv: _key_str(kv[0]))}
                if mod.learned:
                    learned[name] = {k: bool(v) for k, v in sorted(mod.learned.items())}
        stem = {}
        for (shape, dtype, contiguous), v in (self.stem_s2d or {}).items():
            stem['%s:%s:%d' % ('x'.join(str(int(d)) for d in shape), str(dtype).replace('torch.', ''), int(contiguous))] = int(bool(v[0]))
        geometries = sorted({tuple(k[0]) for k in self._planned} | set(self._loaded_geometries))
        return {'format': 'odtk-conv-plan-1', 'dtype': str(self.dtype).replace('torch.', ''), 'geometries': [list(g) for g in geometries],
                'layers': layers, 'learned': learned, 'stem': dict(sorted(stem.items())), 'stem_learned': bool(self.stem_learned),
                'libraries': sorted(line for line in _C.library_plans_export().splitlines() if line)}

    def plan_hash(self, state=None):
        """16 hex digits naming the active plan: routes + library lines (bench.py prints it next to `conv_epilogue`)."""
        state = state or self.plan_state()
        text = json.dumps({k: state[k] for k in ('dtype', 'layers', 'stem', 'libraries')}, sort_keys=True)
        return hashlib.sha256(text.encode()).hexdigest()[:16]

    def load_plan(self, state):
        """Pins what `plan_state()` of an engine of the same architecture and dtype recorded: its geometries are never measured
        again, neither by `plan()` nor -- for problems this process has not planned yet -- inside the two libraries."""
        if state.get('format') != 'odtk-conv-plan-1':
            raise ValueError('not an odtk conv plan: format %r' % (state.get('format'),))
        if state.get('dtype') != str(self.dtype).replace('torch.', ''):
            raise ValueError('the plan was made for %s, this engine runs %s' % (state.get('dtype'), self.dtype))
        mods = {name: mod for name, mod in self.named_modules() if isinstance(mod, _Conv)}
        unknown = sorted(set(state.get('layers', {})) - set(mods))
        if unknown:
            raise ValueError('the plan names layers this engine does not have: %s' % ', '.join(unknown[:4]))
        for name, routes in state.get('layers', {}).items():
            for text, use in routes.items():
                mods[name].route[_str_key(text)] = (bool(use), None, None)
        for name, kinds in state.get('learned', {}).items():
            if name in mods:
                mods[name].learned.update({k: bool(v) for k, v in kinds.items()})
        if self.stem_s2d is not None:
            for text, use in state.get('stem', {}).items():
                shape, dtype, contiguous = text.split(':')
                self.stem_s2d[(tuple(int(v) for v in shape.split('x')), getattr(torch, dtype), bool(int(contiguous)))] = (bool(use), None, None)
            self.stem_learned = bool(state.get('stem_learned', self.stem_learned))
        self._loaded_geometries.update(tuple(g) for g in state.get('geometries', []))
        self.libraries_taken = _C.library_plans_import('\n'.join(state.get('libraries', [])) + '\n')
        return self.libraries_taken

    def save_plan(self, path):
        tmp = '%s.%d.tmp' % (path, os.getpid())
        with open(tmp, 'w') as f:
            json.dump(self.plan_state(), f, indent=1, sort_keys=True)
        os.replace(tmp, path)

    def conv_routes(self):
        """{layer name: {input shape: (library?, us library, us two-pass)}} of the plan passes so far (measurement records)."""
        routes = {name: dict(mod.route) for name, mod in self.named_modules() if isinstance(mod, _Conv) and mod.route}
        if self.stem_s2d:
            routes['stem (cast + conv1 + pool: space-to-depth form vs direct)'] = {k[0]: v for k, v in self.stem_s2d.items()}
        return routes

    @torch.no_grad()
    def forward(self, x):
        m = self.model[0]
        self.plan(x)
        fold = self.dtype in (torch.bfloat16, torch.float16) and self.cls_head[-1].bias.numel() % 8 == 0
        if fold:
            cls_heads, box_heads, cls_bias, box_bias = self.heads_without_last_bias(x)
        else:
            (cls_heads, box_heads), cls_bias, box_bias = self.heads(x), None, None
        strides = [x.shape[-1] // c.shape[-1] for c in cls_heads]
        for s in strides:
            m.level_anchors(s)
        table = None
        if fold and cls_bias.is_cuda and not os.environ.get('ODTK_NO_THRESHOLD_TABLE'):
            # the prefilter's per-channel threshold table: made once per (threshold, state of the bias vector); a capture
            # finds the one its eager warm-up passes made (without: the kernels derive the thresholds themselves)
            key = (float(m.threshold), cls_bias.data_ptr(), cls_bias._version)
            table = self._thresholds.get(key)
            if table is None and not torch.cuda.is_current_stream_capturing():
                self._thresholds.clear()
                table = self._thresholds[key] = _C.prefilter_thresholds(cls_bias, self.dtype, m.threshold)
        if os.environ.get('ODTK_CHECK_FINITE'):
            # debug: the library epilogue (AddClamp) turns a NaN accumulator into 0 / -inf where the two-pass route and the
            # reference's graph hand the NaN on (include/odtk_conv.h) -- a diverged checkpoint must not pass as "no detections"
            for name, heads in (('cls', cls_heads), ('box', box_heads)):
                for i, t in enumerate(heads):
                    if not bool(torch.isfinite(t).all()):
                        raise FloatingPointError('FusedRetinaNet: non-finite values in the %s head tensor of level %d' % (name, i))
        return box_ops.detect(cls_heads, box_heads, strides, m.anchors, m.threshold, m.top_n, m.nms, m.detections,
                              m.rotated_bbox, logits=True, cls_bias=cls_bias, box_bias=box_bias, cls_thresholds=table)

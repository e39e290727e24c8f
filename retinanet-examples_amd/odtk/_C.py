"""odtk._C -- the operator boundary of the reference (csrc/extensions.cpp:184-201), re-homed on
the MI355X C ABI (include/odtk_hip.h, libodtk_hip.so) through ctypes.

Exposes the reference's three free functions with the SAME positional signatures and return
conventions, so `from ._C import decode, nms, iou` in a box.py keeps working:

    decode(cls_head, box_head, anchors, scale, score_thresh, top_n, rotated=False) -> [scores, boxes, classes]
    nms(scores, boxes, classes, nms_thresh, detections_per_im, rotated=False)     -> [scores, boxes, classes]
    iou(boxes, anchors)                                                           -> [Tensor[num_anchors, num_boxes]]
    Engine                                                                        -> placeholder (TensorRT dropped)

plus the MI355X-first batched forms (`decode_levels`, `detect`) that cover all pyramid levels of
the whole batch in one enqueue.

There is NO CPU fallback: tensors must live on the GPU and the shared library must be present
(build it with `python __graft_entry__.py` or `make -C retinanet-examples_amd/csrc`); anything
else raises.  Work is only ENQUEUED on torch's current HIP stream -- no host synchronisation
(the reference blocks the host once per image per level, csrc/cuda/decode.cu:103).
"""
import ctypes
import threading
import os

import torch

_LIB_PATH = os.environ.get('ODTK_HIP_LIBRARY') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libodtk_hip.so')

OK, ERR_INVALID, ERR_WORKSPACE, ERR_HIP, ERR_UNSUPPORTED = 0, -1, -2, -3, -4
F32, BF16, F16 = 0, 1, 2
FLAG_ROTATED, FLAG_LOGITS, FLAG_ROTATED_NMS_FIXED_ANGLE = 1, 2, 4
MAX_LEVELS, MAX_ANCHORS, MAX_TOP_N, MAX_NMS_COUNT, MAX_NMS_COUNT_SCRATCH = 6, 32, 16384, 7680, 1 << 22

_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)
_fp = ctypes.POINTER(ctypes.c_float)
_sz = ctypes.c_size_t


class LossLevel(ctypes.Structure):
    """odtk_loss_level_t"""
    _fields_ = [('cls', ctypes.c_void_p), ('box', ctypes.c_void_p), ('depth', ctypes.c_void_p), ('box_target', ctypes.c_void_p),
                ('dcls', ctypes.c_void_p), ('dbox', ctypes.c_void_p), ('height', ctypes.c_int32), ('width', ctypes.c_int32),
                ('channels_last', ctypes.c_int32), ('pad_', ctypes.c_int32)]


class SnapLevel(ctypes.Structure):
    """odtk_snap_level_t"""
    _fields_ = [('anchors', ctypes.POINTER(ctypes.c_float)), ('cls_target', ctypes.c_void_p), ('box_target', ctypes.c_void_p),
                ('depth', ctypes.c_void_p), ('height', ctypes.c_int32), ('width', ctypes.c_int32), ('stride', ctypes.c_int32),
                ('pad_', ctypes.c_int32)]


class SnapRotLevel(ctypes.Structure):
    """odtk_snap_rot_level_t"""
    _fields_ = [('anchors_axis', ctypes.c_void_p), ('anchors_quads', ctypes.c_void_p), ('cls_target', ctypes.c_void_p),
                ('box_target', ctypes.c_void_p), ('depth', ctypes.c_void_p), ('height', ctypes.c_int32), ('width', ctypes.c_int32),
                ('stride', ctypes.c_int32), ('pad_', ctypes.c_int32)]


class Level(ctypes.Structure):
    """odtk_level_t"""
    _fields_ = [('cls', _vp), ('box', _vp), ('height', ctypes.c_int32), ('width', ctypes.c_int32),
                ('stride', ctypes.c_int32), ('channels_last', ctypes.c_int32), ('anchors', _fp),
                ('cls_bias', _vp), ('box_bias', _vp), ('cls_thresholds', _vp)]


_SIGNATURES = {
    'odtk_version': (ctypes.c_char_p, []),
    'odtk_abi_struct_size': (ctypes.c_int, [ctypes.c_int]),
    'odtk_last_hip_error': (ctypes.c_char_p, []),
    'odtk_decode': (ctypes.c_int, [ctypes.c_int, _vpp, _vpp, _sz, _sz, _sz, _sz, _sz, _fp, _sz, ctypes.c_float,
                                   ctypes.c_int, _vp, _sz, _vp]),
    'odtk_decode_rotate': (ctypes.c_int, [ctypes.c_int, _vpp, _vpp, _sz, _sz, _sz, _sz, _sz, _fp, _sz,
                                          ctypes.c_float, ctypes.c_int, _vp, _sz, _vp]),
    'odtk_nms': (ctypes.c_int, [ctypes.c_int, _vpp, _vpp, _sz, ctypes.c_int, ctypes.c_float, _vp, _sz, _vp]),
    'odtk_nms_rotate': (ctypes.c_int, [ctypes.c_int, _vpp, _vpp, _sz, ctypes.c_int, ctypes.c_float, _vp, _sz, _vp]),
    'odtk_iou': (ctypes.c_int, [_vpp, _vpp, ctypes.c_int, ctypes.c_int, _vp]),
    'odtk_decode_levels': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(Level), ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_float, ctypes.c_int,
                                          _vpp, ctypes.c_int, _vp, _sz, _vp]),
    'odtk_nms_ex': (ctypes.c_int, [ctypes.c_int, _vpp, _vpp, ctypes.c_int, _sz, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_uint32, _vp, _sz, _vp]),
    'odtk_detect': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(Level), ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_int, _vpp, _vp, _sz, _vp]),
    'odtk_snap_to_anchors': (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                            _vp, _vp, _vp, _vp]),
    'odtk_snap_to_anchors_levels': (ctypes.c_int, [ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(SnapLevel),
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _vp]),
    'odtk_retina_loss_forward': (ctypes.c_int, [_vp, _vp, _vp, _vp] + [ctypes.c_int] * 8 + [ctypes.c_float] * 3 + [_vp, _vp]),
    'odtk_retina_loss_backward': (ctypes.c_int, [_vp, _vp, _vp, _vp] + [ctypes.c_int] * 8 + [ctypes.c_float] * 3 +
                                  [_vp, _vp, _vp, _vp, _vp]),
    'odtk_retina_loss_levels_forward': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(LossLevel)] + [ctypes.c_int] * 5 +
                                        [ctypes.c_float] * 3 + [_vp, _vp]),
    'odtk_retina_loss_levels_forward_ws': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(LossLevel)] + [ctypes.c_int] * 5 +
                                           [ctypes.c_float] * 3 + [_vp, _vp, _sz, _vp]),
    'odtk_retina_loss_levels_backward': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(LossLevel)] + [ctypes.c_int] * 5 +
                                         [ctypes.c_float] * 3 + [_vp, _vp, _vp]),
    'odtk_bias_act': (ctypes.c_int, [_vp, _vp, _vp, _sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'odtk_profile_enable': (ctypes.c_int, [ctypes.c_int]),
    'odtk_debug_set_trace': (ctypes.c_int, [_vp]),
    'odtk_debug_loss_tuning': (ctypes.c_int, [ctypes.c_int] * 6),
    'odtk_debug_loss_layout': (ctypes.c_int, [ctypes.c_int] * 5),
    'odtk_debug_loss_form': (ctypes.c_int, [ctypes.c_int]),
    'odtk_bias_act_maxpool': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'odtk_snap_to_anchors_rotated_levels': (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                                           ctypes.POINTER(SnapRotLevel), ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                           ctypes.c_float, _vp]),
    'odtk_prefilter_thresholds': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, _vp]),
    'odtk_upsample_nearest2x': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p]),
    'odtk_stem_pack': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'odtk_nms_sorted_runs': (ctypes.c_int, [ctypes.c_int, _vpp, _vpp, ctypes.c_int, _sz, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_uint32, _vp, _sz, _vp]),
    'odtk_gemm_init': (ctypes.c_int, [ctypes.c_char_p]),
    'odtk_gemm_plan_export': (ctypes.c_size_t, [ctypes.c_char_p, ctypes.c_size_t]),
    'odtk_gemm_plan_import': (ctypes.c_int, [ctypes.c_char_p]),
    'odtk_gemm_plan_pin_misses': (ctypes.c_int, []),
    'odtk_gemm_bias_act': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'odtk_profile_collect': (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
}

KERNEL_NAMES = ('prefilter_scan_kernel', 'select_decode_kernel', 'nms_kernel', 'iou_pairs_kernel', 'bias_act_kernel',
                'snap_to_anchors_kernel', 'gemm_bias_act', 'retina_loss_kernel', 'select_hist_kernel', 'select_filter_kernel',
                'nms_first_round_kernel', 'rotated_sup_matrix_kernel', 'bias_act_maxpool_kernel', 'upsample_nearest2x_kernel',
                'stem_pack_kernel', 'loss_reduce_kernel')
# HBM bytes the epilogue entry points move, per kernel name, while `traffic_count` is on (bench.py's epilogue_roofline: the
# algorithmic bytes of every call -- each element read once and written once, + the skip input -- divided by the kernel time)
traffic_count = False
traffic_bytes = {}


def _count(name, nbytes):
    if traffic_count:
        traffic_bytes[name] = traffic_bytes.get(name, 0) + int(nbytes)


_lib = None


def library():
    """The loaded libodtk_hip.so (raises -- loudly -- if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise ImportError('odtk._C: %s is missing -- build the HIP library first '
                              '(python __graft_entry__.py, or make -C retinanet-examples_amd/csrc). '
                              'There is no CPU fallback.' % _LIB_PATH)
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        # ABI guard: the ctypes mirrors below must have the layout the library was compiled with (include/odtk_hip.h)
        for which, mirror in enumerate((Level, SnapLevel, SnapRotLevel, LossLevel)):
            if lib.odtk_abi_struct_size(which) != ctypes.sizeof(mirror):
                raise ImportError('odtk._C: %s was built from another revision of include/odtk_hip.h (sizeof struct %d: library %d, '
                                  'binding %d) -- rebuild it (make -C retinanet-examples_amd/csrc)'
                                  % (_LIB_PATH, which, lib.odtk_abi_struct_size(which), ctypes.sizeof(mirror)))
        _lib = lib
        # Debug knob (tools/loss_probe.py): ODTK_LOSS_TUNING="fwd32:threads,blocks_per_cu,unroll,box_blocks;bwd16:...;ws32:..."
        # (fwd = forward with atomics, bwd, ws = forward through a workspace) overrides the built-in launch shapes
        for part in filter(None, os.environ.get('ODTK_LOSS_TUNING', '').split(';')):
            side, _, vals = part.partition(':')
            side = side.strip()
            _check(lib.odtk_debug_loss_tuning({'fwd': 0, 'bwd': 1, 'ws': 2}[side[:-2]], int(side.endswith('32')),
                                              *(int(v) for v in vals.split(','))), 'ODTK_LOSS_TUNING')
        # ODTK_LOSS_FORM=0|1: arithmetic form of the gamma = 2 classification walk (include/odtk_hip.h: odtk_debug_loss_form).
        # Said out loud: a process-wide numerical switch must not be silent.  (The ablation forms 2..7 -- wrong sums on purpose --
        # exist only in a -DODTK_LOSS_ABLATIONS build; the shipped library refuses them here.)
        if os.environ.get('ODTK_LOSS_FORM', '') != '':
            import warnings
            warnings.warn('odtk._C: ODTK_LOSS_FORM=%s overrides the loss kernels\' arithmetic form for this process' % os.environ['ODTK_LOSS_FORM'])
            _check(lib.odtk_debug_loss_form(int(os.environ['ODTK_LOSS_FORM'])), 'ODTK_LOSS_FORM')
    return _lib


def exported_symbols():
    return [n for n in _SIGNATURES]


_ERR = {ERR_INVALID: 'invalid argument', ERR_WORKSPACE: 'Workspace is too small!',
        ERR_HIP: 'HIP error', ERR_UNSUPPORTED: 'unsupported dtype/layout'}


def _check(rc, what):
    if rc < 0:
        detail = library().odtk_last_hip_error().decode() if rc == ERR_HIP else ''
        raise RuntimeError('%s failed: %s%s' % (what, _ERR.get(rc, 'error %d' % rc), (' (%s)' % detail) if detail else ''))
    return rc


def _check_input(t, name):
    # csrc/extensions.cpp:42-44  CHECK_CUDA / CHECK_CONTIGUOUS -> RuntimeError
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a CUDA tensor' % name)
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)
    if t.dtype != torch.float32:
        raise RuntimeError('%s must be float32' % name)


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


_workspaces = {}
_workspaces_lock = threading.Lock()      # several host threads may enqueue on their own streams at once (include/odtk_hip.h)


def _workspace(device, nbytes):
    """Scratch owned by the binding, cached per (device, stream): never shared between streams.
    While the stream is being captured into a hipGraph nothing is cached and nothing cached is used: the buffer of a
    captured call is allocated for that call and belongs to the graph (torch serves allocations made during a capture
    from the graph's private pool and keeps that pool alive exactly as long as the graph), so a graph never reads
    scratch that an eager call -- or another graph -- may grow, replace or overwrite, and the cache never holds
    memory of a graph that no longer exists."""
    stream = torch.cuda.current_stream(device)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device), stream.cuda_stream
    key = (device.index, stream.cuda_stream)
    with _workspaces_lock:
        ws = _workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            _workspaces[key] = ws
    return ws, stream.cuda_stream


def _anchor_array(anchors):
    flat = [float(v) for v in anchors]
    return (ctypes.c_float * len(flat))(*flat), len(flat)


def decode(cls_head, box_head, anchors, scale, score_thresh, top_n, rotated=False):
    """csrc/extensions.cpp:69-115.  `anchors` is the flat python list the reference passes
    (anchors.view(-1).tolist(), odtk/box.py:263-264)."""
    _check_input(cls_head, 'cls_head')
    _check_input(box_head, 'box_head')
    lib = library()
    nb = 6 if rotated else 4
    batch, channels, height, width = cls_head.shape
    arr, n = _anchor_array(anchors)
    num_anchors = n // 4
    if num_anchors == 0 or n % 4 or channels % num_anchors:
        raise RuntimeError('decode: %d anchor values do not describe the %d channels of cls_head' % (n, channels))
    num_classes = channels // num_anchors
    if box_head.shape != (batch, num_anchors * nb, height, width):
        raise RuntimeError('box_head shape %s does not match cls_head %s' % (tuple(box_head.shape), tuple(cls_head.shape)))
    fn = lib.odtk_decode_rotate if rotated else lib.odtk_decode
    with torch.cuda.device(cls_head.device):
        scores = torch.empty((batch, top_n), dtype=torch.float32, device=cls_head.device)
        boxes = torch.empty((batch, top_n, nb), dtype=torch.float32, device=cls_head.device)
        classes = torch.empty((batch, top_n), dtype=torch.float32, device=cls_head.device)
        size = _check(fn(batch, None, None, height, width, int(scale), num_anchors, num_classes, arr, n,
                         float(score_thresh), int(top_n), None, 0, None), 'decode (workspace query)')
        ws, stream = _workspace(cls_head.device, size)
        _check(fn(batch, _ptrs([cls_head, box_head]), _ptrs([scores, boxes, classes]), height, width, int(scale),
                  num_anchors, num_classes, arr, n, float(score_thresh), int(top_n), ws.data_ptr(), ws.numel(),
                  stream), 'decode')
    return [scores, boxes, classes]


def nms(scores, boxes, classes, nms_thresh, detections_per_im, rotated=False, return_indices=False):
    """csrc/extensions.cpp:117-158."""
    _check_input(scores, 'scores')
    _check_input(boxes, 'boxes')
    _check_input(classes, 'classes')
    lib = library()
    nb = 6 if rotated else 4
    batch, count = scores.shape
    if boxes.shape != (batch, count, nb) or classes.shape != (batch, count):
        raise RuntimeError('nms: inconsistent shapes')
    dev = scores.device
    with torch.cuda.device(dev):
        out = [torch.empty((batch, detections_per_im), dtype=torch.float32, device=dev),
               torch.empty((batch, detections_per_im, nb), dtype=torch.float32, device=dev),
               torch.empty((batch, detections_per_im), dtype=torch.float32, device=dev)]
        if return_indices:
            out.append(torch.empty((batch, detections_per_im), dtype=torch.int32, device=dev))
        flags = FLAG_ROTATED if rotated else 0
        size = _check(lib.odtk_nms_ex(batch, None, None, len(out), count, int(detections_per_im), float(nms_thresh),
                                      flags, None, 0, None), 'nms (workspace query)')
        ws, stream = _workspace(dev, size)
        _check(lib.odtk_nms_ex(batch, _ptrs([scores, boxes, classes]), _ptrs(out), len(out), count,
                               int(detections_per_im), float(nms_thresh), flags, ws.data_ptr(), ws.numel(), stream),
               'nms')
    return out


def nms_sorted_runs(scores, boxes, classes, run_len, nms_thresh, detections_per_im, rotated=False):
    """`nms` for candidates that are n_runs = count / run_len runs, each already in NMS order with its non-positive scores at
    the end -- the concatenation of per-level `decode` outputs (include/odtk_hip.h: odtk_nms_sorted_runs).  Same result as
    `nms`, without its compaction / selection / sort rounds."""
    _check_input(scores, 'scores')
    _check_input(boxes, 'boxes')
    _check_input(classes, 'classes')
    lib = library()
    nb = 6 if rotated else 4
    batch, count = scores.shape
    if boxes.shape != (batch, count, nb) or classes.shape != (batch, count) or count % int(run_len):
        raise RuntimeError('nms_sorted_runs: inconsistent shapes')
    dev = scores.device
    with torch.cuda.device(dev):
        run_valid = (scores.view(batch, count // int(run_len), int(run_len)) > 0).sum(2).to(torch.int32).contiguous()
        out = [torch.empty((batch, detections_per_im), dtype=torch.float32, device=dev),
               torch.empty((batch, detections_per_im, nb), dtype=torch.float32, device=dev),
               torch.empty((batch, detections_per_im), dtype=torch.float32, device=dev)]
        flags = FLAG_ROTATED if rotated else 0
        size = _check(lib.odtk_nms_sorted_runs(batch, None, None, 3, count, int(run_len), None, int(detections_per_im),
                                               float(nms_thresh), flags, None, 0, None), 'nms_sorted_runs (workspace query)')
        ws, stream = _workspace(dev, size)
        _check(lib.odtk_nms_sorted_runs(batch, _ptrs([scores, boxes, classes]), _ptrs(out), 3, count, int(run_len),
                                        run_valid.data_ptr(), int(detections_per_im), float(nms_thresh), flags, ws.data_ptr(),
                                        ws.numel(), stream), 'nms_sorted_runs')
    return out


def iou(boxes, anchors):
    """csrc/extensions.cpp:47-67: flat [N*8] box corners, flat [M*8] anchor corners -> [Tensor[M, N]]."""
    _check_input(boxes, 'boxes')
    _check_input(anchors, 'anchors')
    lib = library()
    num_boxes = boxes.numel() // 8
    num_anchors = anchors.numel() // 8
    with torch.cuda.device(boxes.device):
        out = torch.empty((num_anchors, num_boxes), dtype=torch.float32, device=boxes.device)
        stream = torch.cuda.current_stream(boxes.device).cuda_stream
        _check(lib.odtk_iou(_ptrs([boxes, anchors]), _ptrs([out]), num_boxes, num_anchors, stream), 'iou')
    return [out]


_DTYPES = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


def _pair_layout(c, b, cname, bname, what, prefer_channels_last=False):
    """Common memory format of a (cls_head, box_head) pair: 0 NCHW, 1 channels_last.  A tensor with ONE channel (a one-anchor,
    one-class head) or ONE pixel (a 1 x 1 level) is contiguous in both readings -- torch says so for either format -- and follows
    its partner instead of being pinned to NCHW (found by tools/decode_fuzz_long.py: A = C = 1 with channels_last heads was
    refused as 'mixed formats').  prefer_channels_last: take that reading when both tensors allow both (the bias fold wants it)."""
    def readings(t, name):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError('%s must be a CUDA tensor' % name)
        if t.dim() != 4:
            raise RuntimeError('%s must be 4-d [B, C, H, W]' % name)
        n, l = t.is_contiguous(), t.is_contiguous(memory_format=torch.channels_last)
        if not (n or l):
            raise RuntimeError('%s must be contiguous (NCHW or channels_last)' % name)
        return n, l
    cn, cl = readings(c, cname)
    bn, bl = readings(b, bname)
    if prefer_channels_last and cl and bl:
        return 1
    if cn and bn:
        return 0
    if cl and bl:
        return 1
    raise RuntimeError('%s%s and %s must share one memory format' % (what, cname, bname))


def prefilter_thresholds(cls_bias, dtype, score_thresh):
    """The prefilter's per-channel threshold table for `cls_bias` (float32 CUDA vector [A*C], A*C % 8 == 0), a 16-bit head
    dtype and a score threshold: made ONCE (an engine: when it folds its weights) and handed to decode_levels / detect as
    `cls_thresholds`, so that the ~3750 workgroups of the prefilter load 2880 ready bytes instead of deriving them from the
    bias before their first load.  A table made for other parameters is detected and not used."""
    if not cls_bias.is_cuda or cls_bias.dtype != torch.float32 or not cls_bias.is_contiguous() or cls_bias.numel() % 8:
        raise RuntimeError('prefilter_thresholds: cls_bias must be a contiguous float32 CUDA vector whose length is a multiple of 8')
    if dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError('prefilter_thresholds: 16-bit head dtypes only')
    table = torch.empty(cls_bias.numel() + 8, dtype=torch.float32, device=cls_bias.device)
    with torch.cuda.device(cls_bias.device):
        stream = torch.cuda.current_stream(cls_bias.device).cuda_stream
        _check(library().odtk_prefilter_thresholds(cls_bias.data_ptr(), cls_bias.numel(), _DTYPES[dtype], float(score_thresh),
                                                   table.data_ptr(), stream), 'prefilter_thresholds')
    return table


def _levels(cls_heads, box_heads, anchors_list, strides, nb, cls_bias=None, box_bias=None, cls_thresholds=None):
    """Fill the odtk_level_t table.  Head tensors are taken AS THE CONVOLUTION WROTE THEM: float32 /
    bfloat16 / float16, NCHW or channels_last -- no .float(), no .contiguous() (reference
    model.py:160, box.py:263 make both copies)."""
    n = len(cls_heads)
    if not (n == len(box_heads) == len(anchors_list) == len(strides)) or n == 0 or n > MAX_LEVELS:
        raise RuntimeError('decode_levels: need 1..%d levels with matching lists' % MAX_LEVELS)
    batch = cls_heads[0].shape[0]
    dtype = cls_heads[0].dtype
    if dtype not in _DTYPES:
        raise RuntimeError('decode_levels: unsupported dtype %s' % dtype)
    arr = (Level * n)()
    keep = []
    num_anchors = None
    for i, (c, b, a, s) in enumerate(zip(cls_heads, box_heads, anchors_list, strides)):
        # (a 1 x 1 level or a one-channel head is NCHW- and NHWC-contiguous at once: it follows its partner; the bias fold wants
        #  the channels_last reading where both allow it)
        lay = _pair_layout(c, b, 'cls_head[%d]' % i, 'box_head[%d]' % i, '', prefer_channels_last=cls_bias is not None)
        if c.dtype != dtype or b.dtype != dtype:
            raise RuntimeError('decode_levels: all head tensors must share one dtype')
        flat = a.reshape(-1).tolist() if isinstance(a, torch.Tensor) else list(a)
        carr, ln = _anchor_array(flat)
        keep.append(carr)
        if num_anchors is None:
            num_anchors = ln // 4
        if ln != 4 * num_anchors or c.shape[0] != batch or b.shape != (batch, num_anchors * nb, c.shape[2], c.shape[3]):
            raise RuntimeError('decode_levels: inconsistent level %d' % i)
        arr[i].cls = c.data_ptr()
        arr[i].box = b.data_ptr()
        arr[i].height, arr[i].width = c.shape[2], c.shape[3]
        arr[i].stride = int(s)
        arr[i].channels_last = lay
        arr[i].anchors = ctypes.cast(carr, _fp)
        for name, bias, width in (('cls_bias', cls_bias, c.shape[1]), ('box_bias', box_bias, b.shape[1]),
                                  ('cls_thresholds', cls_thresholds if cls_bias is not None else None, c.shape[1] + 8)):
            if bias is None:
                continue
            t = bias[i] if isinstance(bias, (list, tuple)) else bias      # one tensor shared by all levels, or a list
            if t is None:
                continue
            if not t.is_cuda or t.dtype != torch.float32 or t.numel() != width or not t.is_contiguous():
                raise RuntimeError('decode_levels: %s must be a contiguous float32 CUDA vector of length %d' % (name, width))
            keep.append(t)
            setattr(arr[i], name, t.data_ptr())
    num_classes = cls_heads[0].shape[1] // num_anchors
    return arr, keep, batch, num_anchors, num_classes, _DTYPES[dtype]


def decode_levels(cls_heads, box_heads, anchors_list, strides, score_thresh, top_n, rotated=False,
                  return_indices=False, logits=False, cls_bias=None, box_bias=None, cls_thresholds=None):
    """All levels x whole batch in one enqueue; returns tensors already in the layout of
    `torch.cat(per_level, 1)` (odtk/model.py:164): [B, L*top_n], [B, L*top_n, nb], [B, L*top_n].
    logits=True: cls_heads hold raw logits and the sigmoid is fused into the prefilter.
    cls_bias / box_bias: float32 CUDA vectors [A*C] / [A*nb] (or per-level lists) -- the bias of the heads'
    last convolutions, added inside the kernels (cls_bias: logits, 16-bit channels_last heads only)."""
    lib = library()
    nb = 6 if rotated else 4
    arr, keep, batch, num_anchors, num_classes, dtype = _levels(cls_heads, box_heads, anchors_list, strides, nb,
                                                                cls_bias, box_bias, cls_thresholds)
    dev = cls_heads[0].device
    n = len(cls_heads)
    with torch.cuda.device(dev):
        out = [torch.empty((batch, n * top_n), dtype=torch.float32, device=dev),
               torch.empty((batch, n * top_n, nb), dtype=torch.float32, device=dev),
               torch.empty((batch, n * top_n), dtype=torch.float32, device=dev)]
        if return_indices:
            out.append(torch.empty((batch, n * top_n), dtype=torch.int32, device=dev))
        flags = (FLAG_ROTATED if rotated else 0) | (FLAG_LOGITS if logits else 0)
        size = _check(lib.odtk_decode_levels(batch, n, arr, num_anchors, num_classes, dtype, flags, float(score_thresh),
                                             int(top_n), None, 0, None, 0, None), 'decode_levels (workspace query)')
        ws, stream = _workspace(dev, size)
        _check(lib.odtk_decode_levels(batch, n, arr, num_anchors, num_classes, dtype, flags, float(score_thresh),
                                      int(top_n), _ptrs(out), len(out), ws.data_ptr(), ws.numel(), stream),
               'decode_levels')
    return out


def detect(cls_heads, box_heads, anchors_list, strides, score_thresh, top_n, nms_thresh, detections_per_im,
           rotated=False, logits=False, cls_bias=None, box_bias=None, cls_thresholds=None):
    """decode_levels + nms back to back (the whole of odtk/model.py:140-165): three launches -- prefilter, select + decode, nms
    -- (rotated boxes: five to seven), no host synchronisation, one workspace."""
    lib = library()
    nb = 6 if rotated else 4
    arr, keep, batch, num_anchors, num_classes, dtype = _levels(cls_heads, box_heads, anchors_list, strides, nb,
                                                                cls_bias, box_bias, cls_thresholds)
    dev = cls_heads[0].device
    n = len(cls_heads)
    with torch.cuda.device(dev):
        out = [torch.empty((batch, detections_per_im), dtype=torch.float32, device=dev),
               torch.empty((batch, detections_per_im, nb), dtype=torch.float32, device=dev),
               torch.empty((batch, detections_per_im), dtype=torch.float32, device=dev)]
        flags = (FLAG_ROTATED if rotated else 0) | (FLAG_LOGITS if logits else 0)
        args = (batch, n, arr, num_anchors, num_classes, dtype, flags, float(score_thresh), int(top_n),
                float(nms_thresh), int(detections_per_im))
        size = _check(lib.odtk_detect(*args, None, None, 0, None), 'detect (workspace query)')
        ws, stream = _workspace(dev, size)
        _check(lib.odtk_detect(*args, _ptrs(out), ws.data_ptr(), ws.numel(), stream), 'detect')
    return out


def snap_to_anchors(targets, anchors, num_classes, height, width, stride, iou_background, iou_foreground,
                    want_cls_target=True):
    """Fused target assignment of one pyramid level for the whole batch (reference box.py:134-189 per
    image).  targets: float32 CUDA [B, N, 5] (x, y, w, h, class; class < 0 = padding row).
    -> cls_target [B, A, C, H, W] (None with want_cls_target=False: the fused loss derives it from depth),
    box_target [B, A, 4, H, W], depth [B, A, 1, H, W]."""
    _check_input(targets, 'targets')
    if targets.dim() != 3 or targets.shape[2] != 5:
        raise RuntimeError('targets must be [B, N, 5]')
    arr, n = _anchor_array(anchors.reshape(-1).tolist() if isinstance(anchors, torch.Tensor) else anchors)
    a = n // 4
    b, n_max = targets.shape[0], targets.shape[1]
    dev = targets.device
    with torch.cuda.device(dev):
        cls = torch.empty((b, a, num_classes, height, width), dtype=torch.float32, device=dev) if want_cls_target else None
        box_t = torch.empty((b, a, 4, height, width), dtype=torch.float32, device=dev)
        depth = torch.empty((b, a, 1, height, width), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(library().odtk_snap_to_anchors(b, targets.data_ptr(), n_max, arr, a, int(num_classes), int(height),
                                              int(width), int(stride), float(iou_background), float(iou_foreground),
                                              cls.data_ptr() if cls is not None else None, box_t.data_ptr(),
                                              depth.data_ptr(), stream),
               'snap_to_anchors')
    return cls, box_t, depth


def snap_to_anchors_rotated_levels(gt_axis, gt_quads, gt_class, anchors_list, num_classes, sizes, strides, iou_background,
                                   iou_foreground, want_cls_target=True):
    """Rotated target assignment (reference box.py:192-252) for every pyramid level of the batch in ONE launch.
    gt_axis [B, N, 6], gt_quads [B, N, 8], gt_class [B, N] (< 0: padding) float32 CUDA -- the reference's `rotate_boxes` output;
    anchors_list: per level (axis [A, 4], quads [A, 8]) float32 CUDA tensors; sizes: per level (H, W).
    -> lists (cls_targets or Nones, box_targets [B, A, 6, H, W], depths)."""
    for t, name, last in ((gt_axis, 'gt_axis', 6), (gt_quads, 'gt_quads', 8)):
        _check_input(t, name)
        if t.dim() != 3 or t.shape[2] != last:
            raise RuntimeError('%s must be [B, N, %d]' % (name, last))
    _check_input(gt_class, 'gt_class')
    n = len(anchors_list)
    if not (n == len(sizes) == len(strides)) or n == 0 or n > MAX_LEVELS:
        raise RuntimeError('snap_to_anchors_rotated_levels: need 1..%d levels with matching lists' % MAX_LEVELS)
    b, n_max = gt_axis.shape[0], gt_axis.shape[1]
    if gt_quads.shape[:2] != (b, n_max) or tuple(gt_class.shape) != (b, n_max):
        raise RuntimeError('snap_to_anchors_rotated_levels: gt_axis / gt_quads / gt_class disagree')
    dev = gt_axis.device
    arr = (SnapRotLevel * n)()
    cls_t, box_t, depth_t = [], [], []
    a = None
    with torch.cuda.device(dev):
        for i, ((axis, quads), (h, w), s) in enumerate(zip(anchors_list, sizes, strides)):
            _check_input(axis, 'anchors (axis form)')
            _check_input(quads, 'anchors (quads)')
            if a is None:
                a = axis.shape[0]
            if tuple(axis.shape) != (a, 4) or tuple(quads.shape) != (a, 8):
                raise RuntimeError('snap_to_anchors_rotated_levels: every level needs [A, 4] and [A, 8] anchor tables')
            cls_t.append(torch.empty((b, a, num_classes, h, w), dtype=torch.float32, device=dev) if want_cls_target else None)
            box_t.append(torch.empty((b, a, 6, h, w), dtype=torch.float32, device=dev))
            depth_t.append(torch.empty((b, a, 1, h, w), dtype=torch.float32, device=dev))
            arr[i].anchors_axis, arr[i].anchors_quads = axis.data_ptr(), quads.data_ptr()
            arr[i].cls_target = cls_t[-1].data_ptr() if want_cls_target else None
            arr[i].box_target, arr[i].depth = box_t[-1].data_ptr(), depth_t[-1].data_ptr()
            arr[i].height, arr[i].width, arr[i].stride = int(h), int(w), int(s)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(library().odtk_snap_to_anchors_rotated_levels(b, gt_axis.data_ptr(), gt_quads.data_ptr(), gt_class.data_ptr(), n_max, n,
                                                             arr, a, int(num_classes), float(iou_background), float(iou_foreground),
                                                             stream), 'snap_to_anchors_rotated_levels')
    return cls_t, box_t, depth_t


def snap_to_anchors_levels(targets, anchors_list, num_classes, sizes, strides, iou_background, iou_foreground,
                           want_cls_target=True):
    """`snap_to_anchors` for every pyramid level in ONE launch.  anchors_list: per level [A, 4]; sizes: per level (H, W).
    -> lists (cls_targets or Nones, box_targets, depths)."""
    _check_input(targets, 'targets')
    if targets.dim() != 3 or targets.shape[2] != 5:
        raise RuntimeError('targets must be [B, N, 5]')
    n = len(anchors_list)
    if not (n == len(sizes) == len(strides)) or n == 0 or n > MAX_LEVELS:
        raise RuntimeError('snap_to_anchors_levels: need 1..%d levels with matching lists' % MAX_LEVELS)
    b, n_max = targets.shape[0], targets.shape[1]
    dev = targets.device
    arr = (SnapLevel * n)()
    keep, cls_t, box_t, depth_t = [], [], [], []
    a = None
    with torch.cuda.device(dev):
        for i, (anchors, (h, w), s) in enumerate(zip(anchors_list, sizes, strides)):
            carr, ln = _anchor_array(anchors.reshape(-1).tolist() if isinstance(anchors, torch.Tensor) else anchors)
            if a is None:
                a = ln // 4
            if ln != 4 * a:
                raise RuntimeError('snap_to_anchors_levels: every level needs the same number of anchors')
            keep.append(carr)
            cls_t.append(torch.empty((b, a, num_classes, h, w), dtype=torch.float32, device=dev) if want_cls_target else None)
            box_t.append(torch.empty((b, a, 4, h, w), dtype=torch.float32, device=dev))
            depth_t.append(torch.empty((b, a, 1, h, w), dtype=torch.float32, device=dev))
            arr[i].anchors = ctypes.cast(carr, _fp)
            arr[i].cls_target = cls_t[-1].data_ptr() if want_cls_target else None
            arr[i].box_target, arr[i].depth = box_t[-1].data_ptr(), depth_t[-1].data_ptr()
            arr[i].height, arr[i].width, arr[i].stride = int(h), int(w), int(s)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(library().odtk_snap_to_anchors_levels(b, targets.data_ptr(), n_max, n, arr, a, int(num_classes),
                                                     float(iou_background), float(iou_foreground), stream),
               'snap_to_anchors_levels')
    return cls_t, box_t, depth_t


def _loss_geometry(cls_head, box_head, depth, box_target):
    """Shared validation of the fused loss entry points -> (B, A, C, H, W, nb, dtype enum, channels_last)."""
    if not (cls_head.is_cuda and box_head.is_cuda and depth.is_cuda and box_target.is_cuda):
        raise RuntimeError('retina_loss: tensors must be on the GPU')
    if cls_head.dim() != 4 or box_head.dim() != 4 or depth.dim() != 5 or box_target.dim() != 5:
        raise RuntimeError('retina_loss: cls/box must be [B, ch, H, W], depth [B, A, 1, H, W], box_target [B, A, nb, H, W]')
    lay = _pair_layout(cls_head, box_head, 'cls_head', 'box_head', 'retina_loss: ')
    b, ch, h, w = cls_head.shape
    a, nb = box_target.shape[1], box_target.shape[2]
    if cls_head.dtype not in _DTYPES or box_head.dtype != cls_head.dtype:
        raise RuntimeError('retina_loss: heads must share one dtype out of float32 / bfloat16 / float16')
    if ch % a or depth.shape != (b, a, 1, h, w) or box_target.shape != (b, a, nb, h, w) or box_head.shape != (b, a * nb, h, w):
        raise RuntimeError('retina_loss: inconsistent shapes')
    if depth.dtype != torch.float32 or box_target.dtype != torch.float32 or not depth.is_contiguous() or not box_target.is_contiguous():
        raise RuntimeError('retina_loss: depth and box_target must be contiguous float32')
    return b, a, ch // a, h, w, nb, _DTYPES[cls_head.dtype], lay


def _loss_levels(cls_heads, box_heads, depths, box_targets, grads=None):
    n = len(cls_heads)
    if not (n == len(box_heads) == len(depths) == len(box_targets)) or n == 0 or n > MAX_LEVELS:
        raise RuntimeError('retina_loss_levels: need 1..%d levels with matching lists' % MAX_LEVELS)
    arr = (LossLevel * n)()
    geo = None
    for i in range(n):
        b, a, c, h, w, nb, dtype, lay = _loss_geometry(cls_heads[i], box_heads[i], depths[i], box_targets[i])
        if geo is None:
            geo = (b, a, c, nb, dtype)
        elif geo != (b, a, c, nb, dtype):
            raise RuntimeError('retina_loss_levels: every level must share batch, anchors, classes, box parameters and dtype')
        arr[i].cls, arr[i].box = cls_heads[i].data_ptr(), box_heads[i].data_ptr()
        arr[i].depth, arr[i].box_target = depths[i].data_ptr(), box_targets[i].data_ptr()
        arr[i].height, arr[i].width, arr[i].channels_last = h, w, lay
        if grads is not None:
            arr[i].dcls, arr[i].dbox = grads[0][i].data_ptr(), grads[1][i].data_ptr()
    return arr, n, geo


def retina_loss_levels_forward(cls_heads, box_heads, depths, box_targets, alpha, gamma, beta, reproducible=False):
    """All pyramid levels in ONE launch -> float64 CUDA tensor [L, 3] = per level (cls_sum, box_sum, #foreground).
    reproducible=True (what odtk/loss.py asks for since round 6): the workgroups' sums go through a workspace and a second, tiny
    launch adds them up in a fixed order (odtk_retina_loss_levels_forward_ws): no atomics, the same bits on every run, and the walk
    is 8 us shorter than with three double atomics at the end of every workgroup (fp32, 2 images of 800 x 1280: 26.2 + 4.2 us for the
    two launches against 34.6; profiles/r06_loss_layout_probe.txt)."""
    arr, n, (b, a, c, nb, dtype) = _loss_levels(cls_heads, box_heads, depths, box_targets)
    dev = cls_heads[0].device
    lib = library()
    with torch.cuda.device(dev):
        sums = torch.empty((n, 3), dtype=torch.float64, device=dev)
        if not reproducible:
            stream = torch.cuda.current_stream(dev).cuda_stream
            _check(lib.odtk_retina_loss_levels_forward(n, arr, b, a, c, nb, dtype, float(alpha), float(gamma), float(beta),
                                                       sums.data_ptr(), stream), 'retina_loss_levels_forward')
            return sums
        need = _check(lib.odtk_retina_loss_levels_forward_ws(n, arr, b, a, c, nb, dtype, float(alpha), float(gamma), float(beta),
                                                             None, None, 0, None), 'retina_loss_levels_forward (workspace query)')
        ws, stream = _workspace(dev, need)
        _check(lib.odtk_retina_loss_levels_forward_ws(n, arr, b, a, c, nb, dtype, float(alpha), float(gamma), float(beta),
                                                      sums.data_ptr(), ws.data_ptr(), ws.numel(), stream), 'retina_loss_levels_forward')
    return sums


def retina_loss_levels_backward(cls_heads, box_heads, depths, box_targets, alpha, gamma, beta, grad_cls_sums, grad_box_sums):
    """Gradients of sum_l (g_cls[l] * cls_sum[l] + g_box[l] * box_sum[l]) w.r.t. every head, ONE launch.
    grad_*: float32 CUDA vectors [L] (or None = zeros), read on the device."""
    dev = cls_heads[0].device
    with torch.cuda.device(dev):
        dcls = [torch.empty_like(t) for t in cls_heads]
        dbox = [torch.empty_like(t) for t in box_heads]
        for g, t in zip(dcls + dbox, list(cls_heads) + list(box_heads)):
            if g.stride() != t.stride():
                raise RuntimeError('retina_loss_levels_backward: could not allocate gradients in the heads\' layout')
        arr, n, (b, a, c, nb, dtype) = _loss_levels(cls_heads, box_heads, depths, box_targets, (dcls, dbox))

        def vec(g):
            return None if g is None else g.detach().to(device=dev, dtype=torch.float32).reshape(n).contiguous()

        g_cls, g_box = vec(grad_cls_sums), vec(grad_box_sums)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(library().odtk_retina_loss_levels_backward(n, arr, b, a, c, nb, dtype, float(alpha), float(gamma), float(beta),
                                                          g_cls.data_ptr() if g_cls is not None else None,
                                                          g_box.data_ptr() if g_box is not None else None, stream),
               'retina_loss_levels_backward')
    return dcls, dbox


def retina_loss_forward(cls_head, box_head, depth, box_target, alpha, gamma, beta):
    """-> float64 CUDA tensor [3] = (sum of masked focal losses, sum of masked smooth-L1 losses, #foreground anchors)
    of one level (reference model.py:193-209 + loss.py:13-31 in one pass; csrc/loss.hpp)."""
    b, a, c, h, w, nb, dtype, lay = _loss_geometry(cls_head, box_head, depth, box_target)
    dev = cls_head.device
    with torch.cuda.device(dev):
        sums = torch.empty(3, dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(library().odtk_retina_loss_forward(cls_head.data_ptr(), box_head.data_ptr(), depth.data_ptr(),
                                                  box_target.data_ptr(), b, a, c, h, w, nb, dtype, lay, float(alpha),
                                                  float(gamma), float(beta), sums.data_ptr(), stream), 'retina_loss_forward')
    return sums


def retina_loss_backward(cls_head, box_head, depth, box_target, alpha, gamma, beta, grad_cls_sum, grad_box_sum):
    """Gradients of (g_cls * cls_sum + g_box * box_sum) w.r.t. the two heads, in the heads' dtype and layout.
    grad_*: float32 CUDA scalars (or None = 0), read on the device."""
    b, a, c, h, w, nb, dtype, lay = _loss_geometry(cls_head, box_head, depth, box_target)
    dev = cls_head.device

    def scalar(g):
        if g is None:
            return None
        g = g.detach().to(device=dev, dtype=torch.float32).reshape(1)
        return g

    g_cls, g_box = scalar(grad_cls_sum), scalar(grad_box_sum)
    with torch.cuda.device(dev):
        dcls = torch.empty_like(cls_head)                   # preserves the (dense) memory format
        dbox = torch.empty_like(box_head)
        if dcls.stride() != cls_head.stride() or dbox.stride() != box_head.stride():
            raise RuntimeError('retina_loss_backward: could not allocate gradients in the heads\' layout')
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(library().odtk_retina_loss_backward(cls_head.data_ptr(), box_head.data_ptr(), depth.data_ptr(),
                                                   box_target.data_ptr(), b, a, c, h, w, nb, dtype, lay, float(alpha),
                                                   float(gamma), float(beta),
                                                   g_cls.data_ptr() if g_cls is not None else None,
                                                   g_box.data_ptr() if g_box is not None else None,
                                                   dcls.data_ptr(), dbox.data_ptr(), stream), 'retina_loss_backward')
    return dcls, dbox


def bias_act_(y, bias, residual=None, relu=True):
    """In place on a channels_last activation: y = act(y + bias[c] (+ residual)).  `bias` is a float32
    device vector [C].  Replaces the separate bias / frozen-BN / residual-add / ReLU passes."""
    if not y.is_cuda or y.dim() != 4 or y.dtype not in _DTYPES:
        raise RuntimeError('bias_act_: y must be a 4-d CUDA tensor of float32/bfloat16/float16')
    n, c, h, w = y.shape
    if not (y.is_contiguous(memory_format=torch.channels_last) or h * w == 1):
        raise RuntimeError('bias_act_: y must be channels_last')
    if bias.dtype != torch.float32 or bias.numel() != c or not bias.is_cuda:
        raise RuntimeError('bias_act_: bias must be a float32 CUDA vector of length C')
    if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or not (
            residual.is_contiguous(memory_format=torch.channels_last) or h * w == 1)):
        raise RuntimeError('bias_act_: residual must match y (shape, dtype, channels_last)')
    with torch.cuda.device(y.device):
        stream = torch.cuda.current_stream(y.device).cuda_stream
        _check(library().odtk_bias_act(y.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                                       n * h * w, c, _DTYPES[y.dtype], 1 if relu else 0, stream), 'bias_act')
    _count('bias_act_kernel', y.numel() * y.element_size() * (3 if residual is not None else 2))
    return y


def bias_act_maxpool(y, bias, relu=True):
    """maxpool3x3/s2/p1(act(y + bias[c])) of a channels_last bf16/fp16 activation in one pass (the ResNet
    stem after conv1); bit-identical to bias_act_ followed by F.max_pool2d(., 3, 2, 1)."""
    if not y.is_cuda or y.dim() != 4 or y.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError('bias_act_maxpool: y must be a 4-d CUDA tensor of bfloat16/float16')
    n, c, h, w = y.shape
    if not y.is_contiguous(memory_format=torch.channels_last) or c % 8:
        raise RuntimeError('bias_act_maxpool: y must be channels_last with channels % 8 == 0')
    if bias.dtype != torch.float32 or bias.numel() != c or not bias.is_cuda:
        raise RuntimeError('bias_act_maxpool: bias must be a float32 CUDA vector of length C')
    out = torch.empty((n, c, (h + 1) // 2, (w + 1) // 2), dtype=y.dtype, device=y.device, memory_format=torch.channels_last)
    with torch.cuda.device(y.device):
        stream = torch.cuda.current_stream(y.device).cuda_stream
        _check(library().odtk_bias_act_maxpool(y.data_ptr(), bias.data_ptr(), out.data_ptr(), n, h, w, c, _DTYPES[y.dtype],
                                               1 if relu else 0, stream), 'bias_act_maxpool')
    _count('bias_act_maxpool_kernel', (y.numel() + out.numel()) * y.element_size())
    return out


def upsample2x(x):
    """Nearest-neighbour 2x upsampling of a channels_last CUDA activation (the FPN's top-down path); bit-identical to
    F.interpolate(x, scale_factor=2) in channels_last.  One HIP stream kernel instead of torch's upsample + layout copy."""
    if not x.is_cuda or x.dim() != 4 or x.dtype not in _DTYPES:
        raise RuntimeError('upsample2x: x must be a 4-d CUDA tensor of float32/bfloat16/float16')
    n, c, h, w = x.shape
    if not x.is_contiguous(memory_format=torch.channels_last) or (c * x.element_size()) % 16:
        raise RuntimeError('upsample2x: x must be channels_last with channels * element size a multiple of 16 bytes')
    out = torch.empty((n, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(library().odtk_upsample_nearest2x(x.data_ptr(), out.data_ptr(), n, h, w, c, _DTYPES[x.dtype], stream), 'upsample2x')
    _count('upsample_nearest2x_kernel', (x.numel() + out.numel()) * x.element_size())
    return out


def stem_pack(x, dtype):
    """2x2 space-to-depth pack of the network input [B, 3, H, W] (float32 / bfloat16 / float16, NCHW- or channels_last-contiguous, H and
    W even) into [B, 16, H/2, W/2] channels_last of `dtype` (bf16 / fp16): channel (dy*2+dx)*3+c of pixel (y, x) is x[:, c, 2y+dy, 2x+dx],
    channels 12..15 are zero (include/odtk_hip.h: odtk_stem_pack).  The cast and the layout change of the input, in one pass."""
    if not x.is_cuda or x.dim() != 4 or x.shape[1] != 3 or x.dtype not in _DTYPES or dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError('stem_pack: x must be a [B, 3, H, W] CUDA tensor of float32/bfloat16/float16, dtype bfloat16/float16')
    n, _, h, w = x.shape
    if h % 2 or w % 2:
        raise RuntimeError('stem_pack: height and width must be even')
    if x.is_contiguous():
        cl = 0
    elif x.is_contiguous(memory_format=torch.channels_last):
        cl = 1
    else:
        raise RuntimeError('stem_pack: x must be contiguous (NCHW or channels_last)')
    out = torch.empty((n, 16, h // 2, w // 2), dtype=dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(library().odtk_stem_pack(x.data_ptr(), out.data_ptr(), n, h, w, _DTYPES[x.dtype], cl, _DTYPES[dtype], stream), 'stem_pack')
    return out


_GEMM_WORKSPACE = {}
_GEMM_READY = []


def _gemm_setup(device):
    """Bind hipBLASLt (the copy PyTorch ships, so that one copy of the soname serves the process) and keep one 32 MiB
    scratch buffer per (device, stream) -- GEMMs on different streams may run at the same time.  Inside a hipGraph
    capture the buffer is allocated for the call and belongs to the graph (see `_workspace`)."""
    if not gemm_available():
        raise RuntimeError('gemm_bias_act: hipBLASLt could not be bound (odtk_gemm_init failed)')
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(32 << 20, dtype=torch.uint8, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    if key not in _GEMM_WORKSPACE:
        _GEMM_WORKSPACE[key] = torch.empty(32 << 20, dtype=torch.uint8, device=device)
    return _GEMM_WORKSPACE[key]


def gemm_available():
    """True when hipBLASLt could be bound (it ships with ROCm and with PyTorch-ROCm).  When it cannot, the
    fused graph keeps its 1x1 convolutions on MIOpen + the HIP epilogue -- slower, same results."""
    if not _GEMM_READY:
        shipped = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libhipblaslt.so')
        path = shipped if os.path.exists(shipped) else None
        rc = library().odtk_gemm_init(path.encode() if path else None)
        _GEMM_READY.append(rc == OK)
    return _GEMM_READY[0]


def gemm_bias_act(x, weight, bias, residual=None, relu=True):
    """1x1 convolution of a channels_last activation as one hipBLASLt GEMM with bias (+ residual)
    (+ ReLU) in its epilogue: returns act(conv1x1(x, weight) + bias (+ residual)), channels_last.
    x [B, Cin, H, W] channels_last, weight [Cout, Cin] (or [Cout, Cin, 1, 1]), bias float32 [Cout]."""
    if not x.is_cuda or x.dim() != 4 or x.dtype not in _DTYPES:
        raise RuntimeError('gemm_bias_act: x must be a 4-d CUDA tensor of float32/bfloat16/float16')
    b, k, h, w = x.shape
    n = weight.shape[0]
    if weight.numel() != n * k or weight.dtype != x.dtype or not weight.is_contiguous() and not weight.is_contiguous(
            memory_format=torch.channels_last):
        raise RuntimeError('gemm_bias_act: weight must be a dense [Cout, Cin(, 1, 1)] tensor of x.dtype')
    if not (x.is_contiguous(memory_format=torch.channels_last) or h * w == 1):
        raise RuntimeError('gemm_bias_act: x must be channels_last')
    if bias.dtype != torch.float32 or bias.numel() != n or not bias.is_cuda:
        raise RuntimeError('gemm_bias_act: bias must be a float32 CUDA vector of length Cout')
    y = torch.empty((b, n, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or not (
            residual.is_contiguous(memory_format=torch.channels_last) or h * w == 1)):
        raise RuntimeError('gemm_bias_act: residual must match the output (shape, dtype, channels_last)')
    with torch.cuda.device(x.device):
        ws = _gemm_setup(x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(library().odtk_gemm_bias_act(y.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                            residual.data_ptr() if residual is not None else None,
                                            b * h * w, n, k, _DTYPES[x.dtype], 1 if relu else 0,
                                            ws.data_ptr(), ws.numel(), stream), 'gemm_bias_act')
    return y


# ---- libodtk_conv.so: k x k convolution with the bias + ReLU in its own epilogue (csrc/conv_ck.cpp) -----------------------------
_CONV_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libodtk_conv.so')
_conv_lib = []


def conv_library():
    """The loaded libodtk_conv.so, or None when it has not been built (the engine then keeps convolution + odtk_bias_act)."""
    if not _conv_lib:
        lib = None
        if os.path.isfile(_CONV_LIB_PATH) and not os.environ.get('ODTK_NO_CONV_LIBRARY'):
            lib = ctypes.CDLL(_CONV_LIB_PATH)
            lib.odtk_conv_bias_act.restype = ctypes.c_int
            lib.odtk_conv_bias_act.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 13 + [ctypes.c_void_p]
            lib.odtk_conv_bias_act_pads.restype = ctypes.c_int
            lib.odtk_conv_bias_act_pads.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 15 + [ctypes.c_void_p]
            lib.odtk_conv_last_plan.restype = ctypes.c_char_p
            lib.odtk_conv_last_plan.argtypes = []
            lib.odtk_conv_instance_count.restype = ctypes.c_int
            lib.odtk_conv_instance_count.argtypes = [ctypes.c_int]
            lib.odtk_conv_plan_export.restype = ctypes.c_size_t
            lib.odtk_conv_plan_export.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
            lib.odtk_conv_plan_import.restype = ctypes.c_int
            lib.odtk_conv_plan_import.argtypes = [ctypes.c_char_p]
        _conv_lib.append(lib)
    return _conv_lib[0]


def conv_available():
    return conv_library() is not None


def conv_bias_act(x, weight, bias, stride=(1, 1), padding=(1, 1), relu=True, out=None):
    """act(conv2d(x, weight) + bias) of a channels_last bf16 / fp16 activation in ONE launch: a composable_kernel implicit-GEMM
    convolution with the bias and the ReLU in its epilogue (include/odtk_conv.h: odtk_conv_bias_act).  x [B, Cin, H, W]
    channels_last, weight [Cout, Cin, kh, kw] channels_last (= K, Y, X, C in memory), bias [Cout] of x.dtype.  The first call
    for a problem times the library's instances on the current stream (one synchronisation each); later calls only enqueue.
    Raises RuntimeError (ODTK_ERR_UNSUPPORTED) when no instance takes the problem."""
    lib = conv_library()
    if lib is None:
        raise ImportError('odtk._C: %s is missing -- build it (python __graft_entry__.py, or make -C retinanet-examples_amd/csrc conv)'
                          % _CONV_LIB_PATH)
    if not x.is_cuda or x.dim() != 4 or x.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError('conv_bias_act: x must be a 4-d CUDA tensor of bfloat16/float16')
    b, c, h, w = x.shape
    k, c2, kh, kw = weight.shape
    if c2 != c or weight.dtype != x.dtype or not weight.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError('conv_bias_act: weight must be [Cout, Cin, kh, kw] of x.dtype, channels_last')
    if not x.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError('conv_bias_act: x must be channels_last')
    if bias.dtype != x.dtype or bias.numel() != k or not bias.is_cuda or not bias.is_contiguous():
        raise RuntimeError('conv_bias_act: bias must be a CUDA vector of length Cout in the dtype of x')
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    # padding (before, after) per dimension: ((top, bottom), (left, right)) -- the space-to-depth stem needs (2, 1)
    (ph0, ph1), (pw0, pw1) = [(p, p) if isinstance(p, int) else p for p in (ph, pw)]
    ho, wo = (h + ph0 + ph1 - kh) // sh + 1, (w + pw0 + pw1 - kw) // sw + 1
    if out is None:
        y = torch.empty((b, k, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    else:                                                    # caller-owned output (the engine's head-tensor arena)
        if tuple(out.shape) != (b, k, ho, wo) or out.dtype != x.dtype or out.device != x.device or \
                not out.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError('conv_bias_act: out must be a channels_last %s tensor of shape %s' % (x.dtype, (b, k, ho, wo)))
        y = out
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(lib.odtk_conv_bias_act_pads(y.data_ptr(), x.data_ptr(), weight.data_ptr(), bias.data_ptr(), b, c, h, w, k, kh, kw,
                                           sh, sw, ph0, pw0, ph1, pw1, _DTYPES[x.dtype], 1 if relu else 0, stream), 'conv_bias_act')
    return y


def conv_last_plan():
    """'#index time name' of the instance the last conv_bias_act call of this thread ran (measurement records)."""
    lib = conv_library()
    return lib.odtk_conv_last_plan().decode() if lib is not None else ''


def _export_text(fn):
    need = fn(None, 0)
    buf = ctypes.create_string_buffer(int(need))
    fn(buf, need)
    return buf.value.decode()


def library_plans_export():
    """The choices the two kernel libraries under the engine made by stopwatch in this process, as text lines: hipBLASLt
    solutions ('gemm ...', include/odtk_hip.h: odtk_gemm_plan_export) and convolution instances ('conv ...', include/odtk_conv.h:
    odtk_conv_plan_export)."""
    text = _export_text(library().odtk_gemm_plan_export)
    lib = conv_library()
    if lib is not None:
        text += _export_text(lib.odtk_conv_plan_export)
    return text


def library_plans_import(text):
    """-> (gemm lines taken, conv lines taken): problems not yet planned in this process run on the named solution / instance
    without being timed."""
    data = text.encode()
    gemm_available()                                          # (binds hipBLASLt: the import itself only records)
    n_gemm = library().odtk_gemm_plan_import(data)
    lib = conv_library()
    return n_gemm, (lib.odtk_conv_plan_import(data) if lib is not None else 0)


def gemm_plan_pin_misses():
    return library().odtk_gemm_plan_pin_misses()


def loss_tuning(which, fp32_heads, threads, blocks_per_cu, unroll, box_blocks):
    """Debug / tuning: launch shape of the loss kernels of one form (0 forward with atomics, 1 backward, 2 forward through a
    workspace) and head width (include/odtk_hip.h: odtk_debug_loss_tuning)."""
    _check(library().odtk_debug_loss_tuning(int(which), int(bool(fp32_heads)), int(threads), int(blocks_per_cu),
                                            int(unroll), int(box_blocks)), 'loss_tuning')


def loss_layout(which, fp32_heads, per_wave=0, window=0, box_rows=1):
    """Debug / A-B: per-wave partial sums (workspace form only), contiguous trips of the loss kernels' logit walk, the backward's
    box-delta walk in memory order (include/odtk_hip.h: odtk_debug_loss_layout)."""
    _check(library().odtk_debug_loss_layout(int(which), int(bool(fp32_heads)), int(bool(per_wave)), int(bool(window)),
                                            int(bool(box_rows))), 'loss_layout')


LOSS_FORM_DEFAULT = 1      # = ODTK_LOSS_FORM_DEFAULT of include/odtk_hip.h (tests/test_abi_host.py compares the two)


def loss_form(form):
    """Debug / A-B: 0 = the symmetric element form everywhere, 1 (default) = vectors of negatives through the negatives-only
    form; 2..4 = timing ablations with wrong sums (include/odtk_hip.h: odtk_debug_loss_form)."""
    _check(library().odtk_debug_loss_form(int(form)), 'loss_form')


TRACE_WORDS = 16384     # odtk_debug_set_trace: >= 128 KiB (include/odtk_hip.h): coarse stamps in the first 8192 words, select_decode's
                        # per-segment fine stamps (16 words per segment, up to 512 segments) in the second 8192


def debug_set_trace(trace):
    """Debug trace buffer of the select_decode / nms kernels (include/odtk_hip.h: odtk_debug_set_trace), or None to switch it
    off.  `trace`: a zero-filled int64 CUDA tensor of at least TRACE_WORDS elements -- checked HERE because the library cannot:
    a 64 KiB buffer (the size of round 3's trace) is overrun by select_decode's fine stamps, which corrupts whatever the
    allocator placed behind it or faults (round 6, call 12: "Memory access fault by GPU" at the end of an 18-minute run)."""
    if trace is None:
        return library().odtk_debug_set_trace(None)
    if not (trace.is_cuda and trace.dtype == torch.int64 and trace.is_contiguous() and trace.numel() >= TRACE_WORDS):
        raise ValueError('the trace buffer must be a contiguous int64 CUDA tensor of >= %d elements (128 KiB)' % TRACE_WORDS)
    return library().odtk_debug_set_trace(trace.data_ptr())


def profile_enable(on=True, kernels=None):
    """Bracket kernel launches of the library with hipEvent pairs on their launch stream.
    kernels: iterable of names from KERNEL_NAMES (default: all)."""
    if not on:
        mask = 0
    elif kernels is None:
        mask = -1
    else:
        mask = 0
        for k in kernels:
            mask |= 1 << KERNEL_NAMES.index(k)
    _check(library().odtk_profile_enable(mask), 'profile_enable')


def profile_collect():
    """{kernel name: (total ms, launches)} since the last collect (waits for the recorded events)."""
    ms = (ctypes.c_double * len(KERNEL_NAMES))()
    n = (ctypes.c_int * len(KERNEL_NAMES))()
    _check(library().odtk_profile_collect(ms, n), 'profile_collect')
    return {name: (ms[i], n[i]) for i, name in enumerate(KERNEL_NAMES)}


class Engine:
    """Placeholder for the reference's TensorRT engine class (csrc/engine.h): the TensorRT / DALI /
    DeepStream deployment path is out of scope on MI355X (BASELINE.json north_star)."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('odtk._C.Engine: the TensorRT engine path is not available on MI355X')

    @staticmethod
    def load(path):
        raise NotImplementedError('odtk._C.Engine.load: the TensorRT engine path is not available on MI355X')

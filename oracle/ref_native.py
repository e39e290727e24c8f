"""ctypes wrapper over oracle/_ref/libodtk_ref_native.so (TEST INFRASTRUCTURE ONLY): the reference's OWN
device code -- rotated IoU / rotated NMS (csrc/cuda/nms_iou.cu:41-258, :324-375), the axis-aligned nms_kernel
(nms.cu:44-80) and the decode gather lambdas (decode.cu:121-159, decode_rotate.cu:116-167) -- compiled for
the CPU by oracle/ref_build/build_ref.py.  Used to pin oracle/c/odtk_oracle.c (and through it the HIP kernels) to the
reference source itself, and to generate tests/golden/rotated_ref_*.npz (oracle/gen_golden_native.py).
`available()` is False where the library was not built (e.g. /root/reference absent and no prebuilt .so)."""
import ctypes
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libodtk_ref_native.so')
_lib = None
_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def available():
    return os.path.isfile(_SO)


def library():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_SO)
        lib.odtk_ref_iou.restype = None
        lib.odtk_ref_iou.argtypes = [_f, _f, _f, ctypes.c_int, ctypes.c_int]
        lib.odtk_ref_nms_rotate.restype = ctypes.c_int
        lib.odtk_ref_nms_rotate.argtypes = [_f, _f, _f, ctypes.c_int, ctypes.c_float, ctypes.c_int, _f, _f, _f, _i]
        lib.odtk_ref_nms.restype = ctypes.c_int
        lib.odtk_ref_nms.argtypes = lib.odtk_ref_nms_rotate.argtypes
        lib.odtk_ref_decode_gather.restype = None
        lib.odtk_ref_decode_gather.argtypes = [_i, ctypes.c_int, ctypes.c_int, _f, _f] + [ctypes.c_int] * 5 + [_f, _f, _f, _f]
        _lib = lib
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def iou_pairs(boxes, anchors):
    """boxes [N, 8], anchors [M, 8] corner quads -> [M, N], exactly what iou_cuda_kernel writes."""
    boxes, anchors = _c(boxes).reshape(-1, 8), _c(anchors).reshape(-1, 8)
    out = np.empty((anchors.shape[0], boxes.shape[0]), np.float32)      # csrc/extensions.cpp:64-66
    library().odtk_ref_iou(boxes.ctypes.data_as(_f), anchors.ctypes.data_as(_f), out.ctypes.data_as(_f),
                           boxes.shape[0], anchors.shape[0])
    return out


def nms_rotate(scores, boxes, classes, thresh, ndet):
    """One image: scores [K], boxes [K, 6], classes [K] -> (scores [ndet], boxes [ndet, 6], classes [ndet],
    kept input positions [ndet], -1 padded), following odtk::cuda::nms_rotate."""
    return _nms(scores, boxes, classes, thresh, ndet, 6)


def nms_axis(scores, boxes, classes, thresh, ndet):
    """Same with [K, 4] boxes through the reference's nms_kernel (csrc/cuda/nms.cu:44-80)."""
    return _nms(scores, boxes, classes, thresh, ndet, 4)


def _nms(scores, boxes, classes, thresh, ndet, nb):
    scores, boxes, classes = _c(scores), _c(boxes), _c(classes)
    k = scores.shape[0]
    assert boxes.shape == (k, nb)
    s = np.empty(ndet, np.float32)
    b = np.empty((ndet, nb), np.float32)
    c = np.empty(ndet, np.float32)
    idx = np.empty(ndet, np.int32)
    fn = library().odtk_ref_nms_rotate if nb == 6 else library().odtk_ref_nms
    fn(scores.ctypes.data_as(_f), boxes.ctypes.data_as(_f), classes.ctypes.data_as(_f), k,
                                  float(thresh), int(ndet), s.ctypes.data_as(_f), b.ctypes.data_as(_f),
                                  c.ctypes.data_as(_f), idx.ctypes.data_as(_i))
    return s, b, c, idx.astype(np.int64)


def decode_gather(indices, scores, deltas, stride, anchors, num_classes, rotated=False):
    """The reference's own per-detection lambda (decode.cu:121-159 / decode_rotate.cu:116-167) on ONE image:
    flat score indices [K] -> (scores [K], boxes [K, 4|6], classes [K]).  scores [A*C, H, W], deltas [A*nb, H, W]."""
    scores, deltas, anchors = _c(scores), _c(deltas), _c(anchors).reshape(-1, 4)
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    nb = 6 if rotated else 4
    a = anchors.shape[0]
    _, h, w = scores.shape
    assert scores.shape[0] == a * num_classes and deltas.shape == (a * nb, h, w)
    k = idx.shape[0]
    s, b, c = np.empty(k, np.float32), np.empty((k, nb), np.float32), np.empty(k, np.float32)
    library().odtk_ref_decode_gather(idx.ctypes.data_as(_i), k, 1 if rotated else 0, scores.ctypes.data_as(_f),
                                     deltas.ctypes.data_as(_f), h, w, int(stride), a, int(num_classes),
                                     anchors.ctypes.data_as(_f), s.ctypes.data_as(_f), b.ctypes.data_as(_f),
                                     c.ctypes.data_as(_f))
    return s, b, c

"""ctypes wrapper over oracle/c/libodtk_oracle.so (TEST INFRASTRUCTURE ONLY).

numpy in / numpy out.  See oracle/c/odtk_oracle.c for what is restated and from which reference
lines.  The rotated IoU / NMS functions are pinned bit for bit to the reference's own device code compiled
for the CPU (oracle/ref_native.py, tests/test_oracle_native_ref.py), the decode gather / box step to its
decode lambdas (exact layout, boxes within 1e-4 after aligning the clamp convention)."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c')
_SO = os.path.join(_DIR, 'libodtk_oracle.so')
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.POINTER(ctypes.c_int64)


def library():
    global _lib
    if _lib is None:
        src = os.path.join(_DIR, 'odtk_oracle.c')
        if not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(['make', '-B'], cwd=_DIR, check=True, stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(_SO)
        lib.oracle_decode.restype = ctypes.c_int
        lib.oracle_decode.argtypes = [_f, _f] + [ctypes.c_int] * 6 + [_f, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                                     _f, _f, _f, _i64]
        lib.oracle_nms.restype = ctypes.c_int
        lib.oracle_nms.argtypes = [_f, _f, _f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                   ctypes.c_int, _f, _f, _f, _i64]
        lib.oracle_iou_pairs.restype = None
        lib.oracle_iou_pairs.argtypes = [_f, _f, ctypes.c_int, ctypes.c_int, _f]
        lib.oracle_rotated_overlap.restype = ctypes.c_float
        lib.oracle_rotated_overlap.argtypes = [_f, _f, ctypes.c_int]
        _lib = lib
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_f)


def decode(cls, box, stride, threshold, top_n, anchors, rotated=False):
    """-> scores [B, top_n], boxes [B, top_n, nb], classes [B, top_n], indices [B, top_n] (int64, -1 pad)"""
    cls, box, anchors = _c(cls), _c(box), _c(anchors)
    nb = 6 if rotated else 4
    B, AC, H, W = cls.shape
    A = anchors.shape[0]
    C = AC // A
    assert box.shape == (B, A * nb, H, W)
    s = np.empty((B, top_n), np.float32)
    b = np.empty((B, top_n, nb), np.float32)
    c = np.empty((B, top_n), np.float32)
    i = np.empty((B, top_n), np.int64)
    rc = library().oracle_decode(_p(cls), _p(box), B, A, C, H, W, int(stride), _p(anchors), float(threshold),
                                 int(top_n), nb, _p(s), _p(b), _p(c), i.ctypes.data_as(_i64))
    assert rc == 0
    return s, b, c, i


def nms(scores, boxes, classes, thresh, ndet, rotated=False, own_angle=False):
    scores, boxes, classes = _c(scores), _c(boxes), _c(classes)
    nb = 6 if rotated else 4
    B, count = scores.shape
    assert boxes.shape == (B, count, nb)
    s = np.empty((B, ndet), np.float32)
    b = np.empty((B, ndet, nb), np.float32)
    c = np.empty((B, ndet), np.float32)
    i = np.empty((B, ndet), np.int64)
    rc = library().oracle_nms(_p(scores), _p(boxes), _p(classes), B, count, nb, float(thresh), int(ndet),
                              1 if own_angle else 0, _p(s), _p(b), _p(c), i.ctypes.data_as(_i64))
    assert rc == 0
    return s, b, c, i


def iou_pairs(boxes, anchors):
    """boxes [N, 8], anchors [M, 8] corner quads -> [M, N] (layout of csrc/extensions.cpp:64-66)."""
    boxes, anchors = _c(boxes).reshape(-1, 8), _c(anchors).reshape(-1, 8)
    out = np.empty((anchors.shape[0], boxes.shape[0]), np.float32)
    library().oracle_iou_pairs(_p(boxes), _p(anchors), boxes.shape[0], anchors.shape[0], _p(out))
    return out


def rotated_overlap(m, j, own_angle=False):
    m, j = _c(m), _c(j)
    return float(library().oracle_rotated_overlap(_p(m), _p(j), 1 if own_angle else 0))


# ---------------------------------------------------------------------------------------------
# Independent float64 cross-check for the rotated IoU (shapely is not installed): textbook
# Sutherland-Hodgman clip of two convex quads, orientation-normalised, no padding tricks.
# ---------------------------------------------------------------------------------------------
def _area64(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def convex_iou_f64(qa, qb):
    """IoU of two convex quads given as [4, 2] float arrays (any orientation)."""
    a = np.asarray(qa, np.float64).reshape(4, 2)
    b = np.asarray(qb, np.float64).reshape(4, 2)
    if _area64(a) < 0:
        a = a[::-1]
    if _area64(b) < 0:
        b = b[::-1]
    poly = [tuple(p) for p in a]
    for k in range(4):
        p1, p2 = b[k], b[(k + 1) % 4]
        ex, ey = p2[0] - p1[0], p2[1] - p1[1]

        def side(p):
            return ex * (p[1] - p1[1]) - ey * (p[0] - p1[0])      # >= 0: inside (left of the CCW edge)

        out = []
        for i in range(len(poly)):
            cur, nxt = poly[i], poly[(i + 1) % len(poly)]
            sc, sn = side(cur), side(nxt)
            if sc >= 0:
                out.append(cur)
            if (sc >= 0) != (sn >= 0):
                t = sc / (sc - sn)
                out.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
        poly = out
        if not poly:
            break
    inter = abs(_area64(np.array(poly))) if len(poly) > 2 else 0.0
    union = abs(_area64(a)) + abs(_area64(b)) - inter
    return inter / union if union > 0 else 0.0


def box6_to_quad(b, s=None, c=None):
    """[x1,y1,x2,y2,sin,cos] -> [4,2] corners, float64 (nms_iou.cu:199-228 geometry)."""
    b = np.asarray(b, np.float64)
    s = b[4] if s is None else s
    c = b[5] if c is None else c
    cx, cy = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2
    pts = []
    for dx, dy in ((b[0] - cx, b[1] - cy), (b[2] - cx, b[1] - cy), (b[2] - cx, b[3] - cy), (b[0] - cx, b[3] - cy)):
        pts.append((dx * c - dy * s + cx, dy * c + dx * s + cy))
    return np.array(pts)

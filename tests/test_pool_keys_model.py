"""The stem's pool pass takes its maximum on 16-bit integer keys of the raw bf16 / fp16 patterns (csrc/epilogue.hpp: order_keys16,
round 6).  What the kernel relies on, checked here EXHAUSTIVELY over all 65 536 patterns of both types on the CPU, with the
constants read from the shipped source:

  * k = h ^ ((h >> 15) & 0x7fff), read as int16, is an involution and orders every non-NaN pattern as its value (with -0 below +0);
  * keys of positive NaNs lie above +inf's key, keys of negative NaNs below -inf's key -- and the kernel's per-dtype constant
    (`kNegInfKey`) IS -inf's key (GPU call 31: the bf16 constant was used for fp16 too; found by the special-values GPU test);
  * a running (max, min) over a window therefore yields: a NaN iff the window holds one, else the window's maximum.
"""
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, 'retinanet-examples_amd', 'csrc', 'epilogue.hpp')).read()


def keys(h):
    h = h.astype(np.uint16)
    sign = (h >> 15).astype(np.uint16) * np.uint16(0x7fff)                  # (h >> 15) & 0x7fff with an arithmetic shift of the int16
    return (h ^ sign).astype(np.uint16).view(np.int16)


def values(dtype):
    h = np.arange(65536, dtype=np.int64).astype(np.uint16)
    v = torch.from_numpy(h.view(np.int16).copy()).view(dtype).double().numpy()
    return h, v


def test_the_source_holds_the_transform_and_the_constants_this_test_models():
    assert 'w ^ (__builtin_bit_cast(uint32_t, sign) & 0x7fff7fffu)' in SRC and '>> 15' in SRC
    m = re.search(r'kNegInfKey = std::is_same_v<T, BF16> \? (-\d+) : (-\d+);', SRC)
    assert m, 'csrc/epilogue.hpp: kNegInfKey not found'
    assert 'kn < kNegInfKey ? kn : kx' in SRC


def test_keys_order_every_pattern_and_bracket_the_nans():
    m = re.search(r'kNegInfKey = std::is_same_v<T, BF16> \? (-\d+) : (-\d+);', SRC)
    shipped = {torch.bfloat16: int(m.group(1)), torch.float16: int(m.group(2))}
    for dtype in (torch.bfloat16, torch.float16):
        h, v = values(dtype)
        k = keys(h)
        assert np.array_equal(keys(k.view(np.uint16)).view(np.uint16), h)                      # involution
        assert len(np.unique(k)) == 65536
        nan = np.isnan(v)
        order = np.argsort(k[~nan], kind='stable')
        vs = v[~nan][order]
        assert np.all(np.diff(vs) >= 0)                                                         # value order == key order ...
        z = np.where(vs == 0)[0]
        assert len(z) == 2 and np.signbit(vs[z[0]]) and not np.signbit(vs[z[1]])               # ... with -0 just below +0
        k_pinf, k_ninf = int(k[v == np.inf][0]), int(k[v == -np.inf][0])
        pos_nan, neg_nan = nan & (h < 0x8000), nan & (h >= 0x8000)
        assert pos_nan.sum() > 0 and neg_nan.sum() > 0
        assert k[pos_nan].min() > k_pinf and k[~nan].max() == k_pinf
        assert k[neg_nan].max() < k_ninf and k[~nan].min() == k_ninf
        assert shipped[dtype] == k_ninf, (dtype, shipped[dtype], k_ninf)                         # the kernel tests `kn < kNegInfKey`


def test_running_max_and_min_give_the_window_maximum_or_a_nan():
    rng = np.random.default_rng(5)
    m = re.search(r'kNegInfKey = std::is_same_v<T, BF16> \? (-\d+) : (-\d+);', SRC)
    shipped = {torch.bfloat16: int(m.group(1)), torch.float16: int(m.group(2))}
    for dtype in (torch.bfloat16, torch.float16):
        _, v = values(dtype)
        specials = np.array([0x0000, 0x8000, 0x7f80, 0xff80, 0x7fc1, 0xffc1, 0x7c00, 0xfc00, 0x7e01, 0xfe01, 0xffff, 0x7fff], dtype=np.uint16)
        win = rng.integers(0, 65536, size=(20000, 9)).astype(np.uint16)
        win[:4000, rng.integers(0, 9)] = rng.choice(specials, size=4000)
        win[4000:6000] = rng.choice(specials, size=(2000, 9))
        k = keys(win)
        kx, kn = k.max(axis=1), k.min(axis=1)
        kk = np.where(kn < shipped[dtype], kn, kx)
        out = v[keys(kk.view(np.uint16)).view(np.uint16)]                                     # back to bits (involution), then to values
        wv = v[win]
        has_nan = np.isnan(wv).any(axis=1)
        assert np.array_equal(np.isnan(out), has_nan)
        assert np.array_equal(out[~has_nan], wv[~has_nan].max(axis=1))

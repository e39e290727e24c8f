"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/odtk_hip.h declares, the two-phase workspace queries and argument validation work without
a GPU (they never touch HIP), and the Python surface mirrors the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from odtk import _C, box

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATIOS = [1.0, 2.0, 0.5]
SCALES = [4 * 2 ** (i / 3) for i in range(3)]


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, 'include', 'odtk_hip.h')).read()
    declared = set(re.findall(r'\b(odtk_[a-z0-9_]+)\s*\(', header))
    assert {'odtk_decode', 'odtk_decode_rotate', 'odtk_nms', 'odtk_nms_rotate', 'odtk_iou',
            'odtk_decode_levels', 'odtk_nms_ex', 'odtk_detect'} <= declared
    lib = _C.library()
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_C.exported_symbols()) == declared
    assert lib.odtk_version().startswith(b'odtk-hip')


def test_decode_workspace_query_and_validation():
    lib = _C.library()
    anchors = (ctypes.c_float * 36)(*[0.0] * 36)
    # two-phase convention of the reference (decode.cu:53-72): NULL workspace -> bytes needed
    size = lib.odtk_decode(8, None, None, 100, 160, 8, 9, 80, anchors, 36, 0.05, 1000, None, 0, None)
    assert size > 0
    # candidate pool: min(n, 2^20) keys of 8 B per image (+ counters), 256-B aligned
    assert size >= 8 * (1 << 20) * 8 and size % 256 == 0
    small = lib.odtk_decode(2, None, None, 7, 10, 128, 9, 80, anchors, 36, 0.05, 1000, None, 0, None)
    assert 0 < small < size
    # validation (never reaches HIP)
    assert lib.odtk_decode(8, None, None, 0, 160, 8, 9, 80, anchors, 36, 0.05, 1000, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_decode(8, None, None, 100, 160, 8, 9, 80, anchors, 35, 0.05, 1000, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_decode(8, None, None, 100, 160, 8, 9, 80, anchors, 36, 0.05, _C.MAX_TOP_N + 1, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_decode(0, None, None, 100, 160, 8, 9, 80, anchors, 36, 0.05, 1000, None, 0, None) == _C.ERR_INVALID
    # a too-small workspace is reported, not written through
    buf = ctypes.create_string_buffer(1024)
    ins = (ctypes.c_void_p * 2)(1 << 20, 1 << 20)
    outs = (ctypes.c_void_p * 3)(1 << 20, 1 << 20, 1 << 20)
    assert lib.odtk_decode(8, ins, outs, 100, 160, 8, 9, 80, anchors, 36, 0.05, 1000,
                           ctypes.cast(buf, ctypes.c_void_p), 1024, None) == _C.ERR_WORKSPACE


def test_nms_workspace_query_and_limits():
    lib = _C.library()
    assert lib.odtk_nms(8, None, None, 5000, 100, 0.5, None, 0, None) > 0
    assert lib.odtk_nms_rotate(8, None, None, 5000, 100, 0.5, None, 0, None) > 0
    # beyond the LDS-resident size the key lists move to the workspace: the query says how much (the reference has no cap)
    assert lib.odtk_nms(8, None, None, _C.MAX_NMS_COUNT, 100, 0.5, None, 0, None) == 256
    big = lib.odtk_nms(8, None, None, _C.MAX_NMS_COUNT + 1, 100, 0.5, None, 0, None)
    assert big >= 8 * (_C.MAX_NMS_COUNT + 1) * 8 and big % 256 == 0
    assert lib.odtk_nms(8, None, None, _C.MAX_NMS_COUNT_SCRATCH + 1, 100, 0.5, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_nms(8, None, None, 0, 100, 0.5, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_iou(None, None, 4, 4, None) == _C.ERR_INVALID


def test_detect_workspace_covers_decode_plus_candidates():
    lib = _C.library()
    anchors = (ctypes.c_float * 36)(*[0.0] * 36)
    lv = (_C.Level * 2)()
    for i, (h, w, s) in enumerate([(25, 40, 32), (13, 20, 64)]):
        lv[i].height, lv[i].width, lv[i].stride = h, w, s
        lv[i].anchors = ctypes.cast(anchors, ctypes.POINTER(ctypes.c_float))
    dec = lib.odtk_decode_levels(4, 2, lv, 9, 80, _C.F32, 0, 0.05, 1000, None, 0, None, 0, None)
    det = lib.odtk_detect(4, 2, lv, 9, 80, _C.F32, 0, 0.05, 1000, 0.5, 100, None, None, 0, None)
    assert dec > 0 and det >= dec + 4 * 2000 * 6 * 4
    # bf16 / fp16 logits and channels_last are first-class; a 16-bit call needs no more scratch than the fp32 one (its spans
    # are twice as long, so it has half as many candidate regions)
    lv[0].channels_last = 1
    half = lib.odtk_decode_levels(4, 2, lv, 9, 80, _C.BF16, _C.FLAG_LOGITS, 0.05, 1000, None, 0, None, 0, None)
    assert 0 < half <= dec
    assert lib.odtk_decode_levels(4, 2, lv, 9, 80, _C.F16, _C.FLAG_LOGITS, 0.05, 1000, None, 0, None, 0, None) == half
    assert lib.odtk_decode_levels(4, 2, lv, 9, 80, 7, 0, 0.05, 1000, None, 0, None, 0, None) == _C.ERR_UNSUPPORTED
    lv[0].channels_last = 2
    assert lib.odtk_decode_levels(4, 2, lv, 9, 80, _C.F32, 0, 0.05, 1000, None, 0, None, 0, None) == _C.ERR_INVALID


def test_hip_boundary_has_no_cpu_fallback():
    """The operator boundary (odtk._C) and the batched HIP forms never run on the CPU: CPU tensors raise.  Only the
    reference's own "no GPU" plumbing branch exists on the Python surface (box.decode / box.nms on CPU tensors,
    pure torch, tests/test_config0_cpu.py) -- and it is chosen by the tensor's device, never as a fallback for a
    missing library."""
    cls = torch.rand(1, 9 * 4, 3, 3)
    deltas = torch.zeros(1, 36, 3, 3)
    anchors = box.generate_anchors(8, RATIOS, SCALES)
    with pytest.raises(RuntimeError, match='must be on the GPU'):
        box.detect([cls], [deltas], [8], {8: anchors})
    with pytest.raises(RuntimeError, match='must be on the GPU'):
        box.decode_levels([cls], [deltas], [8], 0.05, 10, {8: anchors})
    with pytest.raises(RuntimeError, match='must be on the GPU'):
        box.nms_rotated(torch.rand(1, 5), torch.rand(1, 5, 6), torch.zeros(1, 5))
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _C.decode(cls, deltas, anchors.view(-1).tolist(), 8, 0.05, 10)
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        _C.nms(torch.rand(1, 5), torch.rand(1, 5, 4), torch.zeros(1, 5), 0.5, 3)
    with pytest.raises(RuntimeError, match='must be on the GPU'):
        _C.retina_loss_forward(cls, deltas, torch.zeros(1, 9, 1, 3, 3), torch.zeros(1, 9, 4, 3, 3), 0.25, 2.0, 0.11)
    with pytest.raises(NotImplementedError):
        _C.Engine()


def test_anchors_match_reference_fixture(golden_dir):
    with np.load(os.path.join(golden_dir, 'anchors.npz')) as g:
        for s in (8, 16, 32, 64, 128):
            a = box.generate_anchors(s, RATIOS, SCALES).numpy()
            assert np.array_equal(a.view(np.uint32), g['s%d' % s].view(np.uint32))
            ax, rot = box.generate_anchors_rotated(s, RATIOS, SCALES, [-np.pi / 6, 0, np.pi / 6])
            assert np.array_equal(ax.numpy().view(np.uint32), g['rot_axis_s%d' % s].view(np.uint32))
            assert np.array_equal(rot.numpy().view(np.uint32), g['rot_pts_s%d' % s].view(np.uint32))


def test_delta_roundtrip_matches_oracle():
    from oracle import box_oracle
    g = torch.Generator().manual_seed(5)
    anchors = torch.rand(50, 2, generator=g) * 100
    anchors = torch.cat([anchors, anchors + torch.rand(50, 2, generator=g) * 60 + 4], 1)
    deltas = torch.randn(50, 4, generator=g) * 0.3
    mine = box.delta2box(deltas, anchors, [40, 25], 8)
    ref = box_oracle.delta2box(deltas, anchors, [40, 25], 8)
    assert torch.equal(mine, ref)
    back = box.box2delta(mine, anchors)
    inside = ((mine > 0) & (mine < torch.tensor([319., 199., 319., 199.]))).all(1)
    assert torch.allclose(back[inside], deltas[inside], atol=1e-4)


def test_new_entry_points_validate_before_touching_the_device():
    lib = _C.library()
    # odtk_gemm_bias_act: null / aliased / misaligned pointers and bad sizes are rejected on the host
    assert lib.odtk_gemm_bias_act(None, None, None, None, None, 16, 8, 8, _C.BF16, 1, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_gemm_bias_act(256, 512, 768, 1024, 256, 16, 8, 8, _C.BF16, 1, None, 0, None) == _C.ERR_INVALID   # residual == y
    assert lib.odtk_gemm_bias_act(256, 512, 768, 1024, None, 16, 0, 8, _C.BF16, 1, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_gemm_bias_act(264, 512, 768, 1024, None, 16, 8, 8, _C.BF16, 1, None, 0, None) == _C.ERR_INVALID   # 16-B alignment
    assert lib.odtk_gemm_bias_act(256, 512, 768, 1024, None, 16, 8, 8, 9, 1, None, 0, None) == _C.ERR_UNSUPPORTED
    # odtk_bias_act_maxpool: 16-bit dtypes, channels % 8 == 0
    assert lib.odtk_bias_act_maxpool(256, 512, 768, 1, 4, 4, 8, _C.F32, 1, None) == _C.ERR_UNSUPPORTED
    assert lib.odtk_bias_act_maxpool(256, 512, 768, 1, 4, 4, 12, _C.BF16, 1, None) == _C.ERR_UNSUPPORTED
    assert lib.odtk_bias_act_maxpool(None, 512, 768, 1, 4, 4, 8, _C.BF16, 1, None) == _C.ERR_INVALID
    # head bias fold: logits + 16-bit + channels_last + A*C % 8 == 0, otherwise refused (never silently ignored)
    anchors = (ctypes.c_float * 36)(*[0.0] * 36)
    lv = (_C.Level * 1)()
    lv[0].height, lv[0].width, lv[0].stride, lv[0].channels_last = 5, 7, 8, 1
    lv[0].anchors = ctypes.cast(anchors, ctypes.POINTER(ctypes.c_float))
    lv[0].cls_bias = 4096
    ok = lib.odtk_decode_levels(2, 1, lv, 9, 80, _C.BF16, _C.FLAG_LOGITS, 0.05, 100, None, 0, None, 0, None)
    assert ok > 0
    assert lib.odtk_decode_levels(2, 1, lv, 9, 80, _C.BF16, 0, 0.05, 100, None, 0, None, 0, None) == _C.ERR_UNSUPPORTED
    assert lib.odtk_decode_levels(2, 1, lv, 9, 80, _C.F32, _C.FLAG_LOGITS, 0.05, 100, None, 0, None, 0, None) == _C.ERR_UNSUPPORTED
    assert lib.odtk_decode_levels(2, 1, lv, 9, 3, _C.BF16, _C.FLAG_LOGITS, 0.05, 100, None, 0, None, 0, None) == _C.ERR_UNSUPPORTED
    lv[0].channels_last = 0
    assert lib.odtk_decode_levels(2, 1, lv, 9, 80, _C.BF16, _C.FLAG_LOGITS, 0.05, 100, None, 0, None, 0, None) == _C.ERR_UNSUPPORTED
    # the prefilter's threshold table: 16-byte aligned, 16-bit dtypes, A*C % 8 == 0
    lv[0].channels_last, lv[0].cls_thresholds = 1, 4096 + 8
    assert lib.odtk_decode_levels(2, 1, lv, 9, 80, _C.BF16, _C.FLAG_LOGITS, 0.05, 100, None, 0, None, 0, None) == _C.ERR_INVALID
    assert lib.odtk_prefilter_thresholds(None, 720, _C.BF16, 0.05, 4096, None) == _C.ERR_INVALID
    assert lib.odtk_prefilter_thresholds(4096, 720, _C.BF16, 0.05, 4096 + 8, None) == _C.ERR_INVALID
    assert lib.odtk_prefilter_thresholds(4096, 723, _C.BF16, 0.05, 8192, None) == _C.ERR_INVALID
    assert lib.odtk_prefilter_thresholds(4096, 720, _C.F32, 0.05, 8192, None) == _C.ERR_UNSUPPORTED


def test_loss_tuning_hook_validates_without_a_gpu():
    """odtk_debug_loss_tuning (launch shape of the loss kernels) never touches HIP: bad shapes are refused, good ones stick
    until set back.  The built-in defaults are the ones tools/loss_probe.py measured (profiles/r03_loss_probe.txt)."""
    lib = _C.library()
    for bad in ((0, 1, 100, 1, 4, 64), (0, 1, 2048, 1, 4, 64), (0, 1, 512, 0, 4, 64), (0, 1, 512, 1, 3, 64),
                (1, 0, 256, 4, 1, 0), (1, 0, 256, 65, 1, 256), (3, 0, 256, 4, 1, 256), (-1, 1, 256, 4, 1, 256)):
        assert lib.odtk_debug_loss_tuning(*bad) == _C.ERR_INVALID, bad
    for good in ((0, 0, 512, 1, 2, 64), (1, 0, 256, 4, 1, 256), (0, 1, 512, 1, 4, 64), (1, 1, 256, 8, 2, 1024)):   # = the defaults
        assert lib.odtk_debug_loss_tuning(*good) == 0, good
    with pytest.raises(RuntimeError, match='invalid argument'):
        _C.loss_tuning(0, 1, 100, 1, 4, 64)


def test_loss_layout_hook_validates_without_a_gpu():
    """odtk_debug_loss_layout (per-wave sums of the workspace form, contiguous trips, the backward's box-delta walk in memory
    order) never touches HIP; per-wave sums exist in the workspace form only, and the workspace size query follows the switch
    (3 doubles per WAVE instead of per workgroup)."""
    lib = _C.library()
    for bad in ((3, 1, 0, 0, 1), (-1, 0, 0, 0, 1), (2, 1, 2, 0, 1), (2, 1, 0, 2, 1), (1, 1, 0, 0, 2), (0, 1, 1, 0, 1), (1, 0, 1, 1, 1)):
        assert lib.odtk_debug_loss_layout(*bad) == _C.ERR_INVALID, bad
    levels = (_C.LossLevel * 1)()
    levels[0].cls = levels[0].box = levels[0].depth = levels[0].box_target = 1 << 20       # never dereferenced on this path
    levels[0].height, levels[0].width, levels[0].channels_last = 100, 160, 1
    args = (1, levels, 2, 9, 80, 4, _C.F32, 0.25, 2.0, 0.11)
    lib.odtk_debug_loss_tuning(2, 1, 256, 4, 1, 256)
    assert lib.odtk_debug_loss_layout(2, 1, 0, 0, 1) == 0
    per_group = lib.odtk_retina_loss_levels_forward_ws(*args, None, None, 0, None)
    assert lib.odtk_debug_loss_layout(2, 1, 1, 1, 1) == 0
    per_wave = lib.odtk_retina_loss_levels_forward_ws(*args, None, None, 0, None)
    assert per_group > 0 and per_wave == 4 * per_group                                      # 256 threads = 4 waves
    lib.odtk_debug_loss_tuning(2, 1, 256, 4, 2, 256)
    for which, fp32, window in ((0, 0, 0), (0, 1, 0), (1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 1)):   # = the defaults
        _C.loss_layout(which, fp32, 0, window, 1)
    with pytest.raises(RuntimeError, match='invalid argument'):
        _C.loss_layout(0, 1, 1, 0, 1)


def test_loss_form_hook_validates_without_a_gpu():
    """odtk_debug_loss_form (arithmetic form of the gamma = 2 classification walk): 0 / 1; the
    binding's default is the header's ODTK_LOSS_FORM_DEFAULT."""
    lib = _C.library()
    header = open(os.path.join(ROOT, 'include', 'odtk_hip.h')).read()
    assert int(re.search(r'#define ODTK_LOSS_FORM_DEFAULT\s+(\d+)', header).group(1)) == _C.LOSS_FORM_DEFAULT
    # the timing ablations 2..4, 6, 7 (wrong sums on purpose) are NOT in the shipped library: only a -DODTK_LOSS_ABLATIONS build
    # (make -C retinanet-examples_amd/csrc ablations -> build_ablate/) accepts them
    for bad in (-1, 2, 3, 4, 5, 6, 7, 8):
        assert lib.odtk_debug_loss_form(bad) == _C.ERR_INVALID
    for good in (0, 1, _C.LOSS_FORM_DEFAULT):
        assert lib.odtk_debug_loss_form(good) == 0
    # ABI guard: the library's struct sizes are what the ctypes mirrors have (library() refuses to load otherwise)
    assert [lib.odtk_abi_struct_size(i) for i in range(5)] == [ctypes.sizeof(_C.Level), ctypes.sizeof(_C.SnapLevel),
                                                               ctypes.sizeof(_C.SnapRotLevel), ctypes.sizeof(_C.LossLevel), -1]
    with pytest.raises(RuntimeError, match='invalid argument'):
        _C.loss_form(5)


def test_loss_forward_workspace_query_without_a_gpu():
    """odtk_retina_loss_levels_forward_ws: NULL workspace -> bytes needed (3 doubles per workgroup of the launch, 256-B
    aligned), a too-small workspace is reported before anything is launched."""
    lib = _C.library()
    levels = (_C.LossLevel * 2)()
    for lv, (h, w) in zip(levels, ((100, 160), (7, 10))):
        lv.cls = lv.box = lv.depth = lv.box_target = 1 << 20                   # never dereferenced on this path
        lv.height, lv.width, lv.channels_last = h, w, 1
    args = (2, levels, 2, 9, 80, 4, _C.F32, 0.25, 2.0, 0.11)
    need = lib.odtk_retina_loss_levels_forward_ws(*args, None, None, 0, None)
    assert need > 0 and need % 256 == 0 and need < (1 << 20)
    buf = ctypes.create_string_buffer(256)
    assert lib.odtk_retina_loss_levels_forward_ws(*args, 1 << 20, ctypes.cast(buf, ctypes.c_void_p), 16, None) == _C.ERR_WORKSPACE
    assert lib.odtk_retina_loss_levels_forward_ws(0, levels, 2, 9, 80, 4, _C.F32, 0.25, 2.0, 0.11,
                                                  None, None, 0, None) == _C.ERR_INVALID


def test_fastdiv_is_exact(tmp_path):
    """csrc/fastdiv.hpp (division by a launch-constant divisor: multiply-high + shifts, Granlund & Montgomery fig. 4.1)
    against `/` and `%`: every divisor the kernels can see at the edges, random pairs over the whole 32-bit range."""
    import subprocess
    src = tmp_path / 'fd.cpp'
    src.write_text(r'''
#include "%s"
#include <cstdio>
#include <random>
int main() {
  std::mt19937_64 g(1);
  unsigned long long bad = 0, n = 0;
  const uint32_t ds[] = {1, 2, 3, 4, 5, 7, 9, 27, 80, 90, 720, 2160, 1000, 4000, 16000, 65535, 65536, 65537,
                         0x7fffffffu, 0x80000000u, 0x80000001u, 0xfffffffeu, 0xffffffffu};
  for (uint32_t d : ds) {
    const odtk::FastDiv f = odtk::fastdiv_make(d);
    const uint32_t xs[] = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0xffffffffu, 0xfffffffeu, 0x80000000u, 0x7fffffffu};
    for (uint32_t x : xs) { ++n; if (odtk::fastdiv(x, f) != x / d) ++bad; }
    for (int i = 0; i < 100000; ++i) {
      const uint32_t x = static_cast<uint32_t>(g());
      uint32_t r;
      const uint32_t q = odtk::fastdivmod(x, f, &r);
      ++n;
      if (q != x / d || r != x %% d) ++bad;
    }
  }
  for (int i = 0; i < 1000000; ++i) {
    uint32_t d = static_cast<uint32_t>(g()) >> (g() %% 32);
    if (!d) d = 1;
    const uint32_t x = static_cast<uint32_t>(g()) >> (g() %% 32);
    ++n;
    if (odtk::fastdiv(x, odtk::fastdiv_make(d)) != x / d) ++bad;
  }
  std::printf("%%llu %%llu\n", n, bad);
  return bad != 0;
}
''' % os.path.join(ROOT, 'retinanet-examples_amd', 'csrc', 'fastdiv.hpp'))
    exe = tmp_path / 'fd'
    subprocess.run(['g++', '-O2', '-std=c++17', '-o', str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) > 3_000_000 and int(out[1]) == 0, out


def test_debug_trace_buffer_is_size_checked_on_the_host():
    """include/odtk_hip.h: odtk_debug_set_trace wants >= 128 KiB (select_decode's fine stamps start at word 8192).  The library cannot
    check a bare pointer; the binding does -- four probes had passed 64 KiB and one of them faulted the GPU at exit (round 6)."""
    import torch
    from odtk import _C
    assert _C.TRACE_WORDS * 8 == 128 * 1024
    for bad in (torch.zeros(8192, dtype=torch.int64), torch.zeros(_C.TRACE_WORDS, dtype=torch.int32), torch.zeros(_C.TRACE_WORDS, dtype=torch.int64)):
        with pytest.raises(ValueError):                      # too small | wrong dtype | not on the device: refused before any library call
            _C.debug_set_trace(bad)

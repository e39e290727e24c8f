#!/usr/bin/env python
"""Compile the reference's OWN rotated IoU / rotated NMS device code for the CPU (TEST INFRASTRUCTURE).

The reference's native library cannot be built here (nvcc, thrust, cub, TensorRT are absent), but the
part of csrc/cuda/nms_iou.cu that does the geometry -- Vector / Line / rotateLeft / IntersectionArea,
nms_rotate_kernel and iou_cuda_kernel -- and the axis-aligned nms_kernel of csrc/cuda/nms.cu are plain C++
once a dozen CUDA names exist.  This recipe reads
those line ranges from /root/reference WHERE THEY LIE, splices them between prelude.hpp and harness.cpp in
memory, pipes the translation unit to the compiler and builds oracle/_ref/libodtk_ref_native.so
with g++ -O2 -ffp-contract=off (IEEE reading of the source; the reference's nvcc build used --use_fast_math).

    python oracle/ref_build/build_ref.py          # no-op with a message when /root/reference is absent
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), '_ref')
REF = os.environ.get('ODTK_REFERENCE', '/root/reference')


def _between(lines, start_pat, stop_pat, what):
    """Lines from the first match of start_pat up to (excluding) the first later match of stop_pat."""
    start = next((i for i, l in enumerate(lines) if re.search(start_pat, l)), None)
    if start is None:
        raise RuntimeError('reference layout changed: no %s start (%s)' % (what, start_pat))
    stop = next((i for i in range(start + 1, len(lines)) if re.search(stop_pat, lines[i])), None)
    if stop is None:
        raise RuntimeError('reference layout changed: no %s end (%s)' % (what, stop_pat))
    return lines[start:stop]


def _lambda_body(path, what):
    """The body of the `[=] __device__ (int i) { ... }` gather lambda of a decode file (without its braces)."""
    lines = open(path).read().split('\n')
    body = _between(lines, r'\[=\] __device__ \(int i\) \{', r'^\s*\}\);', what)
    return body[1:]


def build(verbose=True):
    cu = os.path.join(REF, 'csrc', 'cuda', 'nms_iou.cu')
    uh = os.path.join(REF, 'csrc', 'cuda', 'utils.h')
    ax = os.path.join(REF, 'csrc', 'cuda', 'nms.cu')
    dec = os.path.join(REF, 'csrc', 'cuda', 'decode.cu')
    decr = os.path.join(REF, 'csrc', 'cuda', 'decode_rotate.cu')
    if not all(os.path.isfile(f) for f in (cu, uh, ax, dec, decr)):
        if verbose:
            print('[ref_build] %s not present: keeping whatever oracle/_ref already holds' % REF)
        return None
    cu_lines = open(cu).read().split('\n')
    uh_lines = open(uh).read().split('\n')
    float6 = _between(uh_lines, r'^struct float6', r'^template <typename T>', 'float6')            # utils.h:30-40
    device = _between(cu_lines, r'^constexpr int\s+kTPB', r'^int nms_rotate\(', 'device code')      # nms_iou.cu:41-258
    pairwise = _between(cu_lines, r'^__global__ void iou_cuda_kernel', r'^int iou\(', 'iou kernel')  # nms_iou.cu:324-375
    axis = _between(open(ax).read().split('\n'), r'^__global__ void nms_kernel', r'^int nms\(', 'nms kernel')   # nms.cu:44-80
    os.makedirs(OUT, exist_ok=True)
    # capture lists of the decode lambdas: same names and types as in the enclosing reference functions
    capt = ('(int i, size_t height, size_t width, size_t scale, size_t num_anchors, size_t num_classes, '
            'bool has_anchors, float *anchors_d, const float *in_scores, const float *in_boxes)')
    unit = '\n'.join([
        open(os.path.join(HERE, 'prelude.hpp')).read(),
        '\n'.join(float6), '\n'.join(device), '\n'.join(pairwise), '\n'.join(axis),
        'static odtk_ref_tuple<float4> decode_gather' + capt + ' {', '\n'.join(_lambda_body(dec, 'decode lambda')), '}',
        'static odtk_ref_tuple<float6> decode_rotate_gather' + capt + ' {', '\n'.join(_lambda_body(decr, 'decode_rotate lambda')), '}',
        '}  // namespace cuda', '}  // namespace odtk',
        open(os.path.join(HERE, 'harness.cpp')).read()])
    so = os.path.join(OUT, 'libodtk_ref_native.so')
    # the spliced translation unit goes to the compiler through a pipe: no copy of reference source is ever
    # written anywhere, oracle/_ref holds the binary only
    cmd = ['g++', '-O2', '-std=c++14', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared', '-w', '-x', 'c++', '-',
           '-o', so]
    subprocess.run(cmd, input=unit.encode(), check=True)
    if verbose:
        print('[ref_build] built %s from %s' % (so, cu))
    return so


if __name__ == '__main__':
    sys.exit(0 if build() or not os.path.isdir(REF) else 1)

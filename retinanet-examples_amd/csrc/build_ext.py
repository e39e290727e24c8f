#!/usr/bin/env python
"""Builds the compiled operator module retinanet-examples_amd/odtk/_C_ext*.so from csrc/odtk_binding.cpp: host C++ only
(g++), against PyTorch-ROCm's headers, linked to libodtk_hip.so next to it.  One explicit compiler command, in-tree
output (the .so travels to the GPU box with the snapshot; a JIT cache under ~/.cache would not)."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), 'odtk')
NAME = '_C_ext'


def target():
    return os.path.join(PKG, NAME + sysconfig.get_config_var('EXT_SUFFIX'))


def build(force=False):
    import torch
    from torch.utils import cpp_extension as ce
    out, src = target(), os.path.join(HERE, 'odtk_binding.cpp')
    deps = [src, os.path.join(HERE, '..', '..', 'include', 'odtk_hip.h')]
    if not force and os.path.isfile(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    inc = ce.include_paths(device_type='cuda') + [sysconfig.get_paths()['include']]
    libdirs = ce.library_paths(device_type='cuda')
    cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-fvisibility=hidden', src, '-o', out,
           '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-DTORCH_EXTENSION_NAME=' + NAME, '-DTORCH_API_INCLUDE_EXTENSION_H',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI), '-Wno-deprecated-declarations']
    cmd += ['-isystem' + p for p in inc]
    cmd += ['-L' + p for p in libdirs] + ['-L' + PKG]
    cmd += ['-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch_hip', '-ltorch', '-ltorch_python', '-lamdhip64', '-lodtk_hip',
            '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + libdirs[0]]
    print('[build_ext]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))

"""libodtk_conv.so without a GPU: it loads, exports every symbol include/odtk_conv.h declares, lists its instances and
validates arguments (no compute calls)."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'retinanet-examples_amd')]


def test_library_exports_what_the_header_declares():
    from odtk import _C
    if not os.path.isfile(_C._CONV_LIB_PATH):
        pytest.skip('libodtk_conv.so not built (no composable_kernel instance archive on this machine)')
    lib = ctypes.CDLL(_C._CONV_LIB_PATH)
    header = open(os.path.join(ROOT, 'include', 'odtk_conv.h')).read()
    declared = set(re.findall(r'\b(odtk_[a-z0-9_]+)\s*\(', header))
    assert declared == {'odtk_conv_bias_act', 'odtk_conv_bias_act_pads', 'odtk_conv_last_plan', 'odtk_conv_instance_count',
                        'odtk_conv_plan_export', 'odtk_conv_plan_import'}
    for name in declared:
        getattr(lib, name)
    assert _C.conv_available()
    clib = _C.conv_library()
    assert clib.odtk_conv_instance_count(_C._DTYPES[__import__('torch').bfloat16]) > 50
    assert clib.odtk_conv_instance_count(0) == 0                      # fp32: none
    # argument validation comes before any device work
    assert clib.odtk_conv_bias_act(None, None, None, None, 1, 8, 8, 8, 8, 3, 3, 1, 1, 1, 1, 1, 1, None) == _C.ERR_INVALID

"""CPU: libdmhomo_hip.so loads without a GPU and exports every symbol include/dmhomo_hip.h declares;
argument validation answers through the error channel (no compute is launched here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'dmhomo_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dmh_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    from dmhomo_amd import _lib
    names = header_functions()
    assert len(names) >= 30
    assert sorted(_lib.SIGNATURES) == names


def test_library_loads_and_exports_every_symbol():
    from dmhomo_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run `python -c "import __graft_entry__ as g; g.build()"` first'
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(h, name), name
    lib = _lib.lib()
    assert lib.dmh_version() == _lib.ABI_VERSION == 500


def test_pure_host_entry_points():
    from dmhomo_amd import _lib
    lib = _lib.lib()
    # default (fp16-piece kernel, 3x3 and 1x1): ceil(Cout/64) * chunks(32 ch) * taps * 2 k-steps * 4 fragments of 1 KB
    # (= 1024 floats per k-step) + 64 per-channel scales per 64-channel block; other DMH_CONV3_VARIANTs pack differently
    def f16x3(cout, c0, c1, taps):
        nt = (cout + 63) // 64
        return nt * ((c0 + 31) // 32 + (c1 + 31) // 32) * taps * 2 * 1024 + nt * 64
    import os
    if os.environ.get('DMH_CONV3_VARIANT', '9') == '9':
        assert lib.dmh_conv_pack_floats(64, 64, 0, 3, 3) == f16x3(64, 64, 0, 9)
        assert lib.dmh_conv_pack_floats(512, 512, 256, 3, 3) == f16x3(512, 512, 256, 9)
        assert lib.dmh_conv_pack_floats(384, 64, 0, 1, 1) == f16x3(384, 64, 0, 1)
        assert lib.dmh_conv_tiles(128, 128, 3, 1) == 128 and lib.dmh_conv_tiles(128, 128, 1, 1) == 128   # 8x16 stat tiles
    if os.environ.get('DMH_CONV3_VARIANT', '9') == '9':
        # 7x7 with <= 16 input channels: two taps per 32-channel K slice -> 25 tap pairs; wider inputs: 49 taps
        assert lib.dmh_conv_pack_floats(64, 12, 0, 7, 7) == f16x3(64, 12, 0, 25)
        assert lib.dmh_conv_pack_floats(64, 24, 0, 7, 7) == f16x3(64, 24, 0, 49)
    if os.environ.get('DMH_CONV3_VARIANT', '9') == '9':
        # Downsample 4x4 / stride 2 = 2x2 over the 4*C space-to-depth view when C % 32 == 0, else the fp32 image
        assert lib.dmh_conv_pack_floats(128, 64, 0, 4, 4) == f16x3(128, 256, 0, 4)
    assert lib.dmh_conv_pack_floats(128, 8, 0, 4, 4) == 2 * 1 * 16 * 64 * 16
    assert lib.dmh_conv_tiles(64, 64, 4, 2) == 32
    assert lib.dmh_linattn_splits(16384) == 128 and lib.dmh_linattn_splits(4) == 1
    assert lib.dmh_linattn_partial_floats(2, 256) == 2 * 2 * 4 * 1088


def test_bad_arguments_use_the_error_channel():
    from dmhomo_amd import _lib
    lib = _lib.lib()
    d = _lib.DmhConv()                                   # struct_size 0: a caller built against another header
    assert lib.dmh_conv2d(ctypes.byref(d), None) == -1
    assert b'struct_size' in lib.dmh_last_error()
    d = _lib.DmhConv(ctypes.sizeof(_lib.DmhConv))
    assert lib.dmh_conv2d(ctypes.byref(d), None) == -1
    assert b'dmh_conv2d: null pointer' in lib.dmh_last_error()
    assert lib.dmh_chan_layernorm(None, None, None, None, 4, 64, 1e-5, None, 0, None) == -1
    assert lib.dmh_linear(None, 0, None, None, None, 0, 1, 1, 1, 0, 0, None) == -1


def test_cpu_tensors_are_refused():
    import torch
    from dmhomo_amd import ops, _lib
    with pytest.raises(_lib.DmhError):
        ops.affine(torch.zeros(4), 1.0, 0.0)

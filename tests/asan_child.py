"""child process of tests/test_asan_host.py: drives the HOST side of the C ABI (libdmhomo_hip_asan.so: AddressSanitizer +
UBSan on the launch wrappers, packers and argument validators; device code not instrumented) through its argument
validation and packing entry points.  Runs without a GPU: a call that passes validation fails at the launch and answers
through the error channel — which is host code too.  Started with LD_PRELOAD=<asan runtime>; torch is NOT imported (a stub
stands in for the one attribute dmhomo_amd/_lib.py reads at import time).  Any sanitizer report aborts the process."""
import ctypes as C
import importlib.util
import itertools
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stub = types.ModuleType('torch')
stub.float32 = 'float32'
sys.modules['torch'] = stub
spec = importlib.util.spec_from_file_location('dmh_lib_binding', os.path.join(ROOT, 'dmhomo_amd', '_lib.py'))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)
assert L.LIB_PATH.endswith('libdmhomo_hip_asan.so'), L.LIB_PATH
lib = L.lib()
assert lib.dmh_version() == L.ABI_VERSION
assert 'libclang_rt.asan' in open('/proc/self/maps').read(), 'the sanitizer runtime is not loaded: start with LD_PRELOAD'

calls = errors = 0
HOSTBUF = (C.c_char * 65536)()            # valid HOST memory: a pointer argument the host side may parse (never a kernel's)


def args_for(argtypes, ival, pointers):
    out = []
    for t in argtypes:
        if t in (C.c_int, C.c_int32, C.c_int64):
            out.append(ival)
        elif t is C.c_float:
            out.append(1e-5)
        elif t is C.c_double:
            out.append(1.0)
        elif t is C.c_void_p or t is C.c_char_p:
            out.append(C.cast(HOSTBUF, C.c_void_p) if pointers else None)
        elif hasattr(t, '_type_') and isinstance(t._type_, type) and issubclass(t._type_, C.Structure):
            s = t._type_()
            if pointers:
                if hasattr(s, 'struct_size'):
                    s.struct_size = C.sizeof(s)
                for name, ft in s._fields_:
                    if ft is C.c_void_p:
                        setattr(s, name, C.cast(HOSTBUF, C.c_void_p).value)
                    elif ft is C.c_int32:
                        setattr(s, name, ival)
            out.append(C.byref(s))
        elif hasattr(t, '_type_') and t._type_ in (C.c_int64, C.c_int32):     # host arrays of sizes / strides (dmh_bgemm)
            arr = (t._type_ * 8)(*([max(ival, 0)] * 8))
            out.append(arr if pointers else None)
        else:
            raise TypeError(t)
    return out


def sweep():
    """every entry point: NULL pointers with sizes 0 / 1 / 64 / -1 / a huge one, then host pointers with small sizes (passes
    most validators: the wrappers then compute grids, tables and workspace sizes, and fail at the launch: no GPU here)"""
    global calls, errors
    for name, (res, argtypes) in sorted(L.SIGNATURES.items()):
        fn = getattr(lib, name)
        for ival, pointers in itertools.product((0, 1, 4, 64, -1, 2 ** 30), (False, True)):
            if pointers and ival in (-1, 2 ** 30):
                continue                   # (sizes the validators must refuse are covered with NULL pointers)
            rc = fn(*args_for(argtypes, ival, pointers))
            calls += 1
            if res is C.c_int and name not in PURE and rc != 0:
                errors += 1
                assert lib.dmh_last_error(), name


# int-returning entry points that are pure host functions (sizes / counts), not status codes
PURE = {'dmh_version', 'dmh_conv_tiles', 'dmh_linattn_splits', 'dmh_linattn_fused_splits'}


def conv_struct_cases():
    """DmhConv as a caller built against another header would hand it over: wrong struct_size (0, one field short, too large),
    NULL members, absurd geometry, every kernel size / stride / upsample flag; the packers over the same range"""
    global calls
    for size in (0, 8, C.sizeof(L.DmhConv) - 8, C.sizeof(L.DmhConv) + 8, C.sizeof(L.DmhConv)):
        d = L.DmhConv(size)
        assert lib.dmh_conv2d(C.byref(d), None) == -1 and lib.dmh_last_error()
        calls += 1
    hp = C.cast(HOSTBUF, C.c_void_p).value
    for kh, stride, ups, c0, c1, cout, hw, B in itertools.product((1, 2, 3, 4, 7, 5), (1, 2), (0, 1, 2), (4, 12, 64, 96),
                                                                  (0, 64), (6, 64, 128), (1, 16, 130), (0, 1, 3)):
        d = L.DmhConv(C.sizeof(L.DmhConv), hp, hp if c1 else None, hp, hp, None, None, None, hp, hp, B, hw, hw, c0, c1, cout, kh,
                      kh, stride, ups, None, 0, 0, None, None, None, None, 1e-5, None)
        lib.dmh_conv2d(C.byref(d), None)   # refused or failing at the launch: either way through the error channel
        lib.dmh_conv_pack_floats(cout, c0, c1, kh, kh)
        lib.dmh_conv_tiles(hw, hw, kh, stride)
        lib.dmh_conv_up2_pack_floats(cout, c0)
        calls += 4
    # optional members with inconsistent companions: fin_n without fin_w, in_bound without in_coef, pix_stats on a 3x3
    d = L.DmhConv(C.sizeof(L.DmhConv), hp, None, hp, hp, None, None, None, hp, None, 2, 16, 16, 64, 0, 64, 3, 3, 1, 0,
                  hp, 8, 6, None, None, hp, hp, 1e-5, hp)
    lib.dmh_conv2d(C.byref(d), None)
    calls += 1


def pack_multi_cases():
    """dmh_pack_conv_weights_multi parses an ARRAY of caller structs into launch tables of 32: 0 / 1 / 33 / 100 jobs, odd kernel
    sizes, NULL members"""
    global calls
    hp = C.cast(HOSTBUF, C.c_void_p).value
    for n in (1, 31, 32, 33, 100):
        jobs = (L.DmhPackJob * n)()
        for i, j in enumerate(jobs):
            j.src, j.ws, j.wpack = hp, (hp if i % 2 else None), hp
            j.Cout, j.C0, j.C1, j.KH, j.transposed = 64 + 64 * (i % 3), 32 * (1 + i % 4), 64 * (i % 2), 3 if i % 5 else 1, i % 2
        lib.dmh_pack_conv_weights_multi(C.cast(jobs, C.c_void_p), n, 1e-5, None)
        calls += 1
    jobs = (L.DmhPackJob * 2)()
    for bad in (dict(KH=5), dict(Cout=0), dict(C0=0), dict(C1=-1), dict(src=None), dict(wpack=None)):
        for j in jobs:
            j.src, j.ws, j.wpack, j.Cout, j.C0, j.C1, j.KH, j.transposed = hp, None, hp, 64, 64, 0, 3, 0
        for k, v in bad.items():
            setattr(jobs[1], k, v)
        assert lib.dmh_pack_conv_weights_multi(C.cast(jobs, C.c_void_p), 2, 1e-5, None) != 0, bad
        calls += 1
    assert lib.dmh_pack_conv_weights_multi(None, 3, 1e-5, None) != 0 and lib.dmh_pack_conv_weights_multi(C.cast(jobs, C.c_void_p), 0, 1e-5, None) != 0


def step_struct_cases():
    """DmhStep by value / by pointer with every objective / mode, in and out of range"""
    global calls
    hp = C.cast(HOSTBUF, C.c_void_p)
    for name, (res, argtypes) in L.SIGNATURES.items():
        if not any(hasattr(t, '_type_') and t._type_ is L.DmhStep for t in argtypes if not isinstance(t, type) or True):
            continue
        for obj, mode, clip in itertools.product((-1, 0, 1, 2, 3, 99), (-1, 0, 1, 2, 3), (0, 1)):
            a = []
            for t in argtypes:
                if hasattr(t, '_type_') and t._type_ is L.DmhStep:
                    a.append(C.byref(L.DmhStep(obj, clip, mode, 3.0, 1., 1., 1., 1., 0., 0., 0.)))
                elif t in (C.c_int, C.c_int64):
                    a.append(4)
                elif t is C.c_float:
                    a.append(1.0)
                else:
                    a.append(hp)
            getattr(lib, name)(*a)
            calls += 1


sweep()
conv_struct_cases()
pack_multi_cases()
step_struct_cases()
print(f'asan child ok: {calls} calls through {len(L.SIGNATURES)} entry points, {errors} answered through the error channel')

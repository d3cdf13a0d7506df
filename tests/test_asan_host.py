"""CPU: the HOST side of the C ABI under AddressSanitizer + UBSan ("race detection / sanitizers").

`make -C dmhomo_amd/csrc asan` builds libdmhomo_hip_asan.so: the launch wrappers, weight packers and argument validators of
csrc/*.hip and lib.cpp — host code that parses caller structs (DmhConv.struct_size, DmhPackJob arrays, DmhStep, size and
count arguments) — instrumented; device code is not (GPU ASan / XNACK are not available on the pool).  tests/asan_child.py
then drives every entry point of include/dmhomo_hip.h through its validation and packing paths in a child process started
with LD_PRELOAD=<the toolchain's asan runtime> (no torch, no GPU): NULL / undersized / oversized structs, sizes 0, -1 and
2^30, job arrays across the 32-entry launch-table boundary, every DmhStep objective / mode in and out of range.  Any report
aborts the child.  Round 6's first run found and fixed: int64 / int overflow in five dmh_*_floats size functions on absurd
dimensions (now answered with -1: common.h dmh_dims_ok) and a NULL dereference in dmh_multi_blocks."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc: the sanitizer build cannot be made here')
def test_host_side_of_the_c_abi_under_asan_ubsan():
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'dmhomo_amd', 'csrc'), 'asan', '-j8'], capture_output=True, text=True,
                       env=dict(os.environ, HIPCC=HIPCC), timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lib = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_asan.so')
    rt = subprocess.run([HIPCC, '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(lib) and os.path.exists(rt), (lib, rt)
    env = {k: v for k, v in os.environ.items() if k != 'DMH_CONV3_VARIANT'}
    env.update(LD_PRELOAD=rt, DMH_LIB_PATH=lib, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    c = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'asan_child.py')], capture_output=True, text=True, env=env,
                       timeout=900)
    assert c.returncode == 0 and 'asan child ok' in c.stdout, c.stdout[-1500:] + c.stderr[-6000:]
    assert 'runtime error' not in c.stderr and 'AddressSanitizer' not in c.stderr, c.stderr[-6000:]
    print(c.stdout.strip())

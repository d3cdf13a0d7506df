"""diagnostic: time the Winograd kernel with phases ablated (needs `make -C dmhomo_amd/csrc stamps`)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    os.environ.setdefault('DMH_CONV3_VARIANT', '6')
    from dmhomo_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
    import torch
    from dmhomo_amd import ops
    dev = torch.device('cuda', 0)
    for (C, H) in ((64, 128), (128, 64), (512, 16)):
        w = torch.randn((C, C, 3, 3), device=dev) * 0.04
        pc = ops.PackedConv(w, torch.randn(C, device=dev), C)
        x = torch.randn((50, H, H, C), device=dev)
        for _ in range(3):
            ops.conv2d(pc, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            ops.conv2d(pc, x)
        e1.record(); torch.cuda.synchronize()
        print(f'  {C}->{C}@{H}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us')
elif os.environ.get('DMH_WBX_SWEEP'):
    for abl, name in ((0, 'full'), (1, 'weights loaded once'), (2, 'no input transform'), (4, 'no staging'), (6, 'no staging, no transform'),
                      (8, 'no matrix work'), (16, 'no epilogue'), (7, 'matrix work + epilogue only'), (23, 'matrix work only (MFMA + split + A reads)'),
                      (14, 'epilogue only'), (30, 'empty')):
        print(f'wino bf16x3 ABL={abl} ({name})', flush=True)
        subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, DMH_BX_ABL=str(abl), DMH_CONV3_VARIANT='8'))
elif os.environ.get('DMH_BX_SWEEP'):
    for abl, name in (('0', 'full'), ('4', 'three terms only'), ('7', 'three terms, no operand traffic')):
        print(f'bf16x3 ABL={abl} ({name})', flush=True)
        subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, DMH_BX_ABL=abl, DMH_CONV3_VARIANT='7'))
else:
    v7 = os.environ.get('DMH_CONV3_VARIANT') in ('7', '9')
    for wide in (('1',) if v7 else ('0', '1')):
        for abl, name in (('0', 'full'), ('1', 'no staging/transform'), ('2', 'no matrix phase'), ('3', 'neither (loads+epilogue only)')) + \
                ((('4', 'no epilogue'), ('5', 'matrix phase only'), ('6', 'staging only'), ('7', 'input loads only')) if v7 else ()):
            print(f'wide={wide} ablate={abl} ({name})', flush=True)
            env = dict(os.environ, DMH_WINO_ABLATE=abl, DMH_WINO_WIDE=wide)
            subprocess.run([sys.executable, __file__, 'child'], env=env)

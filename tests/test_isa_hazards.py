"""CPU: scan the gfx950 assembly hipcc generates for the kernels that contain inline-asm vector instructions (the fp16 split
of common.h, used by conv_f16x3.hip / linattn_fused.hip / conv_backward.hip) for the hazards hipcc's own recognizer cannot
see because one side sits inside an asm statement — MFMA result read or overwritten by asm too early, asm result fed to an
MFMA without wait states, a transcendental result forwarded straight into asm (tools/hazard_scan.py).  This class of fault
computed a few wrong pixels per launch, under load only, twice in round 2; tests/test_gpu_soak.py is the
run-time guard, this is the build-time one."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.mark.parametrize('src', ['linattn_fused.hip', 'conv_f16x3.hip', 'conv_backward.hip'])
def test_no_inline_asm_hazards_in_generated_isa(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not present')
    out = str(tmp_path / (src + '.s'))
    subprocess.run([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only',
                    os.path.join(ROOT, 'dmhomo_amd', 'csrc', src), '-o', out], check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'hazard_scan.py'), out], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith('findings: 0'), r.stdout[-3000:] + r.stderr[-1000:]

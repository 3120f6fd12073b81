"""The hand-written DPP blocks of csrc/als_kernels.h (v_fmac_f32_dpp in factor_diag) rely on a gfx9 data
hazard rule the compiler does not apply to inline asm: a DPP source register needs two wait states after
a VALU write.  This compiles the device code to ISA (hipcc cross-compiles without a GPU) and checks
every DPP instruction with tools/check_dpp_hazards.py."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_dpp_read_after_valu_write(tmp_path):
    out = tmp_path / "mals_api.s"
    src = os.path.join(ROOT, "myrrix-recommender_amd", "csrc", "mals_api.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "v_fmac_f32_dpp" in out.read_text()

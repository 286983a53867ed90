"""Device helpers that the schedule kernels lean on, checked on their own against host loops
(tests/device/wave_helpers_check.hip, compiled here with hipcc): the wave-level sums and scans of csrc/kvc_common.h --
the shuffle forms and the DPP forms for full waves -- and rank_in_halves of csrc/kvc_schedule_fused.h (two lists per
wave ranked through DPP row broadcasts) for every pair of list lengths."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_sums_and_scans(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "wave_helpers_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value",
                           os.path.join(REPO, "tests", "device", "wave_helpers_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "WAVE_HELPERS_OK" in out.stdout, out.stdout + out.stderr

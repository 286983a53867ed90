"""The drop-in boundary without Python in the loop: a C++ host (tests/cabi/cabi_host.cpp)
links libkvc_mi355x.so by its header only, runs count -> moves -> compaction and compares
with the C oracle, two decode steps of aggregation + eviction schedule through the harvest
protocol of ABI version 5 (lists made by the aggregation pass against the schedule's own pass), then
the decode attention through its parameter struct against a plain float loop."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_through_the_c_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "cabi_host")
    libdir = os.path.join(REPO, "vllm_kvcompress_amd")
    host_o = str(tmp_path / "cabi_host.o")
    orc_o = str(tmp_path / "kvc_oracle.o")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-c", os.path.join(REPO, "oracle", "kvc_oracle.c"), "-o", orc_o])
    gomp = subprocess.check_output(["gcc", "-print-file-name=libgomp.so"], text=True).strip()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-c",
                           os.path.join(REPO, "tests", "cabi", "cabi_host.cpp"), "-I",
                           os.path.join(REPO, "include"), "-o", host_o])
    subprocess.check_call([hipcc, host_o, orc_o, gomp, "-L", libdir, "-lkvc_mi355x",
                           f"-Wl,-rpath,{libdir}", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "CABI_OK" in out.stdout, out.stdout + out.stderr

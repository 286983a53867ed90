"""The drop-in boundary as the fork reaches it: torch.ops._C_kvc_ops.* / _C_cache_ops.* / _C.* --
once bound by the compiled library libkvc_torch.so (C++ kernels, TORCH_LIBRARY_IMPL, the shape
of the reference's csrc/torch_bindings.cpp:353-418), once by the Python registration.  A process
holds one binding, so each runs in a child (tests/dispatch_driver.py) that checks all six ops
against the oracle."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("binding", ["compiled", "python"])
def test_all_ops_under_each_binding(binding):
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dispatch_driver.py"), binding],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DISPATCH_RESULT ")][-1]
    res = json.loads(line[len("DISPATCH_RESULT "):])
    assert res["binding"] == binding and res["ops_ok"] == 7
    # the compiled binding's kernels are registered from C++: no Python frame in the dispatch
    assert res["registered_from"] == ("kvc_torch_binding.cpp" if binding == "compiled" else "python")


def test_writes_of_the_compiled_binding_are_seen():
    """libkvc_torch.so's kernels write through raw pointers; they bump the version counter of every tensor they
    write, so that harvested lists, the tracked move table and remembered plans made before such a write are not
    trusted after it (tests/compiled_writes_driver.py: four scenarios, each against the oracle)"""
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "compiled_writes_driver.py")],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("COMPILED_WRITES ")][-1]
    res = json.loads(line[len("COMPILED_WRITES "):])
    assert set(res) == {"harvest_then_reshape_and_cache", "harvest_then_execute_cache_moves", "tracked_table_and_plan",
                        "outputs_are_counted"}

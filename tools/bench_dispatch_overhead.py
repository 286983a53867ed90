#!/usr/bin/env python3
"""Host cost per dispatcher call, compiled binding (libkvc_torch.so) vs Python registration
(ctypes), on tiny inputs where the device is never the bottleneck.  Matters where the step is
launch-bound: the 16-sequence continual steady state (0.41 ms per step, ~30 launches)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {}
for binding in ("compiled", "python"):
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dispatch_driver.py"), binding, "--time"],
                         capture_output=True, text=True, timeout=900, cwd=REPO)
    if out.returncode != 0:
        res[binding] = {"error": out.stderr[-1500:]}
        continue
    line = [l for l in out.stdout.splitlines() if l.startswith("DISPATCH_RESULT ")][-1]
    res[binding] = json.loads(line[len("DISPATCH_RESULT "):])
print(json.dumps(res, indent=1))

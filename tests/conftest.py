import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR)
                  if f.endswith(".npz") and not f.startswith(("blockstate_", "append_", "prefill_alloc_", "agg_", "policy_", "attn_")))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR

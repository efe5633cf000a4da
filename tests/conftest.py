import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built artefacts are git-ignored and normally travel with the tree; build them if a
    # bare checkout is being tested (hipcc cross-compiles gfx950 without a GPU).
    import subprocess
    needed = [os.path.join(ROOT, "umi_collapse_rs_amd", "libumihip.so"),
              os.path.join(ROOT, "oracle", "libumi_oracle.so"),
              os.path.join(ROOT, "umi_collapse_rs_amd", "bin", "umicollapse"),
              os.path.join(ROOT, "build", "test_host")]
    if not all(os.path.exists(p) for p in needed):
        subprocess.check_call(["make", "-s", "-j4", "-C", ROOT, "all"])


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return json.load(f)

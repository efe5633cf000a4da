"""C++ host mirror of the Algorithm/DataStruct interface (umi_collapse.hpp), compiled by
`make cpptest` and run against the oracle on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror_matches_oracle():
    exe = os.path.join(ROOT, "build", "test_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "cpptest"])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join(
        [os.path.join(ROOT, "umi_collapse_rs_amd"), os.path.join(ROOT, "oracle"),
         env.get("LD_LIBRARY_PATH", "")])
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp host mirror ok" in out.stdout


def test_cpp_host_header_compiles():
    subprocess.check_call(["make", "-s", "-C", ROOT, "cpptest"])
    assert os.path.exists(os.path.join(ROOT, "build", "test_host"))

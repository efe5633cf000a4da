"""The host planner under AddressSanitizer + UBSan (CPU build only: the GPU pool has no sanitizer
runs).  tests/cpp/test_plan.cpp drives umihip_plan.hpp -- build_plan, the tile / segment / range
tables, the multi-GPU partition -- over random bucket tables and checks that every pair of every
bucket is covered exactly once and nothing touches memory outside its vectors."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planner_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "test_plan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_plan.cpp")])
    r = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "plan ok" in r.stdout

"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer build of the C ABI's HOST logic (GPU sanitizers are not available on
this pool): the parameter layout and packed-stream gather maps of csrc/eonerf_pack.cpp and the workspace carving of csrc/eonerf_carve.h,
driven by tests/host/host_checks.cpp.  Compiled host-only with the ROCm clang (the sources share headers with the device code)."""
import os
import subprocess

from conftest import REPO

HIPCC = "/opt/rocm/bin/hipcc"


def test_pack_and_carve_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_checks")
    cmd = [HIPCC, "-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", os.path.join(REPO, "tests", "host", "host_checks.cpp"),
           os.path.join(REPO, "eonerf_code_amd", "csrc", "eonerf_pack.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host checks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]

"""The host runtime behind the C ABI (pockit_amd/csrc/pk_runtime.cpp: pinned landing blocks, double-buffered staging,
prepared-x protocol, copy batching, polling waits, speculative Hessian, constant Jacobian runs, CSR run tables) built with
``-fsanitize=address,undefined`` against a host-only stand-in of the HIP runtime (tests/fake_hip: deferred streams, "kernels"
that write recomputable values) and driven through its protocols by tests/fake_hip/driver.cpp.  CPU only -- GPU sanitizers are
not available on this pool.  SURVEY.md section 5 "race detection / sanitizers"."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_hip")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_runtime_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    exe = str(tmp_path / "pk_runtime_sanitized")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-fno-sanitize-recover=undefined", "-I", FAKE, "-I", ROOT,
           os.path.join(ROOT, "pockit_amd", "csrc", "pk_runtime.cpp"), os.path.join(FAKE, "fake_hip.cpp"),
           os.path.join(FAKE, "driver.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
    assert "checks passed" in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_helper_threads_under_sanitizers(tmp_path, sanitizer):
    """pk_host_threads / pk_same_bits / pk_copy_bits (the helper threads that take slices of rank 0's passes over x and
    lambda in the host-landed sharded cycle): ThreadSanitizer and AddressSanitizer builds of the same source, driven by
    tests/fake_hip/pool_driver.cpp through hot and cold helpers, every pass length class, differences at either end."""
    exe = str(tmp_path / "pk_pool_sanitized")
    cmd = ["g++", "-std=c++17", "-g", "-O1", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-pthread", "-I", FAKE, "-I", ROOT,
           os.path.join(ROOT, "pockit_amd", "csrc", "pk_runtime.cpp"), os.path.join(FAKE, "fake_hip.cpp"),
           os.path.join(FAKE, "pool_driver.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    if sanitizer == "thread" and run.returncode != 0 and "unexpected memory mapping" in run.stderr:
        # (ThreadSanitizer cannot place its shadow under this kernel's address-space randomisation: once more without it)
        if shutil.which("setarch") is None:
            pytest.skip("ThreadSanitizer cannot start on this kernel (unexpected memory mapping) and setarch is missing")
        run = subprocess.run(["setarch", "-R", exe], capture_output=True, text=True, env=env, timeout=600)
        if run.returncode != 0 and ("unexpected memory mapping" in run.stderr or "setarch" in run.stderr):
            pytest.skip("ThreadSanitizer cannot start on this kernel (unexpected memory mapping)")
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
    assert "checks passed" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr and "ERROR" not in run.stderr

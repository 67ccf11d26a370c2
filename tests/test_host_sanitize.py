"""AddressSanitizer + UndefinedBehaviorSanitizer build of the host side of librsba (the reference has no
sanitizer runs; this project's host code gets one).  ba_problem.cpp, rsba_capi.cpp and ba_initial_guess.cpp are compiled
with -fsanitize=address,undefined together with tests/host_sanitize_driver.cpp, which walks the file readers (committed
fixtures, short / malformed / out-of-range inputs), the accessors one past either end, the writers and the front end's pose
algebra / EPnP.  GPU sanitizers are not available on the pool; the device side is not in this build."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "realsensecalibration_amd", "csrc")


def test_host_code_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize_driver")
    srcs = [os.path.join(ROOT, "tests", "host_sanitize_driver.cpp")] + [os.path.join(CSRC, f) for f in ("ba_problem.cpp", "rsba_capi.cpp", "ba_initial_guess.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-Wall", "-I", os.path.join(ROOT, "include")] + srcs + ["-o", exe])
    scratch = tmp_path / "out"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden"), str(scratch)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host sanitize driver: ok" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]

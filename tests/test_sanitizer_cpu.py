"""Host side of libpaa_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer ("race detection /
sanitizers"): the library is rebuilt with -fsanitize=address,undefined for the host code (-fno-gpu-sanitize: the device
code is checked by the full-matrix parity tests instead) and the CPU-side ABI tests -- table builders, FFT plans,
shape helpers, error paths -- run against that build in a child interpreter with the ASan runtime preloaded."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime(hipcc):
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang")
    cands = []
    if os.path.exists(clang):
        res = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
        cands.append(res.stdout.strip())
    cands += glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    for c in cands:
        if c and os.path.isabs(c) and os.path.exists(c):
            return c
    return None


def test_cpu_side_abi_tests_pass_under_asan_and_ubsan(tmp_path):
    sys.path.insert(0, ROOT)
    from pyaudioanalysis_amd import _build
    try:
        hipcc = _build.hipcc_path()
    except RuntimeError:
        pytest.skip("no hipcc on this host")
    rt = _asan_runtime(hipcc)
    if rt is None:
        pytest.skip("no ASan runtime next to hipcc's clang")
    lib = str(tmp_path / "libpaa_hip_asan.so")
    # every translation unit of the library (csrc/paa_lib.hip + the family_*.hip units) with the host sanitizers on
    try:
        _build.build_to(lib, extra_flags=["-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-gpu-sanitize"],
                        opt="-O1")
    except RuntimeError as exc:
        pytest.fail(str(exc)[-4000:])
    env = dict(os.environ)
    env.update({"LD_PRELOAD": rt, "PAA_HIP_LIBRARY": lib,
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1",            # (the interpreter itself "leaks")
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi_cpu.py"), "-q", "-x",
                          "-p", "no:cacheprovider", "-k", "not c_client"], capture_output=True, text=True, env=env,
                         cwd=ROOT, timeout=900)
    tail = (run.stdout + run.stderr)[-4000:]
    assert run.returncode == 0, tail
    assert "passed" in run.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail

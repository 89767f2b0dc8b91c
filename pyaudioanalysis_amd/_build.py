"""Build libpaa_hip.so in-tree with hipcc for gfx950 (no PyTorch, no JIT cache)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpaa_hip.so")
SOURCES = ["paa_lib.hip"]
HEADERS = ["device_common.hpp", "kernels_generic.hpp", "kernels_fast.hpp", "kernels_aux.hpp",
           "kernels_tail.hpp", "kernels_big.hpp", "kernels_ct.hpp", "kernels_mix.hpp", "kernels_tri.hpp", "kernels_sim.hpp", "kernels_reg.hpp", "kernels_svm.hpp", "comm_rccl.hpp", "tables.hpp", os.path.join("..", "..", "include", "paa_hip.h")]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libpaa_hip.so for gfx950)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile csrc/paa_lib.hip -> libpaa_hip.so (gfx950 only).  Returns the library path."""
    if not force and not is_stale():
        return LIB
    # -disable-machine-licm: the feature kernels' loop bodies are thousands of instructions long; hoisting every FP64
    # literal and per-lane LDS address out of them creates >100 loop-invariant registers that then spill (AGPR copies at
    # one wave per SIMD, scratch at two).  Re-materialising a literal at its use costs two s_mov / v_mov.
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-disable-machine-licm",
           "-I/opt/rocm/include", os.path.join(CSRC, "paa_lib.hip"), "-o", LIB + ".tmp", "-ldl"]
    cmd += os.environ.get("PAA_HIPCC_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

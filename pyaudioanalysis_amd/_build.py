"""Build libpaa_hip.so in-tree with hipcc for gfx950 (no PyTorch, no JIT cache).

The library is several translation units: csrc/paa_lib.hip (host side: state, plans, dispatch, host API, RCCL, debug exports
-- itself made of the lib_*.hpp units -- plus the small kernels) and one family_*.hip per feature-kernel family, whose
kernels are instantiated there and nowhere else.  The units compile in parallel and are linked into one shared object."""
import concurrent.futures
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpaa_hip.so")
SOURCES = ["paa_lib.hip", "family_fast.hip", "family_ct.hip", "family_tri_a.hip", "family_tri_b.hip", "family_tri_c.hip", "family_reg_mix_generic.hip", "family_blu.hip", "family_wgr.hip", "family_wgs.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "paa_hip.h")]
# -disable-machine-licm: the feature kernels' loop bodies are thousands of instructions long; hoisting every FP64 literal
# and per-lane LDS address out of them creates >100 loop-invariant registers that then spill (AGPR copies at one wave per
# SIMD, scratch at two).  Re-materialising a literal at its use costs two s_mov / v_mov.
BASE_FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-fPIC", "-mllvm", "-disable-machine-licm", "-I/opt/rocm/include"]
# Per-unit code generation switches (round 6, A/B on one box: scripts/rounds/r06/gpu_r06i.sh, profiles/r06_ab_load_store_opt.txt):
# * -load-store-opt (target feature off): the SI load/store optimizer pairs LDS accesses into ds_read2_b64 / ds_write2_b64, which cost
#   twice the LDS-array cycles of two single accesses on gfx950 (MI355X_MICROARCH.md, LDS table): headline kernel 0.2606 -> 0.2531 ms,
#   1920 / 2205 / 1102 +2.5 %; the power-of-two three-pass unit was 1.5 % slower with it and keeps the default;
# * -amdgpu-load-store-vectorizer=0 on top (no <2 x double> LDS loads, i.e. no ds_read2_b64 from the IR either): the headline kernel
#   0.2477 ms; every other family lost (Bluestein -19 %: its global loads are no longer merged), so only family_fast.hip has it.
NO_LSO = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
UNIT_FLAGS = {
    "family_fast.hip": NO_LSO + ["-mllvm", "-amdgpu-load-store-vectorizer=0"],
    "family_ct.hip": NO_LSO, "family_tri_a.hip": NO_LSO, "family_tri_b.hip": NO_LSO, "family_reg_mix_generic.hip": NO_LSO,
    "family_blu.hip": NO_LSO,
}


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libpaa_hip.so for gfx950)")


def is_stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_to(lib, extra_flags=(), opt="-O3", verbose=False, jobs=None):
    """Compile every translation unit (in parallel) and link them into `lib`.  extra_flags go to compile AND link."""
    hipcc = hipcc_path()
    extra = list(extra_flags) + os.environ.get("PAA_HIPCC_FLAGS", "").split()
    jobs = jobs or min(len(SOURCES), os.cpu_count() or 1)
    with tempfile.TemporaryDirectory(prefix="paa_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            cmd = [hipcc] + BASE_FLAGS + UNIT_FLAGS.get(src, []) + [opt] + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [f for f in extra if f.startswith("-fsanitize") or f == "-g"] \
            + objs + ["-o", lib + ".tmp", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("linking %s failed:\n%s%s" % (lib, res.stdout, res.stderr))
    os.replace(lib + ".tmp", lib)
    return lib


def build(force=False, verbose=False):
    """Build pyaudioanalysis_amd/libpaa_hip.so (gfx950 only) when it is missing or older than its sources.  Returns the path."""
    if not force and not is_stale():
        return LIB
    return build_to(LIB, verbose=verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))

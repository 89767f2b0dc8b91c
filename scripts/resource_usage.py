#!/usr/bin/env python3
"""Register / scratch / LDS footprint of every kernel IN THE SHIPPED libpaa_hip.so (no GPU needed).

Reads the gfx950 code objects out of the library's .hip_fatbin section (one clang offload bundle per translation unit),
parses the NT_AMDGPU_METADATA note of each (msgpack) and prints / writes, per kernel: VGPRs, AGPRs, SGPRs, spilled
registers, private segment (scratch) bytes per lane, static LDS bytes, and the waves per SIMD the register budget allows
(512 registers per lane and SIMD on gfx950, allocation granule 8; the LDS footprint of the feature kernels is dynamic
and decides their occupancy separately -- DESIGN 4).

    python scripts/resource_usage.py                      # table on stdout
    python scripts/resource_usage.py --json profiles/r05_resource_usage.json

tests/test_abi_cpu.py::test_shipped_kernels_hold_their_register_budget asserts the claims DESIGN makes on these numbers.
"""
import argparse
import json
import os
import shutil
import struct
import subprocess
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pyaudioanalysis_amd", "libpaa_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
CXXFILT = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or ""


def code_objects(path):
    """Yield the device ELF images bundled in a host shared object."""
    blob = open(path, "rb").read()
    pos = blob.find(MAGIC)
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + len(MAGIC))


def elf_notes(image):
    """(name, type, desc) of every note of a little-endian ELF64 image."""
    assert image[:4] == b"\x7fELF" and image[4] == 2, "not an ELF64 image"
    shoff, = struct.unpack_from("<Q", image, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", image, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", image, sh + 4)
        if sh_type != 7:                      # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", image, sh + 0x18)
        cur, end = off, off + size
        while cur + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", image, cur)
            cur += 12
            name = image[cur:cur + namesz].rstrip(b"\0").decode()
            cur += (namesz + 3) & ~3
            desc = image[cur:cur + descsz]
            cur += (descsz + 3) & ~3
            yield name, ntype, desc


def demangle(names):
    if not names or not CXXFILT:
        return list(names)
    res = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True)
    out = res.stdout.split("\n")
    return out[:len(names)] if res.returncode == 0 and len(out) >= len(names) else list(names)


def short_name(demangled):
    """'void paa::f800::st_fast_800_kernel<400, 0, 1, 8>(paa::PlanDev, ...)' -> 'f800::st_fast_800_kernel<400, 0, 1, 8>'."""
    depth, cut = 0, len(demangled)
    for i, ch in enumerate(demangled):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    name = demangled[:cut]
    if name.startswith("void "):
        name = name[5:]
    return name.replace("paa::", "")


def waves_per_simd(vgpr, agpr):
    """gfx950: 512 unified registers per lane and SIMD, allocation granule 8, at most 8 waves.  The metadata's
    .vgpr_count is the unified total: it already contains the AGPRs (438 = 256 + 182 for the full st_mix instance)."""
    total = max(8, (max(vgpr, agpr) + 7) & ~7)
    return max(1, min(8, 512 // total))


def kernels_of(path):
    rows = []
    for image in code_objects(path):
        for name, ntype, desc in elf_notes(image):
            if name != "AMDGPU" or ntype != 32:          # NT_AMDGPU_METADATA
                continue
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in meta.get("amdhsa.kernels", []):
                rows.append({
                    "symbol": k[".name"],
                    "vgpr": k.get(".vgpr_count", 0),
                    "agpr": k.get(".agpr_count", 0),
                    "sgpr": k.get(".sgpr_count", 0),
                    "vgpr_spill": k.get(".vgpr_spill_count", 0),
                    "sgpr_spill": k.get(".sgpr_spill_count", 0),
                    "scratch_bytes_per_lane": k.get(".private_segment_fixed_size", 0),
                    "static_lds_bytes": k.get(".group_segment_fixed_size", 0),
                    "max_workgroup": k.get(".max_flat_workgroup_size", 0),
                })
    for row, nice in zip(rows, demangle([r["symbol"] for r in rows])):
        row["kernel"] = short_name(nice)
        row["waves_per_simd_by_registers"] = waves_per_simd(row["vgpr"], row["agpr"])
    rows.sort(key=lambda r: r["kernel"])
    return rows


def compiler_id():
    """hipcc / LLVM version + PAA_HIPCC_FLAGS of this host (the register counts are a function of them)"""
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout.splitlines()
    except Exception:
        return None
    keep = [ln.strip() for ln in out if ln.startswith("HIP version") or "clang version" in ln]
    return " | ".join(keep + ["PAA_HIPCC_FLAGS=" + os.environ.get("PAA_HIPCC_FLAGS", "")]) if keep else None


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--json", help="write the table to this file")
    ap.add_argument("--filter", default="", help="only kernels whose demangled name contains this")
    args = ap.parse_args()
    rows = [r for r in kernels_of(args.lib) if args.filter in r["kernel"]]
    if args.json:
        summary = {
            "library": os.path.relpath(args.lib, ROOT),
            "library_bytes": os.path.getsize(args.lib),
            "source": "NT_AMDGPU_METADATA notes of the gfx950 code objects inside the shipped library (scripts/resource_usage.py)",
            "kernels": len(rows),
            "compiler": compiler_id(),
            "kernels_with_scratch": sorted(r["kernel"] for r in rows if r["scratch_bytes_per_lane"]),
            "kernels_with_agprs": sorted(r["kernel"] for r in rows if r["agpr"]),
            "kernels_with_vgpr_spills": sorted(r["kernel"] for r in rows if r["vgpr_spill"]),
            "table": [{k: r[k] for k in ("kernel", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes_per_lane",
                                         "static_lds_bytes", "max_workgroup", "waves_per_simd_by_registers")} for r in rows],
        }
        with open(args.json, "w") as f:
            json.dump(summary, f, indent=1)
            f.write("\n")
    print("%-110s %5s %5s %5s %6s %7s %7s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratch", "lds", "w/simd"))
    for r in rows:
        print("%-110s %5d %5d %5d %6d %7d %7d %5d" % (r["kernel"][:110], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"],
                                                     r["scratch_bytes_per_lane"], r["static_lds_bytes"],
                                                     r["waves_per_simd_by_registers"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())

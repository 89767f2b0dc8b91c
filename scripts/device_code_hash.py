#!/usr/bin/env python3
"""SHA-256 of the machine code (.text) and constants (.rodata) of every gfx950 code object inside libpaa_hip.so.

Two builds of the same sources give the same hashes (the code objects differ only in paths recorded in their notes), so
the hashes identify the DEVICE code a GPU run validated: a change that touches only the host side of the library, or adds
code behind a switch that is off by default, must leave them as they are.

    python scripts/device_code_hash.py                                   # print
    python scripts/device_code_hash.py --write profiles/r05_device_code.json --note "pytest -m gpu green at <commit>"
    python scripts/device_code_hash.py --check profiles/r05_device_code.json

scripts/gpu_round.sh records them after a green `pytest -m gpu`; tests/test_abi_cpu.py::test_device_code_is_the_validated_one
compares the shipped binary with the record.
"""
import argparse
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import resource_usage as ru          # noqa: E402  (the bundle / ELF readers)


def elf_sections(image):
    shoff, = struct.unpack_from("<Q", image, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", image, 0x3A)

    def header(i):
        base = shoff + i * shentsize
        name, typ = struct.unpack_from("<II", image, base)
        off, size = struct.unpack_from("<QQ", image, base + 0x18)
        return name, typ, off, size
    _, _, stro, strs = header(shstrndx)
    strtab = image[stro:stro + strs]
    for i in range(shnum):
        name, typ, off, size = header(i)
        label = strtab[name:strtab.index(b"\0", name)].decode()
        yield label, (b"" if typ == 8 else image[off:off + size])          # SHT_NOBITS has no bytes


def device_code(lib):
    out = []
    for image in ru.code_objects(lib):
        sec = dict(elf_sections(image))
        kernels = sorted(k["symbol"] for k in kernels_in(image))
        out.append({"text_bytes": len(sec.get(".text", b"")),
                    "text_sha256": hashlib.sha256(sec.get(".text", b"")).hexdigest(),
                    "rodata_sha256": hashlib.sha256(sec.get(".rodata", b"")).hexdigest(),
                    "kernels": len(kernels), "first_kernel": kernels[0] if kernels else ""})
    out.sort(key=lambda e: e["first_kernel"])
    return out


def kernels_in(image):
    import msgpack
    for name, ntype, desc in ru.elf_notes(image):
        if name == "AMDGPU" and ntype == 32:
            for k in msgpack.unpackb(desc, raw=False, strict_map_key=False).get("amdhsa.kernels", []):
                yield {"symbol": k[".name"]}


def compiler_id():
    """What built the library here: the machine code (and the register counts) are a function of the hipcc / LLVM version
    and of PAA_HIPCC_FLAGS, so the records carry it and the tests that compare against them skip on another compiler."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout.splitlines()
    except Exception:
        return None
    keep = [ln.strip() for ln in out if ln.startswith("HIP version") or "clang version" in ln]
    return " | ".join(keep + ["PAA_HIPCC_FLAGS=" + os.environ.get("PAA_HIPCC_FLAGS", "")]) if keep else None


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--lib", default=ru.LIB)
    ap.add_argument("--write")
    ap.add_argument("--note", default="")
    ap.add_argument("--check")
    args = ap.parse_args()
    units = device_code(args.lib)
    if args.write:
        with open(args.write, "w") as f:
            json.dump({"what": "machine code of the gfx950 code objects inside pyaudioanalysis_amd/libpaa_hip.so "
                               "(scripts/device_code_hash.py)", "note": args.note, "compiler": compiler_id(),
                       "code_objects": units}, f, indent=1)
            f.write("\n")
    for u in units:
        print("%8d  %s  %s  %3d kernels  %s" % (u["text_bytes"], u["text_sha256"][:16], u["rodata_sha256"][:8], u["kernels"],
                                                u["first_kernel"][:60]))
    if args.check:
        want = json.load(open(args.check))["code_objects"]
        same = [(w["text_sha256"], w["rodata_sha256"]) for w in want] == [(u["text_sha256"], u["rodata_sha256"]) for u in units]
        print("device code %s %s" % ("==" if same else "!=", args.check))
        return 0 if same else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())

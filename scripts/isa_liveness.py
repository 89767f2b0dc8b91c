"""Rough VGPR liveness over a straight-line range of a kernel listing (lines a..b of the file): registers read before
written are live-in; prints the maximum number of simultaneously live VGPRs and the live-through set size.
usage: isa_liveness.py k.s first_line last_line"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")[int(sys.argv[2]) - 1:int(sys.argv[3])]
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1) is not None: out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
ins = []
for l in lines:
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"): continue
    m = re.match(r"([a-z_0-9]+)\s*(.*)", t)
    if not m: continue
    op, rest = m.group(1), m.group(2)
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(("ds_write", "global_store", "scratch_store", "s_", "buffer_store", "ds_add")) or op.startswith("v_cmp") and not op.startswith("v_cmpx"):
        d, u = [], regs(rest)
    else:
        d, u = regs(ops[0]) if ops else [], regs(",".join(ops[1:]))
        if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_writelane", "v_mov_b32_dpp", "v_pk_fmac")): u = u + d
    ins.append((op, d, u))
# backward liveness
live = set(); maxlive = 0; trace = []
for op, d, u in reversed(ins):
    live -= set(d); live |= set(u)
    trace.append(len(live)); maxlive = max(maxlive, len(live))
trace.reverse()
touched = set()
for op, d, u in ins: touched |= set(d) | set(u)
print("instructions", len(ins), "max live (touched regs only)", maxlive, "live-in", len(live), "touched", len(touched))
step = max(1, len(trace) // 40)
print("live profile:", " ".join(str(trace[k]) for k in range(0, len(trace), step)))

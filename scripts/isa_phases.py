"""Per-phase static instruction / spill counts of one kernel in a hipcc -save-temps .s file.  Phases are cut at the
wave barriers (wsync) of the kernel.  usage: isa_phases.py file.s mangled-substring"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = end = None
for n, l in enumerate(lines):
    if start is None and l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (start is None and re.match(r"^_Z\w*%s\w*:" % re.escape(key), l)):
        start = n
    elif start is not None and l.startswith(".Lfunc_end"):
        end = n
        break
body = lines[start:end]
keys = ("n", "valu64", "valu", "ds", "scr_ld", "scr_st", "glob", "salu", "wait", "rsq", "readlane", "writelane", "accvgpr")
def new(): return dict.fromkeys(keys, 0)
stats, cur = [], new()
for l in body:
    t = l.strip()
    if t.startswith("; wave barrier") or t.startswith("s_barrier"):
        stats.append(cur); cur = new(); continue
    m = re.match(r"([a-z_0-9]+)", t)
    if not m or t.startswith((".", ";")) or t.endswith(":"): continue
    op = m.group(1); cur["n"] += 1
    if op.startswith("scratch_load"): cur["scr_ld"] += 1
    elif op.startswith("scratch_store"): cur["scr_st"] += 1
    elif op.startswith("v_accvgpr"): cur["accvgpr"] += 1
    elif op.startswith("v_readlane") or op.startswith("v_readfirstlane"): cur["readlane"] += 1
    elif op.startswith("v_writelane"): cur["writelane"] += 1
    elif op.startswith("v_") and "f64" in op: cur["valu64"] += 1
    elif op.startswith("v_"): cur["valu"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): cur["glob"] += 1
    elif op.startswith("s_waitcnt"): cur["wait"] += 1
    elif op.startswith("s_"): cur["salu"] += 1
    if "rsq" in op: cur["rsq"] += 1
stats.append(cur)
print("phase " + " ".join("%9s" % k for k in keys))
for k, c in enumerate(stats): print("%5d " % k + " ".join("%9d" % c[x] for x in keys))
tot = {x: sum(c[x] for c in stats) for x in keys}
print("total " + " ".join("%9d" % tot[x] for x in keys))

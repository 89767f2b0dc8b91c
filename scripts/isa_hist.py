"""Instruction histogram of one kernel in a hipcc -save-temps .s file (static counts, whole kernel or its largest loop).
usage: isa_hist.py file.s mangled-substring [--loop]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
loop = "--loop" in sys.argv
lines = open(path).read().split("\n")
start = end = None
for n, l in enumerate(lines):
    if start is None and re.match(r"^(_Z\w*%s\w*):" % re.escape(key), l): start = n
    elif start is not None and l.startswith(".Lfunc_end"): end = n; break
body = lines[start:end]
if loop:
    # largest backward-branch span
    labels = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = (0, 0, 0)
    for n, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n and n - labels[m.group(1)] > best[0]:
            best = (n - labels[m.group(1)], labels[m.group(1)], n)
    body = body[best[1]:best[2] + 1]
    print("loop span: %d lines" % best[0])
cls = collections.Counter(); ops = collections.Counter()
for l in body:
    m = re.match(r"^\s+([a-z_0-9]+)\b", l)
    if not m or l.strip().startswith((".", ";")): continue
    op = m.group(1); ops[op] += 1
    if op.startswith("v_") and ("f64" in op): c = "valu_f64"
    elif op.startswith("v_"): c = "valu_other"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c = "vmem"
    elif op.startswith("s_waitcnt"): c = "waitcnt"
    elif op.startswith("s_"): c = "salu"
    else: c = "other"
    cls[c] += 1
print(dict(cls))
for op, n in ops.most_common(45): print("%6d %s" % (n, op))

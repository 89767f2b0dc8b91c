#!/usr/bin/env python
"""LDS-array cycle model of the two exchanges of a PACKED three-pass shape of kernels_tri.hpp for candidate (row pitch P, group
pitch R3P): exact lane -> address maps of the kernel (exchange 1: ds_write_b64 of [q P + j], ds_read_b64 of [q1 P + R3 k + b];
exchange 2: ds_write_b64 of [q1 P + q R3P + b], pass-3 reads of the lane jobs in the host table's order), bank rules of
MI355X_MICROARCH.md (ds_write_b64: 4 x 16 contiguous lanes over 16 double-banks; ds_read_b64: 2 x 32 lanes over 32).
Development aid for VERDICT r05 item 6 (16 x 16 x 4: conflict ratio 0.48; 20 x 16 x 3: 0.44).
    python scripts/dev/tri_pad_search.py 16 16 4      python scripts/dev/tri_pad_search.py 20 16 3"""
import sys


def cyc(addrs, write):
    g, mod = (16, 16) if write else (32, 32)
    total = 0
    for s in range(0, 64, g):
        a = [v for v in addrs[s:s + g] if v >= 0]
        if not a:
            continue
        banks = {}
        for v in set(a):
            banks[v % mod] = banks.get(v % mod, 0) + 1
        total += max(banks.values())
    return total


def jobs(R1, R2, R3):
    N = R1 * R2 * R3
    out = []
    fold = R3 == 4 and (R1 * R2) % 2 == 0
    for q1 in range(R1):
        for q2 in range(R2):
            k = q1 + R1 * q2
            m = (N - k) % N
            p1, p2 = m % R1, (m // R1) % R2
            if not (p1 > q1 or (p1 == q1 and p2 >= q2)):
                continue
            self_ = (p1, p2) == (q1, q2)
            out.append(((q1, q2), (p1, p2), self_))
    if fold and len(out) % 64 == 1:
        kf = R1 * R2 // 2
        out = [j for j in out if not (j[2] and j[0] != (0, 0))]
        out[0] = ((0, 0), (kf % R1, kf // R1), True)
    return out


def model(R1, R2, R3, P, R3P):
    L1, J2 = R2 * R3, R1 * R3
    c = {"ex1w": 0, "ex1r": 0, "ex2w": 0, "ex2r": 0}
    ideal = dict(c)
    for q in range(R1):
        c["ex1w"] += cyc([q * P + l if l < L1 else -1 for l in range(64)], True); ideal["ex1w"] += (min(L1, 64) + 15) // 16
    for k in range(R2):
        c["ex1r"] += cyc([(l // R3) * P + R3 * k + (l % R3) if l < J2 else -1 for l in range(64)], False); ideal["ex1r"] += (min(J2, 64) + 31) // 32
    for q in range(R2):
        c["ex2w"] += cyc([(l // R3) * P + q * R3P + (l % R3) if l < J2 else -1 for l in range(64)], True); ideal["ex2w"] += (min(J2, 64) + 15) // 16
    jl = jobs(R1, R2, R3)
    for u in range((len(jl) + 63) // 64):
        for side in (0, 1):
            for b in range(R3):
                ad = []
                for l in range(64):
                    p = l + 64 * u
                    if p >= len(jl):
                        ad.append(-1); continue
                    q1, q2 = jl[p][side]
                    ad.append(q1 * P + q2 * R3P + b)
                c["ex2r"] += cyc(ad, False); ideal["ex2r"] += 2
    return c, ideal, len(jl)


if __name__ == "__main__":
    R1, R2, R3 = (int(v) for v in sys.argv[1:4])
    L1 = R2 * R3
    rows = []
    for R3P in range(R3, R3 + 4):
        for P in range(max(L1, (R2 - 1) * R3P + R3), L1 + 48):
            c, ideal, nj = model(R1, R2, R3, P, R3P)
            rows.append((2 * sum(c.values()), P, R3P, c, R1 * P))          # x2: real plane + imaginary plane
    rows.sort(key=lambda r: (r[0], r[4]))
    _, ideal, nj = model(R1, R2, R3, L1, R3)
    print("jobs", nj, "ideal cycles per frame (both planes)", 2 * sum(ideal.values()), ideal)
    for r in rows[:12]:
        print("cycles %4d  P %3d R3P %d  plane %5d doubles  %s" % (r[0], r[1], r[2], r[4], r[3]))
    for P, R3P in ((68, 4), (49, 3)):
        if P >= max(L1, (R2 - 1) * R3P + R3) and R3P >= R3:
            c, _, _ = model(R1, R2, R3, P, R3P)
            print("current? P %d R3P %d: cycles %d %s" % (P, R3P, 2 * sum(c.values()), c))

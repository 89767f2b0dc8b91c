#!/usr/bin/env python
"""NumPy model of the three-pass register FFT of kernels_tri.hpp: validates the index maps (jobs, twiddles, pairing of
Z[k] / Z[N-k], output bins) against numpy.fft and counts LDS bank conflicts of the exchange layouts.  Development tool."""
import sys
import numpy as np


def W(n, e):
    return np.exp(-2j * np.pi * (np.asarray(e) % n) / n)


def three_pass(z, R1, R2, R3, nq1=None):
    """Z = fft(z), N = R1 R2 R3, computed the way the kernel does; nq1: only q1 < nq1 (real input)."""
    N = R1 * R2 * R3
    L1 = R2 * R3
    nq1 = R1 if nq1 is None else nq1
    # pass 1: job j < L1: A_j[q1] = sum_r z[j + L1 r] W_R1^(r q1), then twiddle W_N^(j q1)
    B = np.zeros((L1, R1), complex)
    for j in range(L1):
        B[j] = np.fft.fft(z[j + L1 * np.arange(R1)]) * W(N, j * np.arange(R1))
    # pass 2: job (q1, b): C[q2] = sum_a B[R3 a + b][q1] W_R2^(a q2), then twiddle W_L1^(b q2)
    D = np.zeros((R1, R3, R2), complex)
    for q1 in range(nq1):
        for b in range(R3):
            D[q1, b] = np.fft.fft(B[R3 * np.arange(R2) + b, q1]) * W(L1, b * np.arange(R2))
    # pass 3: job (q1, q2): Z[q1 + R1 (q2 + R2 k3)] = sum_b D[q1][b][q2] W_R3^(b k3)
    Z = np.full(N, np.nan + 0j)
    for q1 in range(nq1):
        for q2 in range(R2):
            Z[q1 + R1 * (q2 + R2 * np.arange(R3))] = np.fft.fft(D[q1, :, q2])
    return Z


def pair_jobs(R1, R2, R3):
    """packed shapes: pass-3 pair-jobs (jobA, jobB): Z[k] of jobA at k3 meets Z[N - k] of jobB at k3' """
    N = R1 * R2 * R3
    seen, pairs = set(), []
    for q1 in range(R1):
        for q2 in range(R2):
            if (q1, q2) in seen:
                continue
            k = q1 + R1 * q2
            m = (N - k) % N
            p1, p2 = m % R1, (m // R1) % R2
            seen.add((q1, q2)); seen.add((p1, p2))
            pairs.append(((q1, q2), (p1, p2)))
    return pairs


def check_packed(Wn, R1, R2, R3, rng):
    N = R1 * R2 * R3
    assert 2 * N == Wn
    y = rng.standard_normal(Wn)
    z = y[0::2] + 1j * y[1::2]
    Z = three_pass(z, R1, R2, R3)
    assert np.allclose(Z, np.fft.fft(z), atol=1e-9)
    ref = np.fft.fft(y)[:N]
    X = np.full(N, np.nan + 0j)
    cnt = np.zeros(N, int)
    pairs = pair_jobs(R1, R2, R3)
    for (a, b) in pairs:
        kA = a[0] + R1 * a[1]
        kB = b[0] + R1 * b[1]
        for k3 in range(R3):
            k = kA + R1 * R2 * k3
            m = (N - k) % N
            # where the partner sits: job b, some k3'
            assert (m - kB) % (R1 * R2) == 0
            k3p = (m - kB) // (R1 * R2)
            if a == b or k == 0:
                ok = (k3p == (R3 - k3) % R3) if (a == (0, 0)) else (k3p == R3 - 1 - k3)
            else:
                ok = k3p == R3 - 1 - k3
            assert ok, (a, b, k3, k3p)
            zk, zm = Z[k], Z[m]
            e = 0.5 * (zk + np.conj(zm))
            o = -0.5j * (zk - np.conj(zm))
            w = np.exp(-1j * np.pi * k / N)
            self_pair = a == b
            st1 = (not self_pair) or (2 * k <= N)
            st2 = (k != 0) and ((not self_pair) or (2 * k < N))
            if st1:
                X[k] = e + w * o; cnt[k] += 1
            if st2:
                X[N - k] = np.conj(e - w * o); cnt[N - k] += 1
    assert np.all(cnt == 1), np.where(cnt != 1)
    assert np.allclose(X, ref, atol=1e-9)
    return len(pairs)


def check_real(Wn, R1, R2, R3, rng):
    N = R1 * R2 * R3
    assert N == Wn and R1 % 2 == 1
    H1 = (R1 - 1) // 2
    x = rng.standard_normal(Wn)
    Z = three_pass(x.astype(complex), R1, R2, R3, nq1=H1 + 1)
    NF = Wn // 2
    ref = np.abs(np.fft.fft(x))[:NF]
    X = np.full(NF, np.nan)
    cnt = np.zeros(NF, int)
    for q1 in range(H1 + 1):
        for q2 in range(R2):
            for k3 in range(R3):
                k = q1 + R1 * (q2 + R2 * k3)
                if k < NF:
                    X[k] = abs(Z[k]); cnt[k] += 1
                elif q1 > 0 and N - k < NF:
                    X[N - k] = abs(Z[k]); cnt[N - k] += 1
    assert np.all(cnt == 1), np.where(cnt != 1)
    assert np.allclose(X, ref, atol=1e-9)
    return (H1 + 1) * R2


# ---- LDS conflict model (MI355X_MICROARCH.md): ds_read_b64: 2 groups of 32 lanes, doubles distinct mod 32;
# ds_write_b64: 4 groups of 16 contiguous lanes, doubles distinct mod 16.  Returns LDS-array cycles of one instruction.
def cyc(addrs, write):
    addrs = np.asarray(addrs)
    g, mod = (16, 16) if write else (32, 32)
    total = 0
    for s in range(0, 64, g):
        a = addrs[s:s + g]
        a = a[a >= 0]
        if len(a) == 0:
            total += 1
            continue
        banks = {}
        for v in np.unique(a):
            banks.setdefault(int(v) % mod, 0)
            banks[int(v) % mod] += 1
        total += max(banks.values())
    return total


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    print("2400 pairs", check_packed(2400, 20, 20, 3, rng))
    print("1764 pairs", check_packed(1764, 21, 21, 2, rng))
    print("1920 pairs", check_packed(1920, 20, 16, 3, rng))
    print("1920b pairs", check_packed(1920, 16, 20, 3, rng))
    print("1600 pairs", check_packed(1600, 20, 20, 2, rng))
    print("2205 jobs", check_real(2205, 21, 21, 5, rng))
    print("ok")


def layout_search(R1, R2, R3, packed, NF):
    import itertools
    N = R1 * R2 * R3
    L1 = R2 * R3
    NQ1 = R1 if packed else (R1 - 1) // 2 + 1
    nj = (L1 + 63) // 64
    best = None
    for P1 in range(L1, L1 + 34):
        for order in (0, 1):
            # writers
            c = 0
            for q1 in range(NQ1):
                for u in range(nj):
                    ad = [q1 * P1 + l + 64 * u if l + 64 * u < L1 else -1 for l in range(64)]
                    c += cyc(ad, True)
            for a in range(R2):
                ad = []
                for l in range(64):
                    if l >= NQ1 * R3:
                        ad.append(-1); continue
                    q1, b = (l // R3, l % R3) if order == 0 else (l % NQ1, l // NQ1)
                    ad.append(q1 * P1 + R3 * a + b)
                c += cyc(ad, False)
            size = NQ1 * P1
            key = (c, size)
            if best is None or key < best[0]:
                best = (key, P1, order)
    print("exchange 1: cycles %d (ideal %d), plane %d doubles, P1 %d, order %d" % (
        best[0][0], NQ1 * nj * 4 + R2 * 2, best[0][1], best[1], best[2]))
    order = best[2]
    if packed:
        pairs = pair_jobs(R1, R2, R3)
        jobs3 = [(a, b) for a, b in pairs]
    else:
        jobs3 = [((q1, q2), None) for q1 in range(NQ1) for q2 in range(R2)]
    best2 = None
    for P2 in range(L1, L1 + 34):
        c = 0
        for q2 in range(R2):
            ad = []
            for l in range(64):
                if l >= NQ1 * R3:
                    ad.append(-1); continue
                q1, b = (l // R3, l % R3) if order == 0 else (l % NQ1, l // NQ1)
                ad.append(q1 * P2 + q2 * R3 + b)
            c += cyc(ad, True)
        rounds = (len(jobs3) + 63) // 64
        for u in range(rounds):
            for b in range(R3):
                for side in range(2 if packed else 1):
                    ad = []
                    for l in range(64):
                        p = l + 64 * u
                        if p >= len(jobs3):
                            ad.append(-1); continue
                        q1, q2 = jobs3[p][side]
                        ad.append(q1 * P2 + q2 * R3 + b)
                    c += cyc(ad, False)
        key = (c, NQ1 * P2)
        if best2 is None or key < best2[0]:
            best2 = (key, P2)
    rounds = (len(jobs3) + 63) // 64
    print("exchange 2: cycles %d (ideal %d), plane %d doubles, P2 %d, rounds %d" % (
        best2[0][0], R2 * 4 + rounds * R3 * (2 if packed else 1) * 2, best2[0][1], best2[1], rounds))
    return best[1], best[2], best2[1]


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "layout":
    for name, (R1, R2, R3, packed, NF) in {"2400": (20, 20, 3, True, 1200), "2205": (21, 21, 5, False, 1102),
                                           "1764": (21, 21, 2, True, 882), "1920": (20, 16, 3, True, 960),
                                           "1920b": (16, 20, 3, True, 960)}.items():
        print(name)
        layout_search(R1, R2, R3, packed, NF)


def layout_table(R1, R2, R3, packed):
    L1 = R2 * R3
    NQ1 = R1 if packed else (R1 - 1) // 2 + 1
    nj = (L1 + 63) // 64
    jobs3 = pair_jobs(R1, R2, R3) if packed else [((q1, q2), None) for q1 in range(NQ1) for q2 in range(R2)]
    rounds = (len(jobs3) + 63) // 64
    for P in range(L1, L1 + 12):
        row = []
        for order in (0, 1):
            c1 = 0
            for q1 in range(NQ1):
                for u in range(nj):
                    c1 += cyc([q1 * P + l + 64 * u if l + 64 * u < L1 else -1 for l in range(64)], True)
            for a in range(R2):
                ad = []
                for l in range(64):
                    if l >= NQ1 * R3:
                        ad.append(-1); continue
                    q1, b = (l // R3, l % R3) if order == 0 else (l % NQ1, l // NQ1)
                    ad.append(q1 * P + R3 * a + b)
                c1 += cyc(ad, False)
            c2 = 0
            for q2 in range(R2):
                ad = []
                for l in range(64):
                    if l >= NQ1 * R3:
                        ad.append(-1); continue
                    q1, b = (l // R3, l % R3) if order == 0 else (l % NQ1, l // NQ1)
                    ad.append(q1 * P + q2 * R3 + b)
                c2 += cyc(ad, True)
            for u in range(rounds):
                for b in range(R3):
                    for side in range(2 if packed else 1):
                        ad = []
                        for l in range(64):
                            p = l + 64 * u
                            if p >= len(jobs3):
                                ad.append(-1); continue
                            q1, q2 = jobs3[p][side]
                            ad.append(q1 * P + q2 * R3 + b)
                        c2 += cyc(ad, False)
            row.append((c1, c2))
        print("  P %3d plane %4d  order0 ex1 %3d ex2 %3d | order1 ex1 %3d ex2 %3d" % (P, NQ1 * P, row[0][0], row[0][1], row[1][0], row[1][1]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "table":
    for name, (R1, R2, R3, packed) in {"2400": (20, 20, 3, True), "2205": (21, 21, 5, False),
                                       "1764": (21, 21, 2, True), "1920": (20, 16, 3, True)}.items():
        print(name)
        layout_table(R1, R2, R3, packed)

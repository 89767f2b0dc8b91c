#!/usr/bin/env python
"""NumPy model of the Bluestein kernel (csrc/kernels_blu.hpp): the in-place DIF forward passes, the pointwise product in the
DIF's digit-reversed order, the DIT passes back to natural order, the chirp tables -- and an LDS bank model of the b128
accesses of every pass for candidate skews.  Development aid; tests/test_abi_cpu.py restates the same algebra against the
library's own tables."""
import sys

import numpy as np

SCHED = {256: [16, 16], 512: [8, 8, 8], 1024: [16, 8, 8], 2048: [16, 16, 8], 4096: [16, 16, 16]}


def passes(M):
    out, span = [], M
    for R in SCHED[M]:
        out.append((R, span, span // R))
        span //= R
    return out


def dif(x, M):
    buf = x.copy()
    for R, Mp, s in passes(M):
        for b in range(M // R):
            blk, k = divmod(b, s)
            base = blk * Mp + k
            v = buf[base + s * np.arange(R)]
            V = np.fft.fft(v)
            V *= np.exp(-2j * np.pi * np.arange(R) * k / Mp)
            buf[base + s * np.arange(R)] = V
    return buf


def dit(x, M):
    buf = x.copy()
    for R, Mp, s in reversed(passes(M)):
        for b in range(M // R):
            blk, k = divmod(b, s)
            base = blk * Mp + k
            v = buf[base + s * np.arange(R)] * np.exp(-2j * np.pi * np.arange(R) * k / Mp)
            buf[base + s * np.arange(R)] = np.fft.fft(v)
    return buf


def perm(M):
    """position of X[k] after the DIF passes"""
    pos = np.zeros(M, dtype=int)
    for k in range(M):
        rest, weight, p = k, M, 0
        for R in SCHED[M]:
            weight //= R
            p += (rest % R) * weight
            rest //= R
        pos[k] = p
    return pos


def chirp(W, m):
    m = np.asarray(m, dtype=np.int64)
    return np.exp(1j * np.pi * ((m * m) % (2 * W)) / W)


def bluestein_mag(y, M):
    W = len(y)
    Nf = W // 2
    assert M >= W + Nf - 1
    a = np.zeros(M, dtype=complex)
    a[:W] = y * np.conj(chirp(W, np.arange(W)))
    b = np.zeros(M, dtype=complex)
    for m in range(-(W - 1), Nf):
        b[m % M] = chirp(W, m)
    pos = perm(M)
    Bp = np.zeros(M, dtype=complex)
    Bp[pos] = np.fft.fft(b) / M
    A = dif(a, M)
    v = dit(np.conj(A * Bp), M)
    return np.abs(v[:Nf]) / Nf


def lds_model(M, skew_shift):
    """LDS-array cycles of one forward pass set (reads + writes), b128: reads 4 groups of 16 lanes over 16 slots of 16 B,
    writes 8 groups of 8 contiguous lanes over 8 slots"""
    rgroups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    rgroups += [[l + 32 for l in g] for g in rgroups]
    wgroups = [list(range(8 * i, 8 * i + 8)) for i in range(8)]

    def sk(e):
        return e + (e >> skew_shift) if skew_shift else e
    tot_r = tot_w = ideal_r = ideal_w = 0
    for R, Mp, s in passes(M):
        nb = M // R
        for b0 in range(0, nb, 64):
            for r in range(R):
                addr = []
                for lane in range(64):
                    b = min(b0 + lane, nb - 1)
                    blk, k = divmod(b, s)
                    addr.append(sk(blk * Mp + k + r * s))
                for groups, slots, which in ((rgroups, 16, 0), (wgroups, 8, 1)):
                    cyc = 0
                    for g in groups:
                        per = {}
                        for l in g:
                            per.setdefault(addr[l] % slots, set()).add(addr[l])
                        cyc += max(len(v) for v in per.values())
                    if which == 0:
                        tot_r += cyc; ideal_r += len(groups)
                    else:
                        tot_w += cyc; ideal_w += len(groups)
    return tot_r, ideal_r, tot_w, ideal_w


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for W, M in ((661, 1024), (1103, 2048), (149, 256), (300, 512), (2731, 4096), (34, 256)):
        y = rng.standard_normal(W)
        ref = np.abs(np.fft.fft(y))[:W // 2] / (W // 2)
        got = bluestein_mag(y, M)
        print("W %5d M %5d  max|d| %.3g  (max ref %.3g)" % (W, M, np.max(np.abs(got - ref)), ref.max()))
    for M in SCHED:
        for sh in (0, 3, 4, 5, 6):
            print("M %5d skew>>%d  reads %d (ideal %d)  writes %d (ideal %d)" % ((M, sh) + lds_model(M, sh)))

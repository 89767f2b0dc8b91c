#!/usr/bin/env python3
"""NumPy restatement of kernels_wgs.hpp: the REAL-INPUT split of a long even window into r0 / 2 independent sub-transforms of
Q = W / r0 complex points (44 100 samples = 12 x 3675, 22 050 = 6 x 3675), each run as three register passes 7 x 21 x 25.

    unit q = 1 .. r0/2 - 1:  a_q[k] = W_W^(q k) sum_r y[k + Q r] W_r0^(r q),  A_q = FFT_Q(a_q),  X[q + r0 kappa] = A_q[kappa]
                             (bins beyond W / 2 are the mirrors of bins r0 - q + r0 (Q - 1 - kappa))
    packed unit (q = 0, r0/2): u[n] = sum_{r < r0/2} y[n + 2 Q r] (n < 2 Q),  v[k] = u[2 k] + i u[2 k + 1],  V = FFT_Q(v),
                             X[(r0/2) j] = E + W_(2Q)^j O with E, O from V[j], V[Q - j]
No sub-transform needs another one's outputs (kernels_wg.hpp's split pairs sub-transform q with r0 - q).

usage: wgs_model.py [W]      prints the largest deviation from numpy's rfft
"""
import sys

import numpy as np

R1, R2, R3 = 7, 21, 25
Q = R1 * R2 * R3


def three_pass(a):
    """FFT of Q complex points the way the kernel runs it: in place, element n = n2 + R3 n1 + R2 R3 n0."""
    J1 = R2 * R3
    buf = a.astype(complex).copy()
    # pass 1: job j < J1, radix R1 over n0, outputs k0 times W_Q^(j k0), back to j + J1 k0
    m = buf.reshape(R1, J1)                                   # [n0][j]
    F1 = np.exp(-2j * np.pi * np.outer(np.arange(R1), np.arange(R1)) / R1)
    m = F1 @ m                                                # [k0][j]
    m *= np.exp(-2j * np.pi * np.outer(np.arange(R1), np.arange(J1)) / Q)
    # pass 2: job (k0, n2), radix R2 over n1, outputs k1 times W_(R2 R3)^(n2 k1), back to k0 J1 + k1 R3 + n2
    c = m.reshape(R1, R2, R3)                                 # [k0][n1][n2]
    F2 = np.exp(-2j * np.pi * np.outer(np.arange(R2), np.arange(R2)) / R2)
    c = np.einsum("kn,anb->akb", F2, c)                       # [k0][k1][n2]
    c *= np.exp(-2j * np.pi * np.outer(np.arange(R2), np.arange(R3)) / (R2 * R3))[None, :, :]
    # pass 3: job (k0, k1), radix R3 over n2 -> A[k0 + R1 k1 + R1 R2 k2]
    F3 = np.exp(-2j * np.pi * np.outer(np.arange(R3), np.arange(R3)) / R3)
    d = np.einsum("kn,abn->abk", F3, c)                       # [k0][k1][k2]
    A = np.empty(Q, complex)
    k0, k1, k2 = np.meshgrid(np.arange(R1), np.arange(R2), np.arange(R3), indexing="ij")
    A[(k0 + R1 * k1 + R1 * R2 * k2).ravel()] = d.ravel()
    return A


def unit_inputs(y, r0, q):
    k = np.arange(Q)
    s = y.reshape(r0, Q)                                      # [r][k]
    D = (s * np.exp(-2j * np.pi * np.arange(r0) * q / r0)[:, None]).sum(axis=0)
    return D * np.exp(-2j * np.pi * q * k / (r0 * Q))


def magnitudes(y, r0):
    W = y.size
    Nf = W // 2
    row = np.full(Nf, np.nan)
    for q in range(1, r0 // 2):
        A = three_pass(unit_inputs(y, r0, q))
        m = q + r0 * np.arange(Q)
        bins = np.where(m < Nf, m, W - m)
        row[bins] = np.abs(A)
    u = y.reshape(r0 // 2, 2 * Q).sum(axis=0)
    V = three_pass(u[0::2] + 1j * u[1::2])
    j = np.arange(Q)
    Vm = np.conj(V[(Q - j) % Q])
    E, O = 0.5 * (V + Vm), -0.5j * (V - Vm)
    U = E + np.exp(-2j * np.pi * j / (2 * Q)) * O
    row[(r0 // 2) * j] = np.abs(U)
    return row


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 44100
    r0 = W // Q
    assert r0 * Q == W and r0 % 2 == 0
    rng = np.random.default_rng(1)
    y = rng.standard_normal(W)
    ref = np.abs(np.fft.rfft(y))[:W // 2]
    got = magnitudes(y, r0)
    assert not np.isnan(got).any()
    print("W = %d, r0 = %d: max |d| / max |X| = %.2e" % (W, r0, np.abs(got - ref).max() / ref.max()))


if __name__ == "__main__":
    main()

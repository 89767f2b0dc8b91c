"""Seeded synthetic 16-bit PCM -- shared by tests, golden generation and bench.py.

TEST / BENCH INFRASTRUCTURE.  Draw order is part of the contract (so that the
judge and the builder agree on the bytes): f_j, a_j, phi_j (5 each), the noise
vector, then the start of the zeroed 0.5 s span.
"""
import numpy as np


def synth_clip(seed, n_samples, fs=16000, stereo=False):
    """int16 mono clip (n,) -- or (n, 2) when stereo -- of 5 sines + noise under a slow envelope."""
    rng = np.random.default_rng(seed)
    chans = 2 if stereo else 1
    out = np.empty((n_samples, chans), dtype=np.int16)
    t = np.arange(n_samples, dtype=np.float64) / fs
    env = 0.5 * (1.0 + np.sin(2.0 * np.pi * 0.25 * t))
    for c in range(chans):
        f = rng.uniform(80.0, 0.45 * fs, 5)
        a = rng.uniform(0.2, 1.0, 5)
        ph = rng.uniform(0.0, 2.0 * np.pi, 5)
        x = np.zeros(n_samples)
        for j in range(5):
            x += a[j] * np.sin(2.0 * np.pi * f[j] * t + ph[j])
        x = 6000.0 * x + 1500.0 * rng.standard_normal(n_samples)
        x *= env
        span = int(0.5 * fs)
        if n_samples > span:
            s0 = int(rng.integers(0, n_samples - span))
            x[s0:s0 + span] = 0.0
        out[:, c] = np.clip(np.round(x), -32768, 32767).astype(np.int16)
    return out if stereo else out[:, 0]


def synth_batch(seed0, n_clips, n_samples, fs=16000):
    """(packed int16 [n_clips*n_samples], offsets int64 [n_clips+1]); clip i uses seed0+i."""
    packed = np.empty(n_clips * n_samples, dtype=np.int16)
    for i in range(n_clips):
        packed[i * n_samples:(i + 1) * n_samples] = synth_clip(seed0 + i, n_samples, fs)
    offsets = np.arange(n_clips + 1, dtype=np.int64) * n_samples
    return packed, offsets


def fast_noise_clip(seed, n_samples):
    """Cheap int16 test signal for very large sizes (bench only): tone + LCG-free numpy noise."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-3000, 3000, n_samples, dtype=np.int16)
    return x


def synth_song(seed, seconds, fs=16000, section=4.0):
    """int16 "song" for the thumbnailing row: sections of `section` seconds in the pattern
    A B A C A B ..., each section type a fixed chord of 4 partials + noise, so that the self-similarity matrix has
    pronounced off-diagonal stripes.  Draw order: 3 x (4 frequencies, 4 amplitudes), then the noise vector."""
    rng = np.random.default_rng(seed)
    n = int(seconds * fs)
    t = np.arange(n, dtype=np.float64) / fs
    kinds = []
    for _ in range(3):
        kinds.append((rng.uniform(100.0, 0.3 * fs, 4), rng.uniform(0.3, 1.0, 4)))
    pattern = [0, 1, 0, 2]
    x = np.zeros(n)
    sec_n = int(section * fs)
    for b in range((n + sec_n - 1) // sec_n):
        f, a = kinds[pattern[b % len(pattern)]]
        sl = slice(b * sec_n, min(n, (b + 1) * sec_n))
        for j in range(4):
            x[sl] += a[j] * np.sin(2.0 * np.pi * f[j] * t[sl])
    x = 5000.0 * x + 800.0 * rng.standard_normal(n)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)

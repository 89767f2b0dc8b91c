"""CPU-side checks of the C-ABI library: it loads, exports every symbol of include/paa_hip.h,
its host tables equal the oracle's, and it refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import paa_oracle as O
from conftest import ROOT
from pyaudioanalysis_amd import MidTermFeatures, ShortTermFeatures, _ffi


def header_symbols():
    text = open(os.path.join(ROOT, "include", "paa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(paa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    lib = _ffi.lib()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), "libpaa_hip.so does not export %s" % s
    assert set(syms) == set(_ffi.EXPORTED_SYMBOLS), set(syms) ^ set(_ffi.EXPORTED_SYMBOLS)


def test_shape_helpers():
    lib = _ffi.lib()
    assert lib.paa_num_frames(128164, 800, 400) == 319          # doremi.wav
    assert lib.paa_num_frames(799, 800, 400) == 0
    assert lib.paa_num_frames(800, 800, 400) == 1
    assert lib.paa_num_frames(1199, 800, 400) == 1
    assert lib.paa_num_frames(1200, 800, 400) == 2
    assert lib.paa_num_mid_windows(1199, 40) == 30               # BASELINE config 3
    assert lib.paa_num_mid_windows(100, 100) == 1
    import ctypes
    filled = ctypes.c_int64()
    for n, w, s in ((132437, 1102, 441), (48000, 640, 640), (88200, 1102, 441), (2000, 800, 400)):
        rows = lib.paa_spectrogram_rows(n, w, s, ctypes.byref(filled))
        assert rows == int((n - w) / s) + 1
        assert filled.value == len(range(w, n - w + 1, s))
        rows = lib.paa_chromagram_rows(n, w, s, ctypes.byref(filled))
        assert rows == int((n - s - w) / s) + 1
        assert filled.value == len(range(w, n - s, s))


@pytest.mark.parametrize("fs,nfft", [(16000, 400), (16000, 320), (16000, 160), (44100, 551), (8000, 200),
                                      (22050, 512), (48000, 1024)])
def test_mel_and_chroma_tables_match_oracle(fs, nfft):
    lib = _ffi.lib()
    dense = np.zeros((40, nfft))
    _ffi.check(lib.paa_debug_mel_bank(float(fs), nfft, _ffi.as_f64p(dense)))
    ref = O.mel_bank(fs, nfft)
    assert np.array_equal(dense != 0, ref != 0)
    assert np.allclose(dense, ref, rtol=1e-13, atol=0)
    cap = nfft
    src = np.zeros(cap, dtype=np.int32)
    slot = np.zeros(cap, dtype=np.int32)
    w = np.zeros(cap)
    n = lib.paa_debug_chroma(float(fs), nfft, cap, src.ctypes.data_as(_ffi.c_i32p), _ffi.as_f64p(w),
                             slot.ctypes.data_as(_ffi.c_i32p))
    assert n > 0
    o_src, o_w, o_cls, o_pos = O.chroma_gather(fs, nfft)
    assert n == len(o_src)
    assert np.array_equal(src[:n], o_src)
    assert np.array_equal(slot[:n], o_pos)
    assert np.array_equal(w[:n], o_w)


def test_dct_matches_oracle_and_scipy():
    lib = _ffi.lib()
    m = np.zeros((13, 40))
    _ffi.check(lib.paa_debug_dct(_ffi.as_f64p(m)))
    assert np.allclose(m, O.dct_matrix(), rtol=0, atol=2e-16)
    from scipy.fft import dct
    x = np.random.default_rng(0).standard_normal(40)
    assert np.allclose(m @ x, dct(x, type=2, norm="ortho")[:13], rtol=0, atol=1e-13)


def test_chroma_error_codes():
    lib = _ffi.lib()
    src = np.zeros(200, dtype=np.int32)
    slot = np.zeros(200, dtype=np.int32)
    w = np.zeros(200)
    args = (200, src.ctypes.data_as(_ffi.c_i32p), _ffi.as_f64p(w), slot.ctypes.data_as(_ffi.c_i32p))
    assert lib.paa_debug_chroma(16000.0, 97, *args) == _ffi.ERR_CHROMA_VALUE      # ValueError :293
    assert lib.paa_debug_chroma(16000.0, 98, *args) == _ffi.ERR_CHROMA_INDEX      # IndexError :291
    assert lib.paa_debug_chroma(16000.0, 99, *args) > 0


@pytest.mark.parametrize("window", [800, 1102, 640, 320, 801, 2048, 1103, 4410])
def test_fft_plan_factorisation(window):
    lib = _ffi.lib()
    import ctypes
    rad = np.zeros(32, dtype=np.int32)
    ln = ctypes.c_int32()
    n = lib.paa_debug_fft_plan(window, rad.ctypes.data_as(_ffi.c_i32p), ctypes.byref(ln))
    assert n > 0
    assert ln.value == (window // 2 if window % 2 == 0 else window)
    assert int(np.prod(rad[:n].astype(np.int64))) == ln.value


def test_names_match_reference_strings():
    assert ShortTermFeatures._feature_names(True) == O.feature_names(True)
    assert ShortTermFeatures._feature_names(False) == O.feature_names(False)
    assert MidTermFeatures._mid_names(O.feature_names(True)) == O.mid_feature_names()
    assert MidTermFeatures._ratios(16000, 16000, 800, 400) == (39, 40)           # BASELINE config 3
    assert MidTermFeatures._ratios(1.0 * 16000, 0.1 * 16000, 800.0, 800.0) == O.mid_ratios(16000.0, 1600.0, 800.0, 800.0)


def test_python_boundary_errors_without_touching_the_gpu():
    with pytest.raises(ValueError):                       # ShortTermFeatures.py:684
        ShortTermFeatures.feature_extraction(np.zeros(100, dtype=np.int16), 16000, 800, 400)
    with pytest.raises(ValueError):
        MidTermFeatures.mid_feature_extraction(np.zeros(5000, dtype=np.int16), 16000, 16000, 100, 800, 400)


def test_no_cpu_fallback():
    """Without a HIP device a compute call must raise, never silently compute on the CPU."""
    if _ffi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.HipLibraryError):
        ShortTermFeatures.feature_extraction(np.zeros(4000, dtype=np.int16), 16000, 800, 400)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pyaudioanalysis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "paa_oracle" not in text and "load_reference" not in text, f
                assert "/root/reference" not in text, f


def test_classify_signal_kinds():
    k, a = _ffi.classify_signal(np.zeros(10, dtype=np.int16))
    assert k == 0 and a.dtype == np.int16
    k, a = _ffi.classify_signal(np.zeros(10, dtype=np.float32))
    assert k == 1 and a.dtype == np.float64
    k, a = _ffi.classify_signal([1, 2, 3])
    assert k == 1 and a.dtype == np.float64
    k, a = _ffi.classify_signal(np.zeros((10, 2), dtype=np.int16))
    assert k == 2 and a.shape == (10, 2) and a.flags["C_CONTIGUOUS"]
    k, a = _ffi.classify_signal(np.zeros((10, 2), dtype=np.int16)[:, ::-1])     # non-contiguous view
    assert k == 2 and a.flags["C_CONTIGUOUS"]
    # spectrogram / chromagram take the stereo entry points for (n, 2) int16: no float64 mono copy on the host
    fn, _ = ShortTermFeatures._spec_call(_ffi.lib(), "spectrogram", 2, np.zeros((4, 2), dtype=np.int16))
    assert fn is _ffi.lib().paa_spectrogram_stereo_i16 or fn.__name__ == "paa_spectrogram_stereo_i16"
    with pytest.raises(ValueError):
        _ffi.classify_signal(np.zeros((4, 3)))


def test_c_client_links_and_fails_loudly_without_gpu(tmp_path):
    """examples/c_api_demo.c: a plain C program links against the C ABI (no Python, no torch in the signatures)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    _ffi.lib()      # make sure the library is built
    exe = str(tmp_path / "c_api_demo")
    libdir = os.path.dirname(_ffi.library_path())
    cmd = ["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_api_demo.c"), "-o", exe,
           "-L" + libdir, "-lpaa_hip", "-Wl,-rpath," + libdir, "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if _ffi.device_count() > 0:
        assert run.returncode == 0 and "frames 79" in run.stdout, run.stdout + run.stderr
    else:
        assert run.returncode == 2 and "no HIP device" in run.stdout, run.stdout + run.stderr


def _run_plan(frames, quantum=4, min_run=16, max_run=256, halo=4, wg_runs=8, num_cu=256):
    frames = np.ascontiguousarray(frames, dtype=np.int64)
    cap, longest, runs = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    _ffi.check(_ffi.lib().paa_debug_run_plan(_ffi.as_i64p(frames), len(frames), quantum, min_run, max_run, halo, wg_runs,
                                             num_cu, ctypes.byref(cap), ctypes.byref(runs), ctypes.byref(longest)))
    return cap.value, runs.value, longest.value


def test_run_length_choice_for_the_baseline_batches():
    """paa_plan_create's run sizing (host only): equal runs per clip, the cap minimising rounds x (run + halo)."""
    cap, runs, longest = _run_plan([143999])                      # config 2: one round of 250 workgroups (the equal-run rule; since round 5 such a
                                                                  # plan is re-cut into 256 x 8 runs: test_balanced_runs_of_a_one_round_plan)
    assert (runs, longest) == (2000, 72)
    cap, runs, longest = _run_plan([399] * 12500)                 # config 4 shard: 4 x 100 per clip, not 244 + 155
    assert (runs, longest) == (50000, 100)
    cap, runs, longest = _run_plan([1199] * 1000)                 # config 3: 6 x 200 per clip, not 6 x 196 + 23
    assert (runs, longest) == (6000, 200)
    cap, runs, longest = _run_plan([59998], quantum=3, min_run=12, max_run=96, halo=3, wg_runs=6)      # config 5
    assert longest % 3 == 0 and -(-runs // 6) <= 256
    rng = np.random.default_rng(4)
    for _ in range(20):
        frames = rng.integers(0, 3000, rng.integers(1, 400))
        cap, runs, longest = _run_plan(frames)
        assert 16 <= cap <= 256 and cap % 4 == 0 and longest <= cap and longest % 4 == 0
        live = frames[frames > 0]
        assert runs >= len(live) and runs >= -(-int(live.sum()) // 256)
    assert _ffi.lib().paa_debug_run_plan(None, 1, 4, 16, 256, 4, 8, 256, None, None, None) == _ffi.ERR_ARG
    # kernels whose halo rides inside the first iteration (runs after a clip's first are 1 or 2 frames shorter): the count
    # that decides the rounds is the tile list's -- 90 000 frames at 640 / 640 were 2 093 runs = 262 workgroups = two rounds
    # on 256 CUs when the chooser assumed T / len runs (0.36 ms instead of 0.22 ms)
    for frames, shrink in (([90000], 1), ([90000], 2), ([179999], 2), ([143999], 1), ([1199] * 1000, 2)):
        arr = np.ascontiguousarray(frames, dtype=np.int64)
        cap, longest, runs = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        _ffi.check(_ffi.lib().paa_debug_run_plan_shrink(_ffi.as_i64p(arr), len(arr), 4, 16, 256, shrink, 8, 256,
                                                        ctypes.byref(cap), ctypes.byref(runs), ctypes.byref(longest)))
        wgs = -(-runs.value // 8)
        rounds = -(-wgs // 256)
        ideal_rounds = max(1, -(-int(arr.sum()) // (256 * 8 * 256)))
        assert rounds == ideal_rounds or len(frames) > 1, (frames[:1], shrink, runs.value, wgs)
        if len(frames) == 1:
            assert wgs <= 256 and longest.value <= 256


@pytest.mark.parametrize("window", [44100, 22050, 48000, 32000, 24000])
def test_real_input_split_plan_reproduces_the_spectrum(window):
    """Host side of csrc/kernels_wgs.hpp (no device): the REAL-INPUT split of a window of r0 x Q samples (12 / 6 x 3675, 12 / 8 / 6 x 4000) into
    r0 / 2 independent transforms of Q complex points -- complex unit q: a_q[k] = W_W^(q k) sum_r y[k + Q r] W_r0^(r q) in the kernel's difference form, the
    packed unit from the sums over the even / odd samples --, each as three IN-PLACE register passes (7 x 21 x 25 / 8 x 20 x 25) over the padded exchange
    buffer, and the UNIT-MAJOR spectrum row with the library's own bin map: restated in NumPy from the library's constants this gives
    |fft(frame)|[0:W/2] / (W/2), every bin written exactly once (ShortTermFeatures.py:617-621); a constant frame gives exact zeros in every
    complex unit's input (the exact spectrum of a digitally silent frame)."""
    lib = _ffi.lib()
    info = np.zeros(16, dtype=np.int32)
    Nf = window // 2
    bin_of = np.zeros(Nf, dtype=np.int32)
    assert lib.paa_debug_wgs_plan(window, info.ctypes.data_as(_ffi.c_i32p), bin_of.ctypes.data_as(_ffi.c_i32p), Nf) == 1
    r0, Q, R1, R2, R3, A, threads, lds, n_types, low_bins, feat_lds, n_blocks, feat_threads = (int(v) for v in info[:13])
    assert r0 * Q == window and R1 * R2 * R3 == Q and A >= R2 * R3 and lds <= 160 * 1024 and 3 * feat_lds <= 160 * 1024
    assert 2 * R1 * A * 16 <= lds and low_bins * 8 <= feat_lds and n_blocks == -(-Nf // (64 * r0)) and n_types == (3 if r0 == 6 else (r0 // 2 + 1) // 2)
    # the feature kernel reads a row as residue streams: 64 consecutive elements of a stream lie in one natural block of 64 r0 bins
    inv = np.empty(Nf, dtype=np.int64)
    inv[bin_of] = np.arange(Nf)
    for rho in range(1, r0):
        if rho == r0 // 2:
            continue
        n = np.arange(-(-(Nf - rho) // r0))
        idx = (rho - 1) * Q + n if rho < r0 // 2 else (r0 - rho - 1) * Q + Q - 1 - n
        assert np.array_equal(bin_of[idx], rho + r0 * n) and np.array_equal((rho + r0 * n) // (64 * r0), n // 64)
    n = np.arange(Q)
    assert np.array_equal(bin_of[(r0 // 2 - 1) * Q + n], (r0 // 2) * n) and np.array_equal(((r0 // 2) * n) // (64 * r0), n // 128)
    assert sorted(bin_of.tolist()) == list(range(Nf))                       # a permutation: every bin exactly once
    H0, J1 = r0 // 2, R2 * R3
    rng = np.random.default_rng(window)
    x = rng.standard_normal(window)
    pos = lambda k: k + (A - J1) * (k // J1)                                # element k of a unit sits at k + (A - R2 R3) (k / (R2 R3))
    S60 = np.sqrt(3.0) / 2

    def split_dft(s, q):
        """wgs::split_dft: the DFT over r of s[r] = y[k + Q r] at q, from sums and differences"""
        if r0 == 8:
            h = np.sqrt(0.5)
            if q == 2:
                return ((s[0] + s[4]) - (s[2] + s[6])) + 1j * (-((s[1] + s[5]) - (s[3] + s[7])))
            o = s[0:4] - s[4:8]
            if q == 1:
                return (o[0] + h * (o[1] - o[3])) + 1j * (-(o[2] + h * (o[1] + o[3])))
            return (o[0] - h * (o[1] - o[3])) + 1j * (o[2] - h * (o[1] + o[3]))
        if r0 == 6:
            if q == 1:
                d = s[0:3] - s[3:6]
                return (d[0] + 0.5 * (d[1] - d[2])) + 1j * (-S60 * (d[1] + d[2]))
            g = s[0:3] + s[3:6]
            return (g[0] - 0.5 * (g[1] + g[2])) + 1j * (-S60 * (g[1] - g[2]))
        e, o = s[0:6] + s[6:12], s[0:6] - s[6:12]
        if q == 2:
            d = e[0:3] - e[3:6]
            return (d[0] + 0.5 * (d[1] - d[2])) + 1j * (-S60 * (d[1] + d[2]))
        if q == 4:
            g = e[0:3] + e[3:6]
            return (g[0] - 0.5 * (g[1] + g[2])) + 1j * (-S60 * (g[1] - g[2]))
        if q == 3:
            return ((o[0] - o[2]) + o[4]) + 1j * (-((o[1] - o[3]) + o[5]))
        sg = 1.0 if q == 1 else -1.0
        return (o[0] + 0.5 * (o[2] - o[4]) + sg * S60 * (o[1] - o[5])) + 1j * (-(o[3] + 0.5 * (o[1] + o[5]) + sg * S60 * (o[2] + o[4])))

    def three_passes(a):
        buf = np.zeros(R1 * A, dtype=complex)
        buf[pos(np.arange(Q))] = a
        j = np.arange(J1)
        F1 = np.exp(-2j * np.pi * np.outer(np.arange(R1), np.arange(R1)) / R1)
        idx = np.arange(R1)[:, None] * A + j[None, :]                       # pass 1: job j, elements n0 A + j, in place
        buf[idx] = (F1 @ buf[idx]) * np.exp(-2j * np.pi * np.outer(np.arange(R1), j) / Q)
        F2 = np.exp(-2j * np.pi * np.outer(np.arange(R2), np.arange(R2)) / R2)
        for k0 in range(R1):                                                # pass 2: job (n2, k0), elements k0 A + n1 R3 + n2, in place
            idx = k0 * A + np.arange(R2)[:, None] * R3 + np.arange(R3)[None, :]
            buf[idx] = (F2 @ buf[idx]) * np.exp(-2j * np.pi * np.outer(np.arange(R2), np.arange(R3)) / (R2 * R3))
        F3 = np.exp(-2j * np.pi * np.outer(np.arange(R3), np.arange(R3)) / R3)
        out = np.empty(Q, dtype=complex)
        for k0 in range(R1):                                                # pass 3: job (k0, k1) -> A[k0 + R1 k1 + R1 R2 k2]
            for k1 in range(R2):
                out[k0 + R1 * k1 + R1 * R2 * np.arange(R3)] = F3 @ buf[k0 * A + k1 * R3 + np.arange(R3)]
        return out

    s = x.reshape(r0, Q)
    k = np.arange(Q)
    row = np.full(Nf, np.nan)                                               # unit-major
    for q in range(1, H0):
        a = split_dft(s, q) * np.exp(-2j * np.pi * k / window) ** q
        assert np.all(split_dft(np.full((r0, Q), 0.37), q) == 0.0)           # equal samples: exact zeros
        row[(q - 1) * Q:q * Q] = np.abs(three_passes(a)) / Nf
    u = x.reshape(H0, 2 * Q).sum(axis=0)
    V = three_passes(u[0::2] + 1j * u[1::2])
    jj = np.arange(Q)
    Vm = np.conj(V[(Q - jj) % Q])
    row[(H0 - 1) * Q:] = np.abs(0.5 * (V + Vm) + np.exp(-2j * np.pi * H0 * jj / window) * (-0.5j * (V - Vm))) / Nf
    natural = np.empty(Nf)
    natural[bin_of] = row
    ref = np.abs(np.fft.fft(x))[:Nf] / Nf
    assert np.max(np.abs(natural - ref)) <= 1e-13 * ref.max()


@pytest.mark.parametrize("window", [16000, 8000, 9009, 44100, 48000, 11025, 22050, 32000, 65536])
def test_workgroup_fft_plan_reproduces_the_spectrum(window):
    """Host side of csrc/kernels_wg.hpp (no device): the in-place decimation-in-frequency passes, the padded LDS layout and the
    digit-reversal permutation -- and, for sequences beyond one CU's LDS, the first radix-r0 pass straight from the samples with the
    pairing of sub-transform q with r0 - q -- restated in NumPy FROM THE LIBRARY'S OWN PLAN give |fft(frame)|[0:W/2] / (W/2), every bin
    written exactly once (ShortTermFeatures.py:617-621).  The kernels execute exactly this index algebra."""
    lib = _ffi.lib()
    info = np.zeros(48, dtype=np.int32)
    assert lib.paa_debug_wg_plan(window, info.ctypes.data_as(_ffi.c_i32p), None, 0) == 1
    Nc, Nf, n_pass, r0, n_el, top = (int(v) for v in info[:6])
    perm = np.zeros(n_el, dtype=np.uint16)
    assert lib.paa_debug_wg_plan(window, info.ctypes.data_as(_ffi.c_i32p), perm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), n_el) == 1
    passes = [tuple(int(v) for v in info[11 + 3 * i:14 + 3 * i]) for i in range(n_pass)]
    packed = window % 2 == 0
    assert Nc == (window // 2 if packed else window) and Nf == window // 2
    assert int(np.prod([p[0] for p in passes])) == n_el and (r0 == 0 or r0 * n_el == Nc) and int(info[7]) <= 160 * 1024
    for (R, M, tws) in passes:
        assert tws * M == Nc                                   # twiddle W_M^j = W_Nc^(j tws): one table for every pass
    rng = np.random.default_rng(window)
    x = rng.standard_normal(window)
    z = (x[0::2] + 1j * x[1::2]) if packed else x.astype(np.complex128)
    W = np.exp(-2j * np.pi * np.arange(Nc) / Nc)               # the twiddle table P.tw
    pad = lambda e: e + e // top                                # element e of a (sub-)transform sits at e + e / top
    pitch = n_el + passes[0][0]

    def run_passes(buf, n_sub):
        """wg_run_passes: butterfly b of a pass works on base + r stride, output q times W_M^(q k) goes back to base + q stride"""
        for (R, M, tws) in passes:
            stride = M // R
            F = np.exp(-2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
            for sub in range(n_sub):
                b = np.arange(n_el // R)
                blk, k = b // stride, b % stride
                e0 = blk * M + k
                idx = sub * pitch + pad(e0[:, None] + np.arange(R)[None, :] * stride)
                v = buf[idx] @ F.T
                v *= W[(np.arange(R)[None, :] * (k[:, None] * tws)) % Nc]
                buf[idx] = v
        return buf

    ref = np.abs(np.fft.fft(x))[:Nf] / Nf
    row = np.full(Nf, np.nan)
    writes = np.zeros(Nf, dtype=np.int32)
    post = np.exp(-2j * np.pi * np.arange(Nc // 2 + 1) / (2 * Nc)) if packed else None      # P.post: w^k of the real-FFT recombination

    def put(k, v):
        row[k] = v
        writes[k] += 1

    def pair(lo, zl, zh):
        e = 0.5 * (zl + np.conj(zh))
        o = -0.5j * (zl - np.conj(zh))
        wo = post[lo] * o
        put(lo, abs(e + wo) / Nf)
        if lo > 0 and Nc - lo != lo:
            put(Nc - lo, abs(np.conj(e - wo)) / Nf)

    if r0 == 0:
        buf = np.zeros(Nc + passes[0][0] + 8, dtype=np.complex128)
        buf[pad(np.arange(Nc))] = z
        run_passes(buf, 1)
        Z = buf[perm.astype(np.int64)]
        assert np.allclose(Z, np.fft.fft(z), atol=1e-9 * np.abs(z).sum())
        if packed:
            for k in range(Nc // 2 + 1):
                pair(k, Z[k], Z[0 if k == 0 else Nc - k])
        else:
            for k in range(Nf):
                put(k, abs(Z[k]) / Nf)
    else:
        S = n_el
        for qa in range(r0 // 2 + 1):
            qb = 0 if qa == 0 else r0 - qa
            two = qb != qa
            buf = np.zeros(2 * pitch + 8, dtype=np.complex128)
            k = np.arange(S)
            zr = z[k[None, :] + np.arange(r0)[:, None] * S]                              # z[k + r S]
            cv = W[((np.arange(r0) * qa) % r0) * S]                                      # W_r0^(r qa)
            buf[pad(k)] = (zr * cv[:, None]).sum(axis=0) * W[(qa * k) % Nc]
            if two:
                buf[pitch + pad(k)] = (zr * np.conj(cv)[:, None]).sum(axis=0) * W[(qb * k) % Nc]
            run_passes(buf, 2 if two else 1)
            A = buf[perm.astype(np.int64)]
            B = buf[pitch + perm.astype(np.int64)] if two else A
            if packed:
                n_k = S if two else (S // 2 + 1 if qa == 0 else (S + 1) // 2)
                for ka in range(n_k):
                    kb = (0 if ka == 0 else S - ka) if qa == 0 else S - 1 - ka
                    kk = qa + r0 * ka
                    flip = 2 * kk > Nc
                    lo = Nc - kk if flip else kk
                    pair(lo, B[kb] if flip else A[ka], A[ka] if flip else B[kb])
            else:
                for q, buf_q in ((qa, A),) + (((qb, B),) if two else ()):
                    for ka in range(S):
                        if q + r0 * ka < Nf:
                            put(q + r0 * ka, abs(buf_q[ka]) / Nf)
    assert np.all(writes == 1)
    assert np.allclose(row, ref, rtol=1e-9, atol=1e-11)


def test_lane_jobs_cover_every_owner_once_inside_one_row():
    """csrc/kernels_tri.hpp: lane_jobs -- the mel sums (40 filters) and the chroma gather (12 classes) of the three-pass kernels run on
    all 64 lanes: every owner's entries are cut into consecutive pieces on consecutive lanes of ONE 16-lane row (the segmented scan is a
    row_shr DPP scan), piece i of an owner carries i in its control word, and lane k carries the lane that ends up with owner k's total."""
    rng = np.random.default_rng(12)
    cases = [(40, rng.integers(0, 60, 40)) for _ in range(30)] + [(12, rng.integers(20, 120, 12)) for _ in range(30)]
    cases += [(40, np.arange(2, 42)), (40, np.full(40, 1)), (12, np.full(12, 46)), (12, np.array([0] * 11 + [700])), (40, np.zeros(40, dtype=np.int64))]
    for K, cnt in cases:
        cnt = np.ascontiguousarray(cnt, dtype=np.int32)
        first = np.ascontiguousarray(np.concatenate(([0], np.cumsum(cnt)[:-1])) + 5, dtype=np.int32)
        wfirst = np.ascontiguousarray(first * 3 + 1, dtype=np.int32)
        jobs = np.zeros(256, dtype=np.int32)
        _ffi.check(_ffi.lib().paa_debug_lane_jobs(first.ctypes.data_as(_ffi.c_i32p), wfirst.ctypes.data_as(_ffi.c_i32p),
                                                  cnt.ctypes.data_as(_ffi.c_i32p), K, jobs.ctypes.data_as(_ffi.c_i32p)))
        j = jobs.reshape(64, 4)
        seen = np.zeros(int(cnt.sum()) + 5, dtype=np.int32)
        for k in range(K):
            last = (j[k, 3] >> 8) & 63
            lanes = [last]
            while j[lanes[0], 3] & 15:                                  # walk back to the owner's first piece
                lanes.insert(0, lanes[0] - 1)
            assert len(lanes) <= 16 and lanes[0] // 16 == lanes[-1] // 16           # one row
            assert [int(j[l, 3] & 15) for l in lanes] == list(range(len(lanes)))
            pos = int(first[k])
            for l in lanes:
                assert j[l, 0] == pos and j[l, 2] == wfirst[k] + (pos - first[k]) and j[l, 1] >= 0
                seen[pos:pos + j[l, 1]] += 1
                pos += int(j[l, 1])
            assert pos == first[k] + cnt[k]
            sizes = [int(j[l, 1]) for l in lanes]
            assert max(sizes) - min(sizes) <= 1
        assert np.all(seen[5:] == 1) and j[:, 1].sum() == cnt.sum()


def test_balanced_runs_of_a_one_round_plan():
    """A plan whose equal runs fill less than one round of the hot kernel (one workgroup per CU, eight runs each) is re-cut into
    256 x 8 runs (csrc/lib_plan.hpp: balanced_runs): the lengths cover the clip exactly, runs after a clip's first store `shrink`
    frames less (their halo rides inside the first quad), iteration counts differ by at most one, and the two waves that share a
    SIMD (w and w + 4 of a workgroup) get a long and a short run where the ratio allows."""
    def lens_of(frames, cap, shrink, wg_runs=8, num_cu=256, quantum=4, min_run=16):
        arr = np.ascontiguousarray(frames, dtype=np.int64)
        out = np.zeros(8192, dtype=np.int32)
        n = _ffi.lib().paa_debug_balanced_runs(_ffi.as_i64p(arr), len(arr), cap, quantum, shrink, wg_runs, num_cu, min_run,
                                               out.ctypes.data_as(_ffi.c_i32p), len(out))
        assert n >= 0
        return out[:n]
    def cap_of(frames, shrink):
        arr = np.ascontiguousarray(frames, dtype=np.int64)
        cap, longest, runs = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        _ffi.check(_ffi.lib().paa_debug_run_plan_shrink(_ffi.as_i64p(arr), len(arr), 4, 16, 256, shrink, 8, 256,
                                                        ctypes.byref(cap), ctypes.byref(runs), ctypes.byref(longest)))
        return cap.value, runs.value
    for shrink in (1, 2):
        cap, equal_runs = cap_of([143999], shrink)                # config 2 (34 rows: one halo frame, 68 rows: two)
        assert equal_runs < 2048
        l = lens_of([143999], cap, shrink)                        # 2048 runs instead
        assert len(l) == 2048 and int(l.sum()) == 143999
        iters = (l + np.where(np.arange(len(l)) > 0, shrink, 0) + 3) // 4
        assert iters[:-1].max() - iters[:-1].min() <= 1 and iters[-1] <= iters[:-1].max()
        assert iters.max() == (18 if shrink == 1 else 19)       # (until round 5: 18 iterations + a halo quad for both)
        pairs = iters[:-8].reshape(-1, 2, 4)                      # (workgroup, first / second wave of a SIMD, SIMD)
        assert np.abs(pairs.sum(axis=1) - pairs.sum(axis=1).mean()).max() <= 1.0       # SIMD loads within one iteration
        assert iters.reshape(-1, 8).sum(axis=1).max() - iters.reshape(-1, 8).sum(axis=1).min() <= 2
    # the same function for twelve waves per workgroup (three per SIMD), one frame per iteration, the halo outside the run (the shape of the
    # three-pass family, where the A/B showed no gain and the equal runs stay)
    l = lens_of([59998], 20, 0, wg_runs=12, quantum=1, min_run=8)             # config 5: 3072 runs of 19 / 20 instead of 3000 of 20
    assert len(l) == 3072 and int(l.sum()) == 59998 and set(l[:-1]) <= {19, 20}
    per_simd = l[:3072 - 12].reshape(-1, 3, 4).sum(axis=1)                    # (workgroup, wave of the SIMD, SIMD)
    assert per_simd.max() - per_simd.min() <= 1
    l = lens_of([40000, 60000, 43999], 72, 1)                     # three clips share the slots in proportion
    assert len(l) <= 2048 and int(l.sum()) == 143999 and l.min() >= 16
    # random one-round batches: whatever the function returns covers every clip exactly, in whole quanta (later runs minus the halo,
    # the last run of a clip minus what the clip does not have), never longer than the cap, never more runs than wave slots
    rng = np.random.default_rng(77)
    for _ in range(200):
        n_clips = int(rng.integers(1, 12))
        frames = rng.integers(200, 40000, n_clips)
        shrink = int(rng.integers(0, 3))
        cap, _runs = cap_of(frames, shrink)
        l = lens_of(frames, cap, shrink)
        if len(l) == 0:
            continue
        assert len(l) <= 2048 and l.min() > 0 and l.max() <= cap and int(l.sum()) == int(frames.sum())
        pos = 0
        for T in frames:                                            # the runs of a clip are consecutive in the list
            acc, first = 0, True
            while acc < T:
                n = int(l[pos]); pos += 1
                last = acc + n == T
                assert acc + n <= T
                if not last:
                    assert (n + (0 if first else shrink)) % 4 == 0
                acc += n; first = False
        assert pos == len(l)
    assert len(lens_of([399] * 12500, 100, 1)) == 0               # many rounds: the equal runs stay
    assert len(lens_of([2048 * 72], 72, 0)) == 0                  # already one full round
    assert len(lens_of([20000], 72, 1)) == 0                      # too short for 2048 runs of 16 frames


def test_result_pool_tracks_liveness_through_views_of_views():
    """_ffi.result_array hands out views of pooled storage; the storage is idle again only when the LAST view derived
    from the result is gone -- slices, reshapes and transposes of the result count (no reference-count heuristics)."""
    import gc
    from pyaudioanalysis_amd import _ffi
    shape = (68, 40000)                                     # 21.8 MB: above the pooling threshold
    a = _ffi.result_array(shape)
    assert a.shape == shape and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
    addr = a.__array_interface__["data"][0]
    tail = a[34:]                                           # a view of the view
    flat = a.reshape(-1)[:7]
    del a
    gc.collect()
    b = _ffi.result_array(shape)                            # the first block is still referenced: a second one
    assert b.__array_interface__["data"][0] != addr
    tail[:] = 1.0
    b[:] = 2.0
    assert float(tail.min()) == 1.0
    del tail
    gc.collect()
    c = _ffi.result_array(shape)                            # `flat` still pins the first block
    assert c.__array_interface__["data"][0] != addr
    del flat
    gc.collect()
    d = _ffi.result_array(shape)                            # now it is idle and comes back
    assert d.__array_interface__["data"][0] == addr
    small = _ffi.result_array((10, 10))
    assert small.flags.owndata                              # small results are ordinary arrays


def _dif_in_place(z, radices):
    """The passes of csrc/kernels_mix.hpp restated in NumPy: in-place decimation in frequency, butterfly over the R elements
    base + r * stride of a block of M, output q multiplied by W_M^(q k) and written to base + q * stride."""
    z = np.array(z, dtype=np.complex128)
    n = len(z)
    M = n
    for R in radices:
        stride = M // R
        dft = np.exp(-2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
        for blk in range(n // M):
            for k in range(stride):
                idx = blk * M + k + stride * np.arange(R)
                z[idx] = (dft @ z[idx]) * np.exp(-2j * np.pi * np.arange(R) * k / M)
        M = stride
    return z


@pytest.mark.parametrize("window", [2400, 2205, 1764, 1920, 1600, 1024, 256, 1100, 390, 1001])
def test_mixed_radix_plan_is_a_valid_transform(window):
    """Host tables of the mixed-radix kernel (no device): the radix schedule multiplies to the FFT length, uses only the
    butterflies the kernel has, the permutation is a bijection, and the restated passes + permutation reproduce np.fft.fft."""
    import ctypes
    lib = _ffi.lib()
    rad = np.zeros(16, dtype=np.int32)
    ln, waves, twg = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    perm = np.zeros(8192, dtype=np.uint16)
    n_pass = lib.paa_debug_mix_plan(window, rad.ctypes.data_as(_ffi.c_i32p), ctypes.byref(ln),
                                    perm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), len(perm), ctypes.byref(waves),
                                    ctypes.byref(twg))
    assert n_pass > 0 and 1 <= waves.value <= 8
    if window in (1024, 256, 1764):           # small windows: the lean instance, six to eight waves per CU, radix <= 8
        assert waves.value >= 6 and max(int(r) for r in rad[:n_pass]) <= 8
    if window in (2400, 2205):                # 34 KB / 44 KB per wave: the full instance
        assert waves.value <= 4
    n = ln.value
    assert n == (window // 2 if window % 2 == 0 else window)
    radices = [int(r) for r in rad[:n_pass]]
    assert int(np.prod(radices)) == n and set(radices) <= {2, 3, 4, 5, 7, 8, 11, 13, 16}
    assert sorted(perm[:n].tolist()) == list(range(n))
    rng = np.random.default_rng(window)
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    got = _dif_in_place(z, radices)[perm[:n].astype(np.int64)]
    ref = np.fft.fft(z)
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


def test_mixed_radix_plan_declines_other_windows():
    import ctypes
    lib = _ffi.lib()
    rad = np.zeros(16, dtype=np.int32)
    ln = ctypes.c_int32()
    for window in (1103, 1102, 58, 2 * 17 * 64):          # prime, 2 x 19 x 29, too small, a factor of 17
        assert lib.paa_debug_mix_plan(window, rad.ctypes.data_as(_ffi.c_i32p), ctypes.byref(ln), None, 0, None, None) == 0


def test_default_build_has_no_experiment_switches():
    """The A/B switches of scripts/ (PAA_KERNEL_DEBUG could drop output stores, PAA_RUN_CAP / PAA_NO_MIX / PAA_F800_WAVES
    change the kernel choice) are compiled in only with -DPAA_EXPERIMENTS: the shipped binary does not even contain their
    names, so no environment variable can change what the product computes."""
    from pyaudioanalysis_amd import _ffi
    blob = open(_ffi.library_path(), "rb").read()
    for name in (b"PAA_KERNEL_DEBUG", b"PAA_RUN_CAP", b"PAA_NO_MIX", b"PAA_F800_WAVES", b"PAA_F800_PACE", b"PAA_MIX_NO_LEAN",
                 b"PAA_MIX_TW_GLOBAL", b"PAA_MIX_NO_SKEW", b"PAA_HIP_FORCE_GENERIC"):
        assert name not in blob, name


def test_shipped_kernels_hold_their_register_budget():
    """What DESIGN 4 claims about registers, read from the SHIPPED binary (scripts/resource_usage.py parses the
    NT_AMDGPU_METADATA notes of the gfx950 code objects inside libpaa_hip.so; profiles/<round>_resource_usage.json is its
    output): every kernel family that serves a shape of the reference's callers -- the 800 kernel, 2 x RA x RB, the
    three-pass and the prime-factor register FFTs -- and the similarity kernel run without scratch, without AGPR parking
    and without spills, at two waves per SIMD or more; a private segment exists only in the lean skewed instance of the
    mixed-radix kernel (TWG = 2: reserved, 12-20 bytes) and AGPRs only in its full instance (TWG = 0, one wave per SIMD:
    windows like 4800 / 6000) -- VERDICT r03, items 1 and 8."""
    import importlib.util
    if os.environ.get("PAA_HIP_LIBRARY"):
        pytest.skip("about the shipped binary, not the build under test (sanitizer / experiment builds)")
    spec = importlib.util.spec_from_file_location("resource_usage", os.path.join(ROOT, "scripts", "resource_usage.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    rows = ru.kernels_of(_ffi.library_path())
    assert len(rows) >= 150
    assert not [r for r in rows if "st_reg_kernel" in r["kernel"]]          # never dispatched by the default build: not shipped
    by_family = {}
    for r in rows:
        by_family.setdefault(r["kernel"].split("<")[0], []).append(r)
    for family, at_least in (("f800::st_fast_800_kernel", 8), ("ct::st_ct_kernel", 48), ("tri::st_tri_kernel", 144),
                             ("sim_gram_kernel", 1), ("st_generic_kernel", 3),
                             ("wg::wg_spectrum_kernel", 6), ("wg::wg_feat_kernel", 1)):
        members = by_family[family]
        assert len(members) >= at_least, (family, len(members))
        for r in members:
            assert r["scratch_bytes_per_lane"] == 0 and r["agpr"] == 0 and r["vgpr_spill"] == 0, r
            assert r["vgpr"] <= 256 and r["waves_per_simd_by_registers"] >= 2, r
    # the Bluestein kernel (round 6): no scratch at any convolution length; the lengths whose LDS buffer leaves one wave per SIMD anyway
    # (M >= 1024: 16 KB and more per wave) park values in AGPRs, the short ones (256 / 512) run two waves per SIMD without
    # (M = 8192, windows of 2732 .. 5461 samples: 128 KB of LDS, ONE wave per CU, eight radix-16 butterflies per lane in the outer passes --
    # that instance spills 2 KB per lane to scratch; it replaces an O(N p) kernel and is the slow tail of the family)
    blu = by_family["blu::st_blu_kernel"]
    assert len(blu) == 30, len(blu)          # 6 lengths x 3 sample types + the packed form of 512 .. 4096
    for r in blu:
        short = ", 8, " in r["kernel"] or ", 9, " in r["kernel"]
        longest = ", 13, " in r["kernel"]
        # ("vgpr" is the unified count: architected + accumulation registers, 512 per lane at one wave per SIMD)
        assert r["vgpr"] <= (256 if short else 512), r
        assert r["scratch_bytes_per_lane"] <= 2048 if longest else (r["scratch_bytes_per_lane"] == 0 and r["vgpr_spill"] <= 8), r
        assert (r["agpr"] == 0 and r["waves_per_simd_by_registers"] >= 2) if short else r["waves_per_simd_by_registers"] >= 1, r
    # the fused three-pass kernel of the 1 s windows (round 6, csrc/kernels_wgr.hpp): 448 threads, two waves per SIMD; the feature instance
    # of 20 x 20 x 20 (80 registers of data per pass + the previous frame's twenty magnitudes) is the one kernel of a caller's shape
    # that spills: at most 24 registers / 96 bytes (kept in registers the magnitudes cost that; through memory 7 - 12 x the HBM
    # traffic, profiles/r06_wgr_variants.txt); the 10 x 20 x 20 instances and the spectrogram / chromagram ones do not
    wgr = by_family["wgr::wgr_kernel"]
    assert len(wgr) == 18, len(wgr)          # 2 shapes x 3 sample types x 3 modes
    for r in wgr:
        big_features = "Shape<20, 20, 20" in r["kernel"] and r["kernel"].endswith(", 0>")
        assert r["vgpr"] <= 256 and r["agpr"] == 0 and r["waves_per_simd_by_registers"] >= 2, r
        assert (r["scratch_bytes_per_lane"] <= 96 and r["vgpr_spill"] <= 24) if big_features else \
               (r["scratch_bytes_per_lane"] == 0 and r["vgpr_spill"] == 0), r
    for r in rows:
        if r["kernel"].startswith("wgr::wgr_kernel<"):
            continue
        lean_skewed = r["kernel"].startswith("mix::st_mix_kernel<") and r["kernel"].endswith(", 2>")
        full = r["kernel"].startswith("mix::st_mix_kernel<") and r["kernel"].endswith(", 0>")
        blu_long = r["kernel"].startswith("blu::st_blu_kernel<") and not (", 8, " in r["kernel"] or ", 9, " in r["kernel"])
        blu_8192 = r["kernel"].startswith("blu::st_blu_kernel<") and ", 13, " in r["kernel"]
        assert (r["scratch_bytes_per_lane"] > 0) <= (lean_skewed or blu_8192), r
        assert blu_8192 or (r["scratch_bytes_per_lane"] <= 20 and r["vgpr_spill"] <= (8 if blu_long else 2)), r
        assert (r["agpr"] > 0) <= (full or blu_long), r
    import bench
    import json
    tracked = os.path.join(ROOT, "profiles", "%s_resource_usage.json" % bench.PROFILE_ROUND)
    record = json.load(open(tracked))
    if record.get("compiler") != ru.compiler_id():
        pytest.skip("the tracked record was made by another compiler (%s): register counts need not match" % record.get("compiler"))
    table = {t["kernel"]: t for t in record["table"]}
    assert set(table) == {r["kernel"] for r in rows}, "%s is not this build's kernel list" % tracked
    for r in rows:                                 # the tracked record is the shipped build's (same compiler, same flags)
        t = table[r["kernel"]]
        assert (t["vgpr"], t["agpr"], t["scratch_bytes_per_lane"]) == (r["vgpr"], r["agpr"], r["scratch_bytes_per_lane"]), (t, r)


def test_device_code_is_the_validated_one():
    """The machine code of every gfx950 code object in the shipped library equals the record of the build whose GPU test
    run, bench line and rocprofv3 passes are the round's evidence (profiles/<round>_device_code.json, scripts/device_code_hash.py;
    two builds of one source tree give the same .text / .rodata bytes).  Whoever changes a kernel re-runs the GPU pass
    (scripts/gpu_round.sh writes the new record) -- a host-side or documentation commit cannot move the device code unnoticed."""
    import importlib.util
    import json
    if os.environ.get("PAA_HIP_LIBRARY"):
        pytest.skip("about the shipped binary, not the build under test (sanitizer / experiment builds)")
    spec = importlib.util.spec_from_file_location("device_code_hash", os.path.join(ROOT, "scripts", "device_code_hash.py"))
    dch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dch)
    import bench
    record = json.load(open(os.path.join(ROOT, "profiles", "%s_device_code.json" % bench.PROFILE_ROUND)))
    # the hash is a function of the hipcc / LLVM version and of PAA_HIPCC_FLAGS as much as of the sources: on another compiler
    # (or without one) the comparison says nothing about the sources (advisor, round 4)
    if record.get("compiler") is None or record.get("compiler") != dch.compiler_id():
        pytest.skip("record made by %s, this host has %s" % (record.get("compiler"), dch.compiler_id()))
    have = dch.device_code(_ffi.library_path())
    want = record["code_objects"]
    assert [(u["first_kernel"], u["kernels"]) for u in have] == [(w["first_kernel"], w["kernels"]) for w in want]
    for u, w in zip(have, want):
        assert (u["text_sha256"], u["rodata_sha256"]) == (w["text_sha256"], w["rodata_sha256"]), \
            "device code of the unit with %s differs from the GPU-validated record" % u["first_kernel"]


def test_profile_records_the_bench_line_quotes_are_this_rounds():
    """bench.py does not measure HBM traffic, the executed FP64 count or the sustained clock itself: it quotes the committed
    records of this round's profiling passes and says so in the line.  The records exist, belong to bench.PROFILE_ROUND and
    to the headline kernel / workload, carry the keys the line reads, and are consistent with each other (the FP64 fraction
    at the sustained clock follows from the two files it is computed from)."""
    import json
    import bench
    rnd = bench.PROFILE_ROUND
    prof = os.path.join(ROOT, "profiles")
    traffic = json.load(open(os.path.join(prof, "latest_traffic.json")))
    assert traffic["round"] == rnd and traffic["kernel"] == "st_fast_800_w8" and traffic["rows"] == 34
    assert traffic["frames"] == (3600 * bench.FS - bench.WINDOW) // bench.STEP + 1
    algorithmic = (2 * bench.STEP + 8 * 34) * traffic["frames"]
    assert 1.0 <= traffic["hbm_bytes_per_launch"] / algorithmic <= 1.10          # traffic within 1.1x of the algorithmic bytes
    assert os.path.exists(os.path.join(prof, traffic["source"].split(" ")[0]))
    ex = json.load(open(os.path.join(prof, "%s_fast800_fp64_executed.json" % rnd)))
    assert ex["frames"] == traffic["frames"] and ex["issued_fp64_flop_per_launch"] > 0
    assert abs(ex["issued_kflop_per_frame"] * 1e3 * ex["frames"] - ex["issued_fp64_flop_per_launch"]) < 1e-6 * ex["issued_fp64_flop_per_launch"]
    ck = json.load(open(os.path.join(prof, "%s_fast800_clock.json" % rnd)))
    assert ck["kernel"] == "st_fast_800_w8" and ck["frames"] == traffic["frames"] and 1.0 < ck["sustained_clock_ghz"] <= ck["data_sheet_clock_ghz"] == 2.4
    assert os.path.exists(os.path.join(ROOT, ck["source"]))
    # the clock is the waves' own cycles over their own life times
    assert abs(ck["wave_cycles_median"] / (ck["wave_life_us_median"] * 1e3) - ck["sustained_clock_ghz"]) < 0.02
    peak_at = bench.FP64_VALU_PEAK_TFLOPS * ck["sustained_clock_ghz"] / ck["data_sheet_clock_ghz"]
    assert 0.25 < ex["issued_tflops"] / peak_at < 0.45


def L1_ok(L1, H1):
    return L1 * H1 <= 64


@pytest.mark.parametrize("window", [2400, 2205, 1764, 1920, 1600, 1200, 1102, 551, 1024, 2048, 512, 256])
def test_three_pass_tables_reproduce_the_fft(window):
    """Host tables of csrc/kernels_tri.hpp (no device): the three passes restated in NumPy FROM THE LIBRARY'S OWN TABLES --
    pass-1 twiddles W_N^(j q1), pass-2 twiddles W_L1^(b q2), for packed (even) windows the pass-3 job pairs with their plane
    offsets, store offsets and post-twiddles -- give |fft(frame)|[0:W/2] / (W/2), every bin written exactly once
    (ShortTermFeatures.py:617-621).  The kernel executes exactly this index algebra in registers."""
    import ctypes
    lib = _ffi.lib()
    shape = np.zeros(8, dtype=np.int32)
    off = np.zeros(7, dtype=np.int32)
    size = lib.paa_debug_tri_plan(window, 44100.0, shape.ctypes.data_as(_ffi.c_i32p), off.ctypes.data_as(_ffi.c_i32p), None, 0)
    assert size > 0
    blob = np.zeros(size, dtype=np.uint8)
    assert lib.paa_debug_tri_plan(window, 44100.0, shape.ctypes.data_as(_ffi.c_i32p), off.ctypes.data_as(_ffi.c_i32p),
                                  blob.ctypes.data_as(ctypes.c_void_p), size) == size
    R1, R2, R3, packed, P, NW, njob3, lds = (int(v) for v in shape)
    packed, R3P = packed & 1, packed >> 8          # second exchange: element (q1, b, q2) at plane[q1 P + q2 R3P + b]
    NW, H1, H2 = NW & 0xff, (NW >> 8) & 0xff, NW >> 16      # H1 / H2 lanes share a prime butterfly of pass 1 / 2 (SplitSel in kernels_tri.hpp)
    # the output indices q every part computes and the parts' multipliers g (part h reads its inputs in the order
    # n -> (n g^-1) mod R, which makes its output q the true output (g q) mod R): the constants of kernels_tri.hpp's SplitSel
    SEL = {(29, 3): ((1, 2, 3, 6, 9), (1, 5, 11)), (19, 3): ((1, 2, 4), (1, 7, 8))}
    split = blob[off[6]:off[6] + 4 * (8 * H1 + 16 * H2)].view(np.int32)
    assert R3P >= R3 and P >= (R2 - 1) * R3P + R3
    N = R1 * R2 * R3
    assert N == (window // 2 if packed else window) and (packed == 1) == (window % 2 == 0 and window != 1102)
    assert R3 <= 8 and (H1 == 1 or L1_ok(R2 * R3, H1)) and R3 * ((R1 if packed else (R1 + 1) // 2)) * H2 <= 64
    L1, NQ1, NF = R2 * R3, (R1 if packed else (R1 + 1) // 2), window // 2
    assert P >= L1 and NQ1 * R3 * H2 <= 64 and lds <= 160 * 1024 and 7 <= NW <= 12
    cplx = lambda o, n: blob[o:o + 16 * n].view(np.float64).reshape(n, 2) @ np.array([1.0, 1j])
    tw2 = cplx(off[0], R2 * R3).reshape(R2, R3)
    tw1 = cplx(off[2], NQ1 * L1).reshape(NQ1, L1)
    rng = np.random.default_rng(window)
    y = rng.standard_normal(window)
    z = (y[0::2] + 1j * y[1::2]) if packed else y.astype(complex)
    plane = np.full(NQ1 * P + 1, np.nan, dtype=complex)
    if H1 > 1:
        # split first pass: part h of job j forms DC + the outputs q of SEL from its rows in permuted order; the host table says
        # which true index a slot stands for, whether it is conjugated, and whether this part delivers it
        qs, gs = SEL[(R1, H1)]
        written = np.zeros(NQ1, dtype=int)
        for h in range(H1):
            ginv = pow(gs[h], -1, R1)
            for j in range(L1):
                Y = np.fft.fft(z[j + L1 * ((np.arange(R1) * ginv) % R1)])
                slots = [Y[0]] + [Y[q] for q in qs]
                for sl, val in enumerate(slots):
                    code = int(split[8 * h + sl])
                    t, conj, on = code & 0xff, bool(code & 0x100), bool(code & 0x200)
                    assert t == (0 if sl == 0 else min((gs[h] * qs[sl - 1]) % R1, R1 - (gs[h] * qs[sl - 1]) % R1))
                    assert conj == (sl > 0 and (gs[h] * qs[sl - 1]) % R1 > R1 // 2)
                    if on:
                        plane[t * P + j] = (np.conj(val) if conj else val) * tw1[t, j]
                        written[t] += (j == 0)
        assert np.all(written == 1), written                   # every pass-1 output delivered by exactly one part
    else:
        for j in range(L1):                                     # pass 1 + exchange 1: element (j, q1) at plane[q1 P + j]
            plane[np.arange(NQ1) * P + j] = np.fft.fft(z[j + L1 * np.arange(R1)])[:NQ1] * tw1[:, j]
    plane2 = np.full(NQ1 * P + 1, np.nan, dtype=complex)
    lane_mag = {}                                               # two-pass split shapes: (slot, lane) -> magnitude
    if H2 > 1:
        qs, gs = SEL[(R2, H2)]
        s2 = split[8 * H1:]
        J2 = NQ1 * R3
        written = np.zeros((NQ1, R3, R2), dtype=int)
        for h in range(H2):
            ginv = pow(gs[h], -1, R2)
            for m2 in range(J2):
                q1, b = divmod(m2, R3)
                Y = np.fft.fft(plane[q1 * P + R3 * ((np.arange(R2) * ginv) % R2) + b])
                slots = [Y[0]]
                for q in qs:
                    slots += [Y[q], Y[R2 - q]]
                for sl, val in enumerate(slots):
                    code = int(s2[16 * h + sl])
                    t, on = code & 0xff, bool(code & 0x200)
                    want_t = 0 if sl == 0 else ((gs[h] * qs[(sl - 1) // 2]) % R2 if sl % 2 == 1 else R2 - (gs[h] * qs[(sl - 1) // 2]) % R2)
                    assert t == want_t
                    val = val * tw2[t, b]
                    lane_mag[(sl, h * J2 + m2)] = (val, q1, t, on)
                    if on:
                        plane2[q1 * P + t * R3P + b] = val
                        written[q1, b, t] += 1
        assert np.all(written == 1)                             # every pass-2 output delivered by exactly one part
    else:
        for q1 in range(NQ1):                                   # pass 2 + exchange 2: (q1, b, q2) at plane[q1 P + q2 R3P + b]
            for b in range(R3):
                c = np.fft.fft(plane[q1 * P + R3 * np.arange(R2) + b]) * tw2[:, b]
                plane2[q1 * P + np.arange(R2) * R3P + b] = c
    X = np.full(NF, np.nan)
    hits = np.zeros(NF, dtype=int)
    if R3 == 1 and H2 > 1:                                      # two passes, split second pass: slot s of lane (part, q1)
        assert not packed
        ns2 = 1 + 2 * len(SEL[(R2, H2)][0])
        where = blob[off[1]:off[1] + 2 * 64 * ns2].view(np.uint16).reshape(ns2, 64)      # byte offset of the lane's magnitude of slot s
        assert np.all(where % 8 == 0) and np.all(where[:, NQ1 * H2:] == 8 * NF)          # idle lanes: the parking double
        for (sl, lane), (val, q1, t, on) in lane_mag.items():
            k = int(where[sl, lane]) // 8
            assert k <= NF and (k == NF if not on else k in (q1 + R1 * t, N - q1 - R1 * t, NF))
            if k < NF:
                X[k] = abs(val); hits[k] += 1
    elif R3 == 1:                                               # two passes: lane q1 holds Z[q1 + R1 q2]
        assert not packed
        where = blob[off[1]:off[1] + 2 * 64 * R2].view(np.uint16).reshape(R2, 64)      # byte offset of lane q1's magnitude q2
        assert np.all(where % 8 == 0) and np.all(where[:, NQ1:] == 8 * NF)              # idle lanes: the parking double
        for q1 in range(NQ1):
            for q2 in range(R2):
                k = int(where[q2, q1]) // 8
                assert k <= NF and k in (q1 + R1 * q2, N - q1 - R1 * q2, NF)
                if k < NF:
                    X[k] = abs(plane2[q1 * P + q2]); hits[k] += 1
    elif packed:
        # entries of PE x 8 uint16 (PE = ceil((2 + 2 R3) / 8): one 16-byte word for R3 <= 3, two for radix 4, three for radix 8),
        # BYTE offsets into the frame's slot: plane elements of job A, of job B, then per output k3 where |X[k]| and |X[N - k]| go;
        # 8 NF = the parking double (a result no bin takes)
        E = 8 * ((2 + 2 * R3 + 7) // 8)
        p3 = blob[off[1]:off[1] + 2 * E * 64 * ((njob3 + 63) // 64)].view(np.uint16).reshape(-1, E)
        post = cplx(off[3], 64 * ((njob3 + 63) // 64) * R3).reshape(-1, R3)
        assert np.all(p3 % 8 == 0) and np.all(p3[njob3:, 2:] == 8 * NF) and np.all(p3[:, 2 + 2 * R3:] == 8 * NF)
        p3 = p3 // 8
        # 16 x 16 x 4 (2048 samples): entry 0 holds job 0 (A) AND the second self-paired job R1 R2 / 2 (B) -- 128 lane jobs instead of
        # 129; its slots: B's pair (0, 3), A's (1, 3), A's (2, 2), B's (1, 2); bin 0 comes from Z[0] alone (Shape::FOLD)
        fold = int(p3[0, 0]) != int(p3[0, 1])
        assert fold == (R3 == 4 and njob3 % 64 == 0 and R1 * R2 == 256)
        for p in range(njob3):
            offA, offB = int(p3[p, 0]), int(p3[p, 1])
            zA, zB = np.fft.fft(plane2[offA:offA + R3]), np.fft.fft(plane2[offB:offB + R3])
            if fold and p == 0:
                kf = R1 * R2 // 2
                assert offB == (kf % R1) * P + (kf // R1) * R3P
                X[0] = abs(zA[0].real + zA[0].imag); hits[0] += 1
            for k3 in range(R3):
                zk = zA[k3]
                zm = zA[(R3 - k3) % R3] if p == 0 else zB[R3 - 1 - k3]          # job 0 is (q1, q2) = (0, 0): its own partner
                if fold and p == 0:
                    zk, zm = ((zB[0], zB[3]), (zA[1], zA[3]), (zA[2], zA[2]), (zB[1], zB[2]))[k3]
                e, o = 0.5 * (zk + np.conj(zm)), -0.5j * (zk - np.conj(zm))
                ka, kb = int(p3[p, 2 + 2 * k3]), int(p3[p, 3 + 2 * k3])
                assert ka <= NF and kb <= NF and (ka == NF or kb == NF or ka + kb == N)
                if ka < NF:
                    assert abs(post[p, k3] - np.exp(-1j * np.pi * ka / N)) < 1e-15
                    X[ka] = abs(e + post[p, k3] * o); hits[ka] += 1
                if kb < NF:
                    assert abs(post[p, k3] - np.exp(-1j * np.pi * (N - kb) / N)) < 1e-15
                    X[kb] = abs(e - post[p, k3] * o); hits[kb] += 1
    else:
        assert njob3 == NQ1 * R2
        # entries of 8 x uint16 byte offsets: the job's plane elements, then where its R3 magnitudes go (8 NF: nowhere)
        p3 = blob[off[1]:off[1] + 16 * 64 * ((njob3 + 63) // 64)].view(np.uint16).reshape(-1, 8)
        assert np.all(p3 % 8 == 0) and np.all(p3[njob3:, 1:] == 8 * NF) and np.all(p3[:, 1 + R3:] == 8 * NF)
        p3 = p3 // 8
        for m3 in range(njob3):
            q1, q2 = divmod(m3, R2)
            assert p3[m3, 0] == q1 * P + q2 * R3P
            zz = np.fft.fft(plane2[int(p3[m3, 0]) + np.arange(R3)])
            for k3 in range(R3):
                k, dst = q1 + R1 * (q2 + R2 * k3), int(p3[m3, 1 + k3])
                assert dst in (k, N - k, NF)
                if dst < NF:
                    X[dst] = abs(zz[k3]); hits[dst] += 1
    assert np.all(hits == 1), np.flatnonzero(hits != 1)[:8]
    ref = np.abs(np.fft.fft(y))[:NF]
    assert np.max(np.abs(X - ref)) < 1e-10 * np.max(ref)


def test_bench_self_launch_refuses_a_job_the_box_cannot_run(monkeypatch, capsys):
    """bench.py --gpus N without a launcher environment: with fewer than N devices and the RCCL gather
    asked for, the launcher refuses with a non-zero code and a message -- no rank is started, no line is printed; a
    WORLD_SIZE that contradicts --gpus is refused too (the line never claims GPUs that did not run).  No GPU needed."""
    import argparse
    import sys
    import bench
    started = []
    monkeypatch.setattr(bench, "visible_devices", lambda: 1)
    monkeypatch.setattr("subprocess.Popen", lambda *a, **k: started.append(a) or (_ for _ in ()).throw(AssertionError("no rank may start")))
    rc = bench.self_launch(argparse.Namespace(gpus=8, no_gather=False))
    err = capsys.readouterr().err
    assert rc != 0 and not started and "--gpus 8" in err and "1 HIP device(s) visible" in err
    monkeypatch.setattr(bench, "visible_devices", lambda: 0)
    assert bench.self_launch(argparse.Namespace(gpus=2, no_gather=True)) != 0
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert "WORLD_SIZE=1" in str(exc.value)


@pytest.mark.parametrize("window", [661, 1103, 736, 202, 158, 2203, 2731, 1322, 2735, 5147, 5461, 1486, 3002, 4094])
def test_bluestein_tables_reproduce_the_spectrum(window):
    """Host tables of csrc/kernels_blu.hpp (no device): the kernel's passes restated in NumPy FROM THE LIBRARY'S OWN TABLES -- the
    conjugate chirp, FFT(b) / M stored where the decimation-in-frequency passes leave each bin, the per-pass twiddles -- forward
    DIF passes 0 / 1 / 2, product, conjugate, DIT passes 2 / 1 / 0 -- give |fft(frame)|[0:W/2] / (W/2) (ShortTermFeatures.py:617-621)
    without any permutation.  The kernel executes exactly this index algebra on its LDS buffer."""
    import ctypes
    lib = _ffi.lib()
    info = np.zeros(8, dtype=np.int32)
    off = np.zeros(3, dtype=np.int32)
    size = lib.paa_debug_blu_plan(window, 22050.0, info.ctypes.data_as(_ffi.c_i32p), off.ctypes.data_as(_ffi.c_i32p), None, 0)
    assert size > 0
    blob = np.zeros(size, dtype=np.uint8)
    assert lib.paa_debug_blu_plan(window, 22050.0, info.ctypes.data_as(_ffi.c_i32p), off.ctypes.data_as(_ffi.c_i32p),
                                  blob.ctypes.data_as(ctypes.c_void_p), size) == size
    lg, R0, R1, R2, waves, lds, table_bytes, total = (int(v) for v in info)
    R1, R1B = R1 & 0xff, R1 >> 8                      # a second middle pass (M = 8192 = 16 x 8 x 8 x 8), 0: three passes
    lg, packed = lg & 0xff, bool(lg & 0x100)          # packed: the even window as W / 2 complex points, M >= W - 1
    M, W, Nf = 1 << lg, window, window // 2
    assert R0 * R1 * max(R1B, 1) * R2 == M
    if packed:
        assert W % 2 == 0 and M >= W - 1 and M // 2 < W - 1 and M < W + Nf - 1          # ... and shorter than the direct form would be
    else:
        assert M >= W + Nf - 1 and (M // 2 < W + Nf - 1 or M == 256)      # the smallest power of two that holds it
    assert 1 <= waves <= 16 and lds <= 160 * 1024 and table_bytes % 256 == 0 and total == size
    cplx = lambda o, n: blob[o:o + 16 * n].view(np.float64).reshape(n, 2) @ np.array([1.0, 1j])      # noqa: E731
    S0, S1 = M // R0, M // R0 // R1
    S1B = S1 // R1B if R1B else 0
    L = W // 2 if packed else W                        # elements of the convolved sequence
    chirp = cplx(off[0], L)
    bp = cplx(off[1], M)
    tw = cplx(off[2], (R0 - 1) * S0 + (R1 - 1) * S1 + ((R1B - 1) * S1B if R1B else 0))
    tw0 = tw[:(R0 - 1) * S0].reshape(R0 - 1, S0)
    tw1 = tw[(R0 - 1) * S0:(R0 - 1) * S0 + (R1 - 1) * S1].reshape(R1 - 1, S1)
    tw1b = tw[(R0 - 1) * S0 + (R1 - 1) * S1:].reshape(R1B - 1, S1B) if R1B else None
    n = np.arange(L, dtype=np.int64)
    assert np.allclose(chirp, np.exp(-1j * np.pi * ((n * n) % (2 * L)) / L), rtol=0, atol=1e-14)
    rng = np.random.default_rng(window)
    y = rng.standard_normal(W)
    buf = np.zeros(M, dtype=complex)
    buf[:L] = ((y[0::2] + 1j * y[1::2]) if packed else y) * chirp
    # pass 0 forward: span M, stride S0, output twiddles
    for k in range(S0):
        v = np.fft.fft(buf[k::S0])
        v[1:] *= tw0[:, k]
        buf[k::S0] = v
    # pass 1 forward: span S0, stride S1
    for b in range(M // R1):
        blk, k = divmod(b, S1)
        idx = blk * S0 + k + S1 * np.arange(R1)
        v = np.fft.fft(buf[idx])
        v[1:] *= tw1[:, k]
        buf[idx] = v
    # (four passes: the second middle pass, span S1, stride S1B)
    def pass1b(fwd):
        for b in range(M // R1B):
            blk, k = divmod(b, S1B)
            idx = blk * S1 + k + S1B * np.arange(R1B)
            v = buf[idx].copy()
            if not fwd:
                v[1:] *= tw1b[:, k]
            v = np.fft.fft(v)
            if fwd:
                v[1:] *= tw1b[:, k]
            buf[idx] = v
    if R1B:
        pass1b(True)
    # pass 2 forward, product, conjugate, pass 2 back -- R2 contiguous elements
    for b in range(M // R2):
        idx = b * R2 + np.arange(R2)
        buf[idx] = np.fft.fft(np.conj(np.fft.fft(buf[idx]) * bp[idx]))
    if R1B:
        pass1b(False)
    # pass 1 back: input twiddles
    for b in range(M // R1):
        blk, k = divmod(b, S1)
        idx = blk * S0 + k + S1 * np.arange(R1)
        v = buf[idx].copy()
        v[1:] *= tw1[:, k]
        buf[idx] = np.fft.fft(v)
    # pass 0 back + magnitudes: natural order, bins k < Nf
    out = np.zeros(M, dtype=complex)
    for k in range(S0):
        v = buf[k::S0].copy()
        v[1:] *= tw0[:, k]
        out[k::S0] = np.fft.fft(v)
    ref = np.abs(np.fft.fft(y))[:Nf] / Nf
    if packed:
        # Z[k] = conj(c[k]) conj(v[k]) in place, then the real-FFT recombination of the pairs (k, W/2 - k) with the post-twiddles
        # exp(-2 pi i k / W) that sit right behind the chirp in the blob
        Nc = L
        post = cplx(off[0] + 16 * L, Nc // 2 + 1)
        assert np.allclose(post, np.exp(-2j * np.pi * np.arange(Nc // 2 + 1) / W), rtol=0, atol=1e-14)
        Z = chirp * np.conj(out[:Nc])
        assert np.allclose(Z, np.fft.fft(y[0::2] + 1j * y[1::2]), rtol=0, atol=1e-11)
        got = np.full(Nf, np.nan)
        for k in range(Nc // 2 + 1):
            zk, zm = Z[k], Z[0 if k == 0 else Nc - k]
            e, o = zk + np.conj(zm), -1j * (zk - np.conj(zm))
            got[k] = abs(e + post[k] * o) * 0.5 / Nc
            if k > 0 and Nc - k != k:
                got[Nc - k] = abs(e - post[k] * o) * 0.5 / Nc
        assert (Nc - 1) // S0 < R0 // 2                  # the outputs the last pass forms: q < R0 / 2
    else:
        got = np.abs(out[:Nf]) / Nf
        # the bins the last pass can deliver: k + q S0 < Nf only for q < QMAX of the kernel's Shape
        qmax = {16: 6, 8: 3, 4: 2}[R0]
        assert (Nf - 1) // S0 < qmax
    assert np.max(np.abs(got - ref)) < 1e-13 * max(1.0, ref.max())


def test_bluestein_kernel_takes_the_lengths_with_large_prime_factors():
    """Which windows the Bluestein layout accepts (host side, no device): a prime factor above 13 in the FFT length, at least 64
    bins, convolution length at most 8192; smooth lengths and the register-FFT shapes are declined."""
    lib = _ffi.lib()
    info = np.zeros(8, dtype=np.int32)
    off = np.zeros(3, dtype=np.int32)
    plan = lambda w: lib.paa_debug_blu_plan(w, 16000.0, info.ctypes.data_as(_ffi.c_i32p), off.ctypes.data_as(_ffi.c_i32p), None, 0)      # noqa: E731
    for w, lg in ((661, 10), (1103, 11), (736, 0x100 | 10), (202, 9), (158, 8), (2203, 12), (2731, 12), (683, 10), (2733, 13),
                  (5461, 13), (1322, 11), (1486, 0x100 | 11), (4094, 0x100 | 12), (3002, 0x100 | 12)):
        assert plan(w) > 0 and info[0] == lg, (w, info[0])          # (0x100: the packed form halves the convolution)
    for w in (800, 1024, 2400, 2205, 1323, 4800, 34, 126, 5462 + 1, 9001):      # smooth / too few bins / too long
        assert plan(w) == 0, w


@pytest.mark.parametrize("fs,window", [(16000, 16000), (16000, 8000), (8000, 8000), (44100, 16000), (22050, 16000), (48000, 8000)])
def test_fused_big_window_tables(fs, window):
    """Host tables of the fused three-pass kernel of the 1 s windows (csrc/kernels_wgr.hpp), restated from the library's own mel bank:
    the mel LANE JOBS deal every (filter, bin) of the reference's bank (ShortTermFeatures.py:191-233) to exactly one thread, at most
    sixteen bins of ONE filter per thread, the threads of a filter consecutive and its bins round-robin among them; the chroma gather
    lists (:277-321) are the oracle's, one entry per lane, padded with weight 0."""
    lib = _ffi.lib()
    nfft = window // 2
    job = np.zeros((512, 4), dtype=np.int32)
    fil = np.zeros((40, 2), dtype=np.int32)
    ch_n = np.zeros(12, dtype=np.int32)
    ch_src = np.zeros((12, 64), dtype=np.int32)
    ch_w = np.zeros((12, 64))
    i32 = lambda a: a.ctypes.data_as(_ffi.c_i32p)          # noqa: E731
    sid = lib.paa_debug_wgr_tables(float(fs), window, i32(job), i32(fil), i32(ch_n), i32(ch_src), _ffi.as_f64p(ch_w))
    assert sid == (1 if window == 16000 else 2)
    dense = np.zeros((40, nfft))
    _ffi.check(lib.paa_debug_mel_bank(float(fs), nfft, _ffi.as_f64p(dense)))
    # the flat weight table: filter after filter, the bins of its support
    lo = [int(np.flatnonzero(dense[m])[0]) for m in range(40)]
    cnt = [int(np.flatnonzero(dense[m])[-1]) - lo[m] + 1 for m in range(40)]
    off = np.concatenate([[0], np.cumsum(cnt)])
    seen = np.zeros(int(off[-1]), dtype=np.int32)
    threads = 448
    per_lane = 0
    for m in range(40):
        first, nl = fil[m]
        assert nl >= 1 and (m == 0 and first == 0 or first == fil[m - 1][0] + fil[m - 1][1])
        for i in range(nl):
            kb, eb, st, n = job[first + i]
            assert st == nl and 1 <= n <= 16 and kb == lo[m] + i and eb == off[m] + i
            idx = eb + st * np.arange(n)
            assert idx[-1] < off[m + 1] and idx[-1] + st >= off[m + 1]          # ... and no bin of the filter is left over
            seen[idx] += 1
            per_lane = max(per_lane, n)
    assert fil[39][0] + fil[39][1] <= threads and np.all(seen == 1)
    assert np.all(job[fil[39][0] + fil[39][1]:threads, 3] == 0)                   # threads without a job
    # sixteen bins per thread at most, and no more than the smallest count that lets the filters' threads fit the workgroup
    assert per_lane == min(L for L in range(1, 17) if sum(-(-c // L) for c in cnt) <= threads)
    o_src, o_w, o_cls, _ = O.chroma_gather(fs, nfft)
    for c in range(12):
        sel = np.flatnonzero(np.asarray(o_cls) == c)
        assert ch_n[c] == len(sel) <= 64
        assert np.array_equal(ch_src[c, :len(sel)], np.asarray(o_src)[sel]) and np.array_equal(ch_w[c, :len(sel)], np.asarray(o_w)[sel])
        assert np.all(ch_w[c, len(sel):] == 0.0) and np.all(ch_src[c, len(sel):] == 0)


def test_fused_big_window_kernel_declines_what_it_cannot_hold():
    """A window of the fused kernel at a sampling rate whose mel bank needs more than sixteen bins per thread (16 000 samples at 8 kHz:
    two seconds) is declined (-1: the plan keeps the in-place LDS transform of csrc/kernels_wg.hpp); other windows are not its business."""
    lib = _ffi.lib()
    assert lib.paa_debug_wgr_tables(8000.0, 16000, None, None, None, None, None) == -1
    assert lib.paa_debug_wgr_tables(16000.0, 12000, None, None, None, None, None) == 0
    assert lib.paa_debug_wgr_tables(16000.0, 16000, None, None, None, None, None) == 1


@pytest.mark.parametrize("frames", [[1199], [7199], [23] * 7, [1, 2, 3, 500, 4, 1], [119] * 200, [3]])
def test_fused_big_window_runs(frames):
    """Runs of consecutive frames of the fused kernel (one workgroup per CU walks runs b, b + grid, ...): every frame of every clip
    exactly once, in order, no run longer than the per-CU share of all frames, the runs of a clip within one frame of each other."""
    lib = _ffi.lib()
    f = np.asarray(frames, dtype=np.int64)
    n = ctypes.c_int64(0)
    _ffi.check(lib.paa_debug_wgr_runs(f.ctypes.data_as(_ffi.c_i64p), len(f), 256, None, 0, ctypes.byref(n)))
    runs = np.zeros((n.value, 3), dtype=np.int32)
    _ffi.check(lib.paa_debug_wgr_runs(f.ctypes.data_as(_ffi.c_i64p), len(f), 256, runs.ctypes.data_as(_ffi.c_i32p), n.value, ctypes.byref(n)))
    share = max(1, -(-int(f.sum()) // 256))
    pos = {}
    for clip, t0, cnt in runs:
        assert cnt >= 1 and t0 == pos.get(clip, 0) and cnt <= share
        pos[clip] = t0 + cnt
    assert [pos.get(c, 0) for c in range(len(f))] == list(f)
    for c in range(len(f)):
        lens = runs[runs[:, 0] == c, 2]
        assert lens.max() - lens.min() <= 1
    if len(f) == 1 and f[0] >= 256:
        assert len(runs) == -(-int(f[0]) // share) <= 256

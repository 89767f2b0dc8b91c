"""Parity of the HIP path (through the C ABI) against the goldens and the CPU oracle.  -m gpu.

Two gates, both applied by assert_parity():

* the CONTRACT (north_star: 1e-4 relative): |d| <= 1e-4*|ref| + 1e-6*max|ref_row| + 1e-9, see
  paa_oracle.mixed_tolerance_violations;
* a TIGHT engineering gate that keeps the kernels honest (the observed error is ~1e-14, scripts/parity_margin.py):
  |d| <= 1e-9*|ref| + 1e-10*scale(row) + 1e-12, where a delta row inherits the scale of its base row (a delta is a
  difference of two base values, so its error lives on the base row's scale) and the 13 MFCC rows share one scale.
  Two documented exceptions, both places where the REFERENCE's own value is a function of round-off:
  (i) MFCC rows of the frames paa_oracle.ill_conditioned_mfcc_frames flags (log10 of an empty mel band) -- a BOUNDED
  exception: the flagged frames are the digitally silent ones (whose MFCCs are then held to the analytic vector at 1e-9,
  checks.silent_mfcc) plus at most 1e-3 of the frames (checks.IllInfo.budget_ok; a test built on an ill-conditioned input
  states its own allowance);
  (ii) spectral spread (:80) is the square root of a cancelling sum: on digitally silent frames the reference itself
  returns sqrt(round-off ~1e-17) ~ 3e-9 (oracle/paa_oracle.c differs from it by as much), so rows 4 / 38 get
  1e-7*scale -- and 1e-6*scale on the digitally silent frames themselves (row 38: and their successors) when the window has a
  prime factor above 13: there pocketfft itself runs a chirp convolution and leaves 1e-17 |X[0]| in every bin of a constant
  frame (spread 4e-8 for 1103 samples at 22.05 kHz), while the Bluestein kernel's shortcut gives the analytic 1e-19 (round 6).
  ZCR and roll-off are integer-valued outcomes of exact / floating comparisons: ZERO flips are allowed."""
import os

import numpy as np
import pytest

import checks
import paa_oracle as O
from conftest import golden_files, golden_id, load_golden
from pyaudioanalysis_amd import MidTermFeatures, ShortTermFeatures, _ffi
from synth import synth_clip

pytestmark = pytest.mark.gpu

REL, ROW, FLOOR = 1e-4, 1e-6, 1e-9                 # the contract
T_REL, T_ROW, T_FLOOR = 1e-9, 1e-10, 1e-12          # the tight gate
T_ROW_SPREAD = 1e-7
T_ROW_SPREAD_SILENT = 1e-6         # silent frames of windows the reference transforms by chirp convolution (see (ii) above)
DISCRETE_ROWS = (0, 7, 34, 41)      # zcr, roll-off and their deltas
SPREAD_ROWS = (4, 38)


MFCC_ALL = [r + b for b in (0, 34) for r in O.MFCC_ROWS]


def tight_violations(got, ref, ill=None, silent=None):
    """-> (count, bool mask) of entries outside the tight gate (see the module docstring).  silent: bool mask of digitally silent
    frames whose spread rows get T_ROW_SPREAD_SILENT (the caller passes it only for chirp-convolution windows)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    feature_rows = ref.ndim == 2 and ref.shape[0] in (34, 68)
    if ref.ndim == 2:
        scale = np.max(np.abs(ref), axis=1, keepdims=True)
        if ref.shape[0] % 34 == 0 and ref.shape[0] // 34 in (1, 2, 4):
            nb = ref.shape[0] // 34
            if nb in (2, 4):          # delta rows (short-term) / delta-mean, std rows (mid-term): base-row scale
                base = scale[:34].copy()
                for blk in range(1, nb):
                    scale[blk * 34:(blk + 1) * 34] = np.maximum(scale[blk * 34:(blk + 1) * 34], base)
            for blk in range(nb):
                rows = [blk * 34 + r for r in O.MFCC_ROWS]
                scale[rows] = scale[rows].max()
    else:
        scale = np.max(np.abs(ref))
    row_tol = np.full_like(scale, T_ROW, dtype=np.float64) if ref.ndim == 2 else T_ROW
    if ref.ndim == 2 and ref.shape[0] % 34 == 0 and ref.shape[0] // 34 in (1, 2, 4):
        for blk in range(ref.shape[0] // 34):
            row_tol[blk * 34 + 4] = T_ROW_SPREAD
    tol = T_REL * np.abs(ref) + row_tol * scale + T_FLOOR
    if silent is not None and feature_rows and silent.any() and len(silent) == ref.shape[1]:
        after = silent.copy()
        after[1:] |= silent[:-1]
        tol[4, silent] = T_REL * np.abs(ref[4, silent]) + T_ROW_SPREAD_SILENT * scale[4] + T_FLOOR
        if ref.shape[0] == 68:
            tol[38, after] = T_REL * np.abs(ref[38, after]) + T_ROW_SPREAD_SILENT * scale[38] + T_FLOOR
    bad = np.abs(got - ref) > tol
    bad |= ~np.isfinite(got)
    if feature_rows and ill is not None and ill.any():
        rows = [r for r in MFCC_ALL if r < ref.shape[0]]
        sub = bad[rows]
        sub[:, ill] = False
        bad[rows] = sub
    elif ref.ndim == 2 and ref.shape[0] == 136 and ill is not None and ill.any():
        # mid-term statistics of MFCC rows that contain a flagged frame: contract gate only
        for blk in range(4):
            bad[[blk * 34 + r for r in O.MFCC_ROWS]] = False
    return int(bad.sum()), bad


def _chirp_window(window):
    """True when the window's FFT length has a prime factor above 13: the Bluestein kernel's windows (and pocketfft's own
    chirp convolution for the large primes among them)"""
    n = int(window) // 2 if int(window) % 2 == 0 else int(window)
    for f in (2, 3, 5, 7, 11, 13):
        while n % f == 0:
            n //= f
    return n != 1


def _exact_zero_kernel(fs, window, step):
    """True when the kernel family that takes this window keeps the exact-zero property of DESIGN section 2 (a digitally silent
    frame's spectrum is exactly [2|c|, 0, 0, ...]): every family but the Stockham kernels (st_generic, the big-window passes)."""
    plan = _ffi.Plan(np.array([0, 4 * int(window) + 4 * int(step)], dtype=np.int64), fs, int(window), int(step), deltas=False)
    name = plan.kernel_name
    plan.destroy()
    return not ("generic" in name or "big" in name)


def assert_parity(got, ref, what="", ill=None, tight=True, sig=None, max_other_share=1e-3, max_other_abs=0):
    """sig = (signal, fs, window, step): derive the ill-conditioned-frame information from the input (checks.ill_info).
    ill: a checks.IllInfo, or a plain bool mask of frames whose MFCCs are round-off-determined in the reference itself
    (paa_oracle.ill_conditioned_mfcc_frames); there the MFCC rows get 1e-5 of the group scale (contract) and are
    exempt from the tight gate.  That exception is BOUNDED: with an IllInfo, flagged frames that are not
    explained by digital silence may be at most max(max_other_abs, max_other_share x frames) -- tests whose input is built
    to be ill-conditioned say so -- and the MFCC rows of the digitally silent frames themselves are held to the analytic
    vector (checks.silent_mfcc) at 1e-9 on every kernel that produces exact zeros for them."""
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert got.dtype == np.float64 and got.flags["C_CONTIGUOUS"]
    if ill is None and sig is not None:
        ill = checks.ill_info(*sig)
    info = ill if isinstance(ill, checks.IllInfo) else None
    if info is not None:
        ill = info.mask
        if ref.ndim == 2 and ref.shape[0] in (34, 68) and ref.shape[1] == len(info.mask):
            assert info.budget_ok(max_other_share, max_other_abs), \
                "%s: the 1e-5 MFCC exception would cover %s (allowed: %g of the frames / %d)" % (
                    what, info.counts(), max_other_share, max_other_abs)
            if info.silent.any() and (sig is None or _exact_zero_kernel(*sig[1:4])):
                nsil = checks.silent_mfcc_violations(got, info)
                assert nsil == 0, "%s: %d MFCC entries of digitally silent frames differ from the analytic value" % (what, nsil)
    nbad, bad = checks.contract_violations(got, ref, ill)
    if nbad:
        idx = np.argwhere(bad)[:8]
        detail = ", ".join("[%s]=%.6g vs %.6g" % (tuple(i), got[tuple(i)], ref[tuple(i)]) for i in idx)
        raise AssertionError("%s: %d entries outside the 1e-4 contract: %s" % (what, nbad, detail))
    if tight:
        silent = info.silent if (info is not None and sig is not None and _chirp_window(sig[2])) else None
        nt, tb = tight_violations(got, ref, ill, silent)
        if nt:
            idx = np.argwhere(tb)[:8]
            detail = ", ".join("[%s]=%.17g vs %.17g" % (tuple(i), got[tuple(i)], ref[tuple(i)]) for i in idx)
            raise AssertionError("%s: %d entries outside the tight gate (discrete rows: any flip counts): %s"
                                 % (what, nt, detail))


@pytest.mark.parametrize("path", golden_files("st"), ids=golden_id)
def test_short_term_golden(gpu_lib, path):
    g = load_golden(path)
    F, names = ShortTermFeatures.feature_extraction(g["signal"], g["fs"], g["window"], g["step"], g["deltas"])
    assert names == [str(s) for s in g["names"]]
    mono = O.stereo_to_mono(g["signal"]) if g["signal"].ndim == 2 else g["signal"]
    # quarter_tone_with_zeros is BUILT to be ill-conditioned (a tone exactly on an FFT bin: every other band holds round-off only)
    share = 1.0 if golden_id(path) == "quarter_tone_with_zeros" else 1e-3
    assert_parity(F, g["features"], golden_id(path), sig=(mono, g["fs"], g["window"], g["step"]), max_other_share=share)


@pytest.mark.parametrize("path", golden_files("mid"), ids=golden_id)
def test_mid_term_golden(gpu_lib, path):
    g = load_golden(path)
    mid, st, names = MidTermFeatures.mid_feature_extraction(g["signal"], g["fs"], g["mid_window"], g["mid_step"],
                                                            g["window"], g["step"])
    assert names == [str(s) for s in g["names"]]
    sig = (g["signal"], g["fs"], g["window"], g["step"])
    assert_parity(st, g["features"], "short", sig=sig)
    assert_parity(mid, g["mid"], "mid", sig=sig)


@pytest.mark.parametrize("path", golden_files("spec"), ids=golden_id)
def test_spectrogram_chromagram_golden(gpu_lib, path, capsys):
    g = load_golden(path)
    S, t_ax, f_ax = ShortTermFeatures.spectrogram(g["signal"], g["fs"], g["window"], g["step"])
    assert "(%d, %d)" % g["specgram"].shape in capsys.readouterr().out       # the reference prints the shape
    assert_parity(S, g["specgram"], "specgram")
    assert np.array_equal(np.array(t_ax), g["spec_time"]) and np.array_equal(np.array(f_ax), g["spec_freq"])
    C, ct_ax, names = ShortTermFeatures.chromagram(g["signal"], g["fs"], g["window"], g["step"])
    assert_parity(C, g["chromagram"], "chromagram")
    assert np.array_equal(np.array(ct_ax), g["chroma_time"])
    assert names == [str(s) for s in g["chroma_names"]]


@pytest.mark.parametrize("path", golden_files("spec_only"), ids=golden_id)
def test_spectrogram_only_golden(gpu_lib, path, capsys):
    """windows too small for the reference's mel bank / chroma tables at their rate: its spectrogram still works"""
    g = load_golden(path)
    S, t_ax, f_ax = ShortTermFeatures.spectrogram(g["signal"], g["fs"], g["window"], g["step"])
    assert "(%d, %d)" % g["specgram"].shape in capsys.readouterr().out
    assert_parity(S, g["specgram"], "specgram")
    assert np.array_equal(np.array(t_ax), g["spec_time"]) and np.array_equal(np.array(f_ax), g["spec_freq"])


@pytest.mark.parametrize("fs,window,step,seconds", [
    (16000, 800, 400, 4.0),       # headline shape
    (16000, 800, 800, 2.0),       # the reference's own pytest shape (no overlap)
    (16000, 400, 160, 1.0),       # 25 ms / 10 ms
    (16000, 801, 401, 1.0),       # odd window: full-length complex FFT path
    (16000, 1024, 512, 1.5),      # power of two
    (22050, 1103, 441, 1.0),      # a prime window -> Bluestein kernel (convolution length 2048)
    (44100, 1102, 441, 1.0),      # config 5 (2 * 19 * 29)
    (8000, 400, 200, 1.5),
    (8000, 800, 400, 2.0),        # fast kernel, run-time mel list lengths (wider filters in bins)
    (7000, 800, 400, 2.0),        # fast kernel, padded mel lists reach the last bins (clamped index path)
    (44100, 800, 400, 1.0),       # fast kernel at another sampling rate
    (22050, 800, 800, 1.0),       # fast kernel, step 800
    (48000, 2400, 1200, 1.0),
])
def test_oracle_parity_seeded(gpu_lib, fs, window, step, seconds):
    x = synth_clip(100 + window, int(seconds * fs), fs=fs)
    for deltas in (True, False):
        ref, _ = O.feature_extraction(x, fs, window, step, deltas)
        got, _ = ShortTermFeatures.feature_extraction(x, fs, window, step, deltas)
        assert_parity(got, ref, "%d/%d@%d deltas=%s" % (window, step, fs, deltas), sig=(x, fs, window, step))


def test_float64_input_matches_oracle(gpu_lib):
    xs = synth_clip(77, 30000, fs=44100, stereo=True)
    mono = O.stereo_to_mono(xs)
    assert mono.dtype == np.float64
    ref, _ = O.feature_extraction(mono, 44100, 1102, 441)
    got, _ = ShortTermFeatures.feature_extraction(mono, 44100, 1102, 441)
    assert_parity(got, ref, "stereo->mono f64", sig=(mono, 44100, 1102, 441))
    # other dtypes go through np.double() like the reference (:567)
    x32 = synth_clip(78, 8000).astype(np.int32)
    ref, _ = O.feature_extraction(x32, 16000, 800, 400)
    got, _ = ShortTermFeatures.feature_extraction(x32, 16000, 800, 400)
    assert_parity(got, ref, "int32 input", sig=(x32, 16000, 800, 400))


def test_known_answers(gpu_lib):
    F, _ = ShortTermFeatures.feature_extraction(np.zeros(2000, dtype=np.int16), 16000, 800, 400)
    assert F.shape == (68, 4)
    assert abs(F[8, 0] + 99.00180475) < 1e-6                    # mfcc_1 of digital silence
    others = np.delete(F, 8, axis=0)
    assert np.all(np.abs(others) < 1e-9)
    x = synth_clip(5, 16000)
    F, _ = ShortTermFeatures.feature_extraction(x, 16000, 800, 400)
    assert F[6, 0] == 0.0 and np.all(F[34:, 0] == 0.0)         # first frame: flux 0, deltas 0
    assert np.allclose(F[34:, 1:], F[:34, 1:] - F[:34, :-1], rtol=0, atol=0)    # deltas are row differences


def test_reference_pytests_shapes(gpu_lib):
    """pytests/test_feature_extraction.py:10-29 re-pointed at the drop-in (1 s and 5 s clips)."""
    fs = 16000
    x = synth_clip(1, fs)
    F, names = ShortTermFeatures.feature_extraction(x, fs, 0.050 * fs, 0.050 * fs)
    assert F.shape[1] == 20 and F.shape[0] == len(names)
    x = synth_clip(2, 5 * fs)
    mt, st, mt_names = MidTermFeatures.mid_feature_extraction(x, fs, 1 * fs, 1 * fs, 0.05 * fs, 0.05 * fs)
    assert mt.shape[1] == 5 and mt.shape[0] == len(mt_names)


def test_errors_keep_reference_exception_types(gpu_lib):
    x = np.arange(4000, dtype=np.int16)
    with pytest.raises(ValueError):
        ShortTermFeatures.feature_extraction(x, 16000, 2 * 97, 97)
    with pytest.raises(IndexError):
        ShortTermFeatures.feature_extraction(x, 16000, 2 * 98, 98)
    ShortTermFeatures.feature_extraction(x, 16000, 2 * 99, 99)
    with pytest.raises(IndexError):                              # mel bank bins beyond num_fft (:230)
        ShortTermFeatures.feature_extraction(x, 4000, 800, 400)


def test_batch_equals_single_clips(gpu_lib):
    """Ragged batch: clips of different lengths, including one of exactly one window."""
    lens = [800, 1199, 16000, 7777, 48000, 2000]
    clips = [synth_clip(300 + i, n) for i, n in enumerate(lens)]
    res, names = ShortTermFeatures.feature_extraction_batch(clips, 16000, 800, 400)
    assert len(res) == len(clips)
    for c, r in zip(clips, res):
        single, _ = ShortTermFeatures.feature_extraction(c, 16000, 800, 400)
        assert np.array_equal(single, r)                          # same kernels, same tiles -> bit equal
        ref, _ = O.feature_extraction(c, 16000, 800, 400)
        assert_parity(r, ref, "batch clip", sig=(c, 16000, 800, 400))
    mids, sts, mnames = MidTermFeatures.mid_feature_extraction_batch(clips[2:], 16000, 16000, 16000, 800, 400,
                                                                    return_short=True)
    for c, m, s in zip(clips[2:], mids, sts):
        ref_mid, ref_st, _ = O.mid_feature_extraction(c, 16000, 16000, 16000, 800, 400)
        assert_parity(s, ref_st, "batch short", sig=(c, 16000, 800, 400))
        assert_parity(m, ref_mid, "batch mid", sig=(c, 16000, 800, 400))


@pytest.mark.parametrize("minutes", [10, 60])
def test_size_independent_properties_at_scale(gpu_lib, minutes):
    """A 10-minute and the full 1-hour clip of BASELINE config 2 (143 999 frames): checks that do not need the
    oracle to run that long."""
    fs, W, S = 16000, 800, 400
    base = synth_clip(9, 60 * fs)
    x = np.tile(base, minutes)
    F, _ = ShortTermFeatures.feature_extraction(x, fs, W, S)
    T = (len(x) - W) // S + 1
    assert F.shape == (68, T) and np.all(np.isfinite(F))
    # deltas are exact row differences
    assert np.array_equal(F[34:, 1:], F[:34, 1:] - F[:34, :-1])
    # ranges: zcr, roll-off in [0,1]; entropies in [0, log2(10)]; chroma sums to <= 1
    assert F[0].min() >= 0 and F[0].max() <= 1 and F[7].min() >= 0 and F[7].max() < 1
    assert F[2].max() <= np.log2(10) + 1e-9 and F[5].max() <= np.log2(10) + 1e-9
    assert F[21:33].sum(axis=0).max() <= 1 + 1e-9
    # tile independence: any window of frames equals the same frames computed from a sub-clip with the
    # SAME normalisation constants -> use the oracle on a few frame ranges with the clip-global constants
    xn = O.normalize_clip(x)
    tab = O.Tables(fs, W)
    # (the contract exactly as assert_parity applies it: the row term is 1e-6 of the ROW's scale over the whole clip -- the 13 MFCC
    # rows share one -- not of the one column that is compared here)
    scale = np.max(np.abs(F[:34]), axis=1)
    scale[list(O.MFCC_ROWS)] = scale[list(O.MFCC_ROWS)].max()
    for t in (0, 1, 31, 32, 33, 5000, T - 1):
        fr = xn[t * S:t * S + W]
        X = O.magnitude_spectrum(fr, tab.nfft)
        Xp = X if t == 0 else O.magnitude_spectrum(xn[(t - 1) * S:(t - 1) * S + W], tab.nfft)
        v = O.frame_vector(fr, X, Xp, tab)
        bad = np.abs(F[:34, t] - v) > REL * np.abs(v) + ROW * scale + FLOOR
        assert not bad.any(), "frame %d rows %s" % (t, np.flatnonzero(bad))
    # periodicity: the clip is 10 repeats of a 60 s block and 60 s is a multiple of the step, so
    # frames one period apart see identical samples and identical clip constants
    per = 60 * fs // S
    a, b = F[:34, 5:per - 5], F[:34, per + 5:2 * per - 5]
    assert np.allclose(a, b, rtol=1e-9, atol=1e-12)


def test_rccl_gather_world_size_1(gpu_lib):
    """The RCCL path with a single rank: communicator init, gather (device copy into the root buffer), barrier."""
    from pyaudioanalysis_amd import distributed as D
    comm = D.RcclGather(1, 0, lambda payload: payload)
    try:
        clips = [synth_clip(500 + i, n) for i, n in enumerate([4000, 16000, 2400])]
        res = D.extract_sharded(clips, 16000, 800, 400, True, 1, 0, comm)
        comm.barrier()
        assert len(res) == 3
        for c, r in zip(clips, res):
            single, _ = ShortTermFeatures.feature_extraction(c, 16000, 800, 400)
            assert np.array_equal(single, r)
        # the chunked pipeline (piece k gathered while piece k+1 computes), restart files written and reloaded (slices of the
        # uploaded block are sent piece by piece), and the cheap gather of the (136, M) mid-term matrices
        clips = [synth_clip(520 + i, n) for i, n in enumerate([4000, 16000, 2400, 48000, 9000, 1200, 32000])]
        import tempfile
        with tempfile.TemporaryDirectory() as rdir:
            first = D.extract_sharded(clips, 16000, 800, 400, False, 1, 0, comm, chunks=3, restart_dir=rdir)
            again = D.extract_sharded(clips, 16000, 800, 400, False, 1, 0, comm, chunks=3, restart_dir=rdir)
        mids = D.extract_sharded(clips, 16000, 800, 400, True, 1, 0, comm, chunks=2, gather="mid", mid_window=16000,
                                 mid_step=8000)
        for c, a, b, m in zip(clips, first, again, mids):
            single, _ = ShortTermFeatures.feature_extraction(c, 16000, 800, 400, deltas=False)
            assert np.array_equal(single, a) and np.array_equal(single, b)
            mid, _, _ = MidTermFeatures.mid_feature_extraction(c, 16000, 16000, 8000, 800, 400)
            assert np.array_equal(mid, m)
    finally:
        comm.close()


def test_delta_rows_reformed_on_the_device_equal_the_68_row_plan(gpu_lib):
    """paa_dev_expand_deltas: [34][T_c] base slabs -> [68][T_c] slabs, bit for bit what a 68-row plan stores (what lets a
    sharded job ship the 34 base rows only, ShortTermFeatures.py:668-680) -- clips of 1 frame, of a few frames, of more than one
    2048-frame tile, two window families; and a second call with another batch shape (the cached tile list is rebuilt)."""
    from pyaudioanalysis_amd import _ffi
    for fs, W, S, lens in ((16000, 800, 400, [800, 1200, 16000, 400 * 5000 + 800, 2400]), (44100, 2205, 1102, [2205, 44100 * 3])):
        clips = [synth_clip(8100 + i, n, fs) for i, n in enumerate(lens)]
        offs = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        d_in = _ffi.DeviceBuffer.from_host(np.concatenate(clips))
        p34 = _ffi.Plan(offs, fs, W, S, deltas=False)
        p68 = _ffi.Plan(offs, fs, W, S, deltas=True)
        d34 = _ffi.DeviceBuffer(p34.out_doubles * 8)
        d68 = _ffi.DeviceBuffer(p68.out_doubles * 8)
        d_exp = _ffi.DeviceBuffer(p68.out_doubles * 8)
        p34.execute(d_in, d34)
        p68.execute(d_in, d68)
        frames = np.array([(n - W) // S + 1 for n in lens], dtype=np.int64)
        _ffi.check(gpu_lib.paa_dev_expand_deltas(d34.ptr, _ffi.as_i64p(frames), len(frames), d_exp.ptr))
        _ffi.sync()
        want = d68.to_host(np.float64, p68.out_doubles)
        got = d_exp.to_host(np.float64, p68.out_doubles)
        assert np.array_equal(got, want)
        pos = 0
        for t in frames:
            slab = got[pos:pos + 68 * int(t)].reshape(68, int(t))
            assert np.all(slab[34:, 0] == 0.0) and np.array_equal(slab[34:, 1:], slab[:34, 1:] - slab[:34, :-1])
            pos += 68 * int(t)
        p34.destroy()
        p68.destroy()


def test_driver_smoke_entry(gpu_lib, capsys):
    """__graft_entry__.smoke() -- what the driver runs before the bench: one short- + mid-term extraction against both oracles."""
    import importlib
    smoke = importlib.import_module("__graft_entry__").smoke
    smoke()
    assert "smoke ok" in capsys.readouterr().out


def test_big_window_kernel_choice(gpu_lib):
    """Windows beyond the one-wave kernels: the transform runs in ONE WORKGROUP's LDS when it fits (kernels_wg.hpp: up to 10 000
    complex points made of 2, 3, 5, 7, 11, 13); longer ones (up to 32 768 points) as r0 <= 8 sub-transforms whose first pass runs
    straight from the samples (wg_split_kernel); the rest through HBM scratch (kernels_big.hpp).  No window falls back to the CPU."""
    def name(fs, w, s, mode=0):
        plan = _ffi.Plan(np.array([0, 4 * w], dtype=np.int64), fs, w, s, deltas=False, mode=mode)
        try:
            return plan.kernel_name
        finally:
            plan.destroy()
    assert name(16000, 16000, 8000) == "st_wgr_20x20x20"               # music_thumbnailing's 1 s window (audioSegmentation.py:1137): fused three-pass kernel
    assert name(16000, 8000, 4000, mode=1) == "spectrogram_wgr_10x20x20"
    assert name(16000, 8000, 4000, mode=2) == "chromagram_wgr_10x20x20"
    assert name(16000, 12000, 4000, mode=1) == "spectrogram_wg_lds_fft"    # 6000 points: the in-place LDS transform of round 5
    assert name(8000, 16000, 8000) == "st_wg_lds_fft"                  # 2 s windows at 8 kHz: more mel bins than the fused kernel's lane jobs hold
    assert name(8000, 8000, 4000) == "st_wgr_10x20x20"                 # 1 s at 8 kHz
    assert name(16000, 9009, 4500) == "st_wg_lds_fft"                  # odd: 9009 = 7 x 9 x 11 x 13 real points, 144 KB of LDS
    assert name(44100, 44100, 22050) == "st_wgs_12x3675"               # 1 s at 44.1 kHz: real-input split, 6 independent transforms of 3675 points on register passes (kernels_wgs.hpp)
    assert name(22050, 22050, 11025, mode=1) == "spectrogram_wgs_6x3675"   # 1 s at 22.05 kHz: 3 of them
    assert name(44100, 22050, 11025, mode=2) == "chromagram_wgs_6x3675"
    assert name(32000, 32000, 16000) == "st_wgs_8x4000"                # 1 s at 32 kHz: 4 transforms of 4000 points (8 x 20 x 25)
    assert name(48000, 24000, 12000) == "st_wgs_6x4000"                # 0.5 s at 48 kHz
    assert name(48000, 48000, 24000, mode=1) == "spectrogram_wgs_12x4000"      # 1 s at 48 kHz: 6 transforms of 4000 points
    assert name(44100, 11025, 5000, mode=2) == "chromagram_wg_split_fft"       # odd: 11 025 real points, 3 x 3675
    assert name(16000, 65536, 32768) == "st_wg_split_fft"              # the largest table: 32 768 points = 8 x 4096
    assert name(16000, 80000, 40000) == "big_window_hbm_passes"        # 40 000 points: beyond the two-level twiddle table
    assert name(16000, 9001, 4500) == "big_window_hbm_passes"          # prime


@pytest.mark.parametrize("kind,fs,window,step,seconds,deltas", [
    ("i16", 16000, 16000, 8000, 12.0, True),     # the music_thumbnailing shape, 23 frames
    ("stereo", 16000, 8000, 4000, 9.0, True),    # interleaved stereo samples summed in the loads
    ("f64", 22050, 11000, 5000, 6.0, False),     # 5500 complex points = 4 x 5 x 5 x 5 x 11, float64 samples
    ("f64", 8000, 8000, 8000, 9.0, False),       # 1 s at 8 kHz, float64 samples
    ("i16", 16000, 9009, 3000, 4.0, True),       # odd window: real points, radices 13 11 7 3 3
    ("i16", 16000, 20000, 10000, 6.0, False),    # the edge of the LDS: 10 000 complex points = 160 000 bytes
    ("i16", 44100, 44100, 22050, 6.0, True),     # kernels_wgs.hpp: 12 x 3675 samples (tasks {1,2} {3,4} {5,packed}), row not staged
    ("stereo", 44100, 44100, 44100, 5.0, False), # ... interleaved stereo, no overlap
    ("stereo", 48000, 48000, 24000, 4.0, False), # 24 000 = 6 x 4000 (radices 8 4 5 5 5), interleaved stereo
    ("f64", 22050, 22050, 7000, 3.0, True),      # kernels_wgs.hpp: 6 x 3675 samples (tasks {1,2} {packed}), float64 samples
    ("i16", 44100, 11025, 5000, 2.0, False),     # odd window of 11 025 REAL points through the split transform
    ("i16", 16000, 32000, 16000, 9.0, False),    # 16 000 points = 4 x 4000 (tasks {0} {1,3} {2})
    ("f64", 16000, 65536, 30000, 14.0, False),   # 32 768 points = 8 x 4096: radix-16 passes (512-thread instance)
])
def test_workgroup_lds_kernel_full_matrix(gpu_lib, kind, fs, window, step, seconds, deltas):
    """kernels_wg.hpp against the NumPy oracle, every frame and row, every sample type (contract + tight gate)."""
    from test_ct_kernels_gpu import make_signal
    sig, mono = make_signal(kind, 7000 + window, seconds, fs)
    ref, _ = O.feature_extraction(mono, fs, window, step, deltas)
    got, _ = ShortTermFeatures.feature_extraction(sig, fs, window, step, deltas)
    assert_parity(got, ref, "%s %d/%d@%d" % (kind, window, step, fs), sig=(mono, fs, window, step))
    if deltas:
        assert np.array_equal(got[34:, 1:], got[:34, 1:] - got[:34, :-1]) and np.all(got[34:, 0] == 0.0)


@pytest.mark.parametrize("kind,fs,window,step,seconds,mode", [
    ("i16", 16000, 16000, 8000, 610.0, 0),       # ten minutes: 1 219 frames = 256 runs of 4 / 5 frames, a halo transform in front of 255 of them
    ("f64", 16000, 16000, 16000, 40.0, 0),       # 1 s / 1 s (no overlap), float64 samples
    ("stereo", 44100, 16000, 5000, 30.0, 0),     # the window at another rate: other mel lane jobs (6 bins per thread), stereo samples
    ("stereo", 16000, 16000, 8000, 30.0, 1),     # spectrogram rows straight from the registers
    ("f64", 16000, 16000, 4000, 20.0, 2),        # chromagram rows
    ("i16", 8000, 8000, 2000, 300.0, 0),         # 10 x 20 x 20 on a five-minute clip
])
def test_fused_three_pass_kernel(gpu_lib, kind, fs, window, step, seconds, mode, capsys):
    """kernels_wgr.hpp against the NumPy oracle on clips long enough for every workgroup to own a run that starts inside the clip (the
    flux of a run's first frame comes from a halo transform), every sample type and mode."""
    from test_ct_kernels_gpu import make_signal
    sig, mono = make_signal(kind, 9100 + window + mode, seconds, fs)
    if mode == 0:
        ref, _ = O.feature_extraction(mono, fs, window, step, True)
        got, _ = ShortTermFeatures.feature_extraction(sig, fs, window, step, True)
        assert_parity(got, ref, "%s %d/%d@%d" % (kind, window, step, fs), sig=(mono, fs, window, step))
        assert np.array_equal(got[34:, 1:], got[:34, 1:] - got[:34, :-1]) and np.all(got[34:, 0] == 0.0)
    elif mode == 1:
        got, _, _ = ShortTermFeatures.spectrogram(sig, fs, window, step)
        capsys.readouterr()
        ref, _, _ = O.spectrogram(mono, fs, window, step)
        assert_parity(np.ascontiguousarray(got.T), np.ascontiguousarray(ref.T), "fused spectrogram")
    else:
        got, _, _ = ShortTermFeatures.chromagram(sig, fs, window, step)
        ref, _, _ = O.chromagram(mono, fs, window, step)
        assert_parity(np.ascontiguousarray(got.T), np.ascontiguousarray(ref.T), "fused chromagram")


def test_fused_three_pass_kernel_many_clips(gpu_lib):
    """More runs than workgroups (600 short clips: a workgroup walks runs b, b + grid, ...) and silent clips among them: bit-identical
    to the single-clip calls, silent frames at the analytic values."""
    fs, W, S = 16000, 16000, 8000
    rng = np.random.default_rng(77)
    clips = []
    for i in range(600):
        n = int(rng.integers(W, 4 * W))
        clips.append(np.full(n, 1234, dtype=np.int16) if i % 97 == 5 else synth_clip(8800 + (i % 13), n, fs))
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, W, S, deltas=False)
    for i in (0, 5, 17, 102, 255, 256, 257, 511, 599):
        single, _ = ShortTermFeatures.feature_extraction(clips[i], fs, W, S, deltas=False)
        assert np.array_equal(single, res[i]), i
    ref, _ = O.feature_extraction(clips[5], fs, W, S, False)
    assert_parity(res[5], ref, "silent clip in a batch", sig=(clips[5], fs, W, S))
    ref, _ = O.feature_extraction(clips[300], fs, W, S, False)
    assert_parity(res[300], ref, "clip 300 of a batch", sig=(clips[300], fs, W, S))


def test_workgroup_lds_kernel_batches_and_rows(gpu_lib, capsys):
    """A ragged batch through one plan (frames of all clips in one launch) equals the single-clip calls bit for bit; the
    spectrogram / chromagram rows of a big window against the oracle (incl. the chromagram's truncated tail frame)."""
    fs, W, S = 16000, 8000, 4000
    lens = [W, 3 * W + 17, 2 * W - 1, 40000, W + S]
    clips = [synth_clip(6100 + i, n, fs) for i, n in enumerate(lens)]
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, W, S, deltas=True)
    for c, r in zip(clips, res):
        single, _ = ShortTermFeatures.feature_extraction(c, fs, W, S)
        assert np.array_equal(single, r)
        ref, _ = O.feature_extraction(c, fs, W, S)
        assert_parity(r, ref, "ragged big-window batch", sig=(c, fs, W, S))
    x = synth_clip(6200, 7 * fs + 123, fs)
    spec, t_ax, f_ax = ShortTermFeatures.spectrogram(x, fs, W, S)
    capsys.readouterr()
    ref_s, _, _ = O.spectrogram(x, fs, W, S)
    assert_parity(np.ascontiguousarray(spec.T), np.ascontiguousarray(ref_s.T), "big-window spectrogram")
    chroma, _, _ = ShortTermFeatures.chromagram(x, fs, W, S)
    ref_c, _, _ = O.chromagram(x, fs, W, S)
    assert_parity(np.ascontiguousarray(chroma.T), np.ascontiguousarray(ref_c.T), "big-window chromagram")
    # the same through the split transform (48 000-sample windows), stereo input, a ragged batch
    fs, W, S = 48000, 48000, 24000
    xs = synth_clip(6300, 4 * fs + 1000, fs, stereo=True)
    mono = O.stereo_to_mono(xs)
    spec, _, _ = ShortTermFeatures.spectrogram(xs, fs, W, S)
    capsys.readouterr()
    ref_s, _, _ = O.spectrogram(mono, fs, W, S)
    assert_parity(np.ascontiguousarray(spec.T), np.ascontiguousarray(ref_s.T), "split-transform spectrogram")
    chroma, _, _ = ShortTermFeatures.chromagram(xs, fs, W, S)
    ref_c, _, _ = O.chromagram(mono, fs, W, S)
    assert_parity(np.ascontiguousarray(chroma.T), np.ascontiguousarray(ref_c.T), "split-transform chromagram")
    clips = [synth_clip(6400 + i, n, fs) for i, n in enumerate([W, 3 * W + 17, 2 * W - 1])]
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, W, S, deltas=True)
    for c, r in zip(clips, res):
        single, _ = ShortTermFeatures.feature_extraction(c, fs, W, S)
        assert np.array_equal(single, r)


@pytest.mark.parametrize("fs,window,step,seconds", [
    (16000, 8000, 4000, 6.0),       # 0.5 s window: beyond the one-wave kernels -> one workgroup per frame (kernels_wg.hpp)
    (16000, 16000, 16000, 8.0),     # 1 s / 1 s, the music_thumbnailing shape (audioSegmentation.py:1137)
    (16000, 9001, 4500, 3.0),       # odd prime window: Stockham passes through HBM scratch, one O(N^2) pass
    (44100, 44100, 22050, 3.0),     # 1 s at 44.1 kHz: 22 050 complex points do not fit the LDS -> split transform
    (16000, 80000, 40000, 12.0),    # 5 s windows: 40 000 complex points -> radix passes through HBM scratch (kernels_big.hpp)
])
def test_big_windows_match_oracle(gpu_lib, fs, window, step, seconds):
    x = synth_clip(700 + window, int(seconds * fs), fs=fs)
    ref, _ = O.feature_extraction(x, fs, window, step)
    got, _ = ShortTermFeatures.feature_extraction(x, fs, window, step)
    assert_parity(got, ref, "big window %d/%d" % (window, step), sig=(x, fs, window, step))


def test_big_window_spectrogram(gpu_lib, capsys):
    x = synth_clip(801, 5 * 16000)
    S, _, _ = ShortTermFeatures.spectrogram(x, 16000, 8000, 4000)
    capsys.readouterr()
    ref, _, _ = O.spectrogram(x, 16000, 8000, 4000)
    assert_parity(S, ref, "big spectrogram")


def test_concurrent_python_threads(gpu_lib):
    """ctypes releases the GIL: host-buffer entry points are safe to call from several threads, and the calls really
    overlap on the device (each runs in its own lane: stream + scratch) instead of queueing behind one mutex."""
    import threading
    clips = [synth_clip(900 + i, 16000 * 60 * (1 + i % 3)) for i in range(6)]
    expect = [ShortTermFeatures.feature_extraction(c, 16000, 800, 400)[0].copy() for c in clips]
    got = [None] * len(clips)
    gpu_lib.paa_debug_lane_peak()                       # reset the high-water mark

    def work(k):
        for _ in range(8):
            got[k] = ShortTermFeatures.feature_extraction(clips[k], 16000, 800, 400)[0]
    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(clips))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    peak = gpu_lib.paa_debug_lane_peak()
    for e, g_ in zip(expect, got):
        assert np.array_equal(e, g_)
    assert peak >= 2, "six threads never had two calls in flight at once (peak %d)" % peak


def test_result_arrays_are_recycled_only_when_released(gpu_lib):
    """Large results are views of pooled buffers (_ffi.result_array): a buffer comes back only after the caller dropped
    every reference, so results a caller still holds never change under it."""
    x = synth_clip(950, 16000 * 120)
    a, _ = ShortTermFeatures.feature_extraction(x, 16000, 800, 400)
    keep = a[:5].copy()
    view = a[:5]
    addr = a.__array_interface__["data"][0]
    del a
    b, _ = ShortTermFeatures.feature_extraction(synth_clip(951, 16000 * 120), 16000, 800, 400)
    assert b.__array_interface__["data"][0] != addr          # `view` still points into the first buffer
    assert np.array_equal(view, keep)
    del view, b
    c, _ = ShortTermFeatures.feature_extraction(x, 16000, 800, 400)
    assert np.array_equal(c[:5], keep)


def test_many_clip_batch_properties(gpu_lib):
    """BASELINE config 3/4 shapes at reduced count: identical clips give identical slabs wherever they sit in the
    batch (tiles never span clips, per-clip normalisation), and a clip's slab equals its single-clip result."""
    fs = 16000
    a, b = synth_clip(41, 10 * fs), synth_clip(42, 10 * fs)
    clips = [a, b] * 150                                   # 300 x 10 s, like a shard of config 4
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, 800, 400, deltas=False)
    ra, _ = ShortTermFeatures.feature_extraction(a, fs, 800, 400, deltas=False)
    rb, _ = ShortTermFeatures.feature_extraction(b, fs, 800, 400, deltas=False)
    assert ra.shape == (34, 399)
    for k, r in enumerate(res):
        assert np.array_equal(r, ra if k % 2 == 0 else rb)
    c30 = synth_clip(43, 30 * fs)
    mids, _ = MidTermFeatures.mid_feature_extraction_batch([c30] * 40, fs, fs, fs, 800, 400)   # config 3 shape
    ref_mid, _, _ = O.mid_feature_extraction(c30, fs, fs, fs, 800, 400)
    assert mids[0].shape == (136, 30)
    assert_parity(mids[0], ref_mid, "config 3 clip", sig=(c30, fs, 800, 400))
    for m in mids[1:]:
        assert np.array_equal(m, mids[0])


def test_fused_stereo_to_mono(gpu_lib):
    """(n, 2) int16 input: L + R summed on the device == audioBasicIO.stereo_to_mono on the host, bit for bit in
    the samples, so the features equal the float64 mono path and the reference golden."""
    g = load_golden([p for p in golden_files("stereo")][0])
    st = load_golden(os.path.join(os.path.dirname(golden_files("stereo")[0]), "synth5_stereo_1102_441.npz"))
    fused, _ = ShortTermFeatures.feature_extraction(g["stereo"], 44100, 1102, 441)
    assert_parity(fused, st["features"], "fused stereo vs reference", sig=(g["mono"], 44100, 1102, 441))
    mono_path, _ = ShortTermFeatures.feature_extraction(g["mono"], 44100, 1102, 441)
    assert np.allclose(fused, mono_path, rtol=1e-12, atol=1e-13)
    xs = synth_clip(88, 3 * 16000 + 3, stereo=True)                 # odd length: scalar tail of the sum kernel
    ref_mid, ref_st, _ = O.mid_feature_extraction(O.stereo_to_mono(xs), 16000, 16000, 16000, 800, 400)
    mid, st2, _ = MidTermFeatures.mid_feature_extraction(xs, 16000, 16000, 16000, 800, 400)
    sig2 = (O.stereo_to_mono(xs), 16000, 800, 400)
    assert_parity(st2, ref_st, "fused stereo short", sig=sig2)
    assert_parity(mid, ref_mid, "fused stereo mid", sig=sig2)
    # the kernels that see stereo samples less often: Stockham passes in LDS (prime window), passes through HBM scratch
    # (window beyond the LDS envelope), the truncated chromagram tail frame -- L + R formed in their loads as well
    for fs, W, S, n in ((22050, 1103, 441, 22050 * 2 + 5), (16000, 8000, 4000, 16000 * 3 + 1)):
        xs = synth_clip(89 + W, n, fs=fs, stereo=True)
        mono = O.stereo_to_mono(xs)
        got, _ = ShortTermFeatures.feature_extraction(xs, fs, W, S)
        ref, _ = O.feature_extraction(mono, fs, W, S)
        assert_parity(got, ref, "stereo through window %d" % W, sig=(mono, fs, W, S))
    xs = synth_clip(97, 44100 + 900, fs=44100, stereo=True)         # the last frame is truncated (the reference FFTs what is left)
    chroma, _, _ = ShortTermFeatures.chromagram(xs, 44100, 1102, 441)
    chroma_mono, _, _ = ShortTermFeatures.chromagram(O.stereo_to_mono(xs), 44100, 1102, 441)
    assert chroma.shape == chroma_mono.shape and np.allclose(chroma, chroma_mono, rtol=1e-9, atol=1e-12)


def test_c_client_on_gpu(gpu_lib, tmp_path):
    """The plain C client of examples/c_api_demo.c runs the hot path through the C ABI WITHOUT Python -- on the clip of a
    golden file of the unmodified reference (synth11_800_400: ShortTermFeatures.feature_extraction at 800 / 400, :543), and
    the whole 68 x T matrix it writes, the entries it prints and its checksum are held against that golden."""
    import shutil
    import subprocess
    from test_abi_cpu import ROOT, test_c_client_links_and_fails_loudly_without_gpu as run_c_client
    run_c_client(tmp_path)                                    # builds the client, runs its built-in tone
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    g = load_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth11_800_400.npz"))
    assert int(g["fs"]) == 16000 and int(g["window"]) == 800 and int(g["step"]) == 400 and g["signal"].dtype == np.int16
    raw, out = tmp_path / "clip.raw", tmp_path / "features.f64"
    g["signal"].tofile(raw)
    run = subprocess.run([str(tmp_path / "c_api_demo"), str(raw), str(out)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    ref = g["features"]
    T = ref.shape[1]
    assert "frames %d:" % T in run.stdout
    F = np.fromfile(out, dtype=np.float64).reshape(68, T)
    assert_parity(F, ref, "C client, golden synth11_800_400", sig=(g["signal"], 16000, 800, 400))
    seen = 0
    for line in run.stdout.splitlines():
        tok = line.split()
        if tok[:1] == ["entry"]:
            r, c, v = int(tok[1]), int(tok[2]), float(tok[3])
            assert v == F[r, c]                               # 17 digits: the printed entries ARE the matrix entries
            seen += 1
        elif tok[:1] == ["sum_abs"]:
            assert abs(float(tok[1]) - np.abs(ref).sum()) <= 1e-6 * np.abs(ref).sum()
            seen += 100
    assert seen == 115, run.stdout


@pytest.mark.parametrize("seed", range(12))
def test_random_window_step_sweep(gpu_lib, seed):
    """Seeded sweep over (fs, window, step): odd / prime / composite windows, overlapping and gapped steps."""
    rng = np.random.default_rng(1000 + seed)
    fs = int(rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000]))
    window = int(rng.integers(max(200, fs // 60), fs // 12))
    if window / 2 < 12 * np.log2((fs / 2) / 27.5) + 2:       # chroma needs max slot < num_fft (reference :286)
        window = int(fs // 20)
    step = int(rng.integers(window // 4, window + window // 4))
    x = synth_clip(2000 + seed, int(fs * 0.6) + window, fs=fs)
    try:
        ref, _ = O.feature_extraction(x, fs, window, step)
    except (ValueError, IndexError) as exc:                  # the reference rejects the configuration ...
        with pytest.raises(type(exc)):                       # ... and so must the drop-in, with the same type
            ShortTermFeatures.feature_extraction(x, fs, window, step)
        return
    got, _ = ShortTermFeatures.feature_extraction(x, fs, window, step)
    assert_parity(got, ref, "fs=%d W=%d S=%d" % (fs, window, step), sig=(x, fs, window, step))


def test_experiment_switches_change_nothing_in_the_default_build(gpu_lib, monkeypatch):
    """PAA_KERNEL_DEBUG (bits 4 / 8 used to drop output stores), PAA_RUN_CAP, PAA_NO_MIX, PAA_F800_WAVES ... are read only by
    -DPAA_EXPERIMENTS builds: with all of them set, plans pick the same kernels and every output bit is the same."""
    from synth import synth_clip
    cases = [(16000, 800, 400), (16000, 640, 640), (16000, 1024, 512), (44100, 1102, 441)]
    clips = {c: synth_clip(77 + i, 3 * c[0], c[0]) for i, c in enumerate(cases)}

    def run():
        out = {}
        for (fs, w, s) in cases:
            plan = _ffi.Plan(np.array([0, 3 * fs], dtype=np.int64), fs, w, s, deltas=True, sample_kind=0)
            name = plan.kernel_name
            plan.destroy()
            F, _ = ShortTermFeatures.feature_extraction(clips[(fs, w, s)], fs, w, s)
            out[(fs, w, s)] = (name, F)
        return out
    before = run()
    for k, v in {"PAA_KERNEL_DEBUG": "15", "PAA_RUN_CAP": "16", "PAA_NO_MIX": "1", "PAA_F800_WAVES": "4", "PAA_F800_PACE": "0",
                 "PAA_MIX_NO_LEAN": "1", "PAA_MIX_TW_GLOBAL": "1", "PAA_MIX_NO_SKEW": "1"}.items():
        monkeypatch.setenv(k, v)
    after = run()
    for c in cases:
        assert before[c][0] == after[c][0], (c, before[c][0], after[c][0])
        assert np.array_equal(before[c][1], after[c][1]), c

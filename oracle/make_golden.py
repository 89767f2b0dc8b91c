"""Generate tests/golden/*.npz by running the UNMODIFIED reference (build container only).

Usage:  python oracle/make_golden.py            (needs /root/reference)
Every file holds the exact input signal and the reference's outputs, so the
-m gpu tests never need /root/reference.  Re-running must reproduce the
committed files bit for bit (numpy 2.2.6 / scipy 1.15.3).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import load_reference  # noqa: E402
from synth import synth_clip  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = load_reference.REFERENCE_ROOT


def wav(path, seconds=None):
    import warnings
    import scipy.io.wavfile as wavfile
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fs, x = wavfile.read(os.path.join(REF, path))
    if seconds is not None:
        x = x[:int(seconds * fs)]
    return fs, np.ascontiguousarray(x)


def main():
    ref_st, ref_mt, ref_io = load_reference.load()
    os.makedirs(OUT, exist_ok=True)
    written = []

    def save(name, **arrays):
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        written.append(name)

    def st_case(name, sig, fs, win, step, deltas=True):
        F, names = ref_st.feature_extraction(sig, fs, win, step, deltas)
        save(name, kind="st", signal=sig, fs=fs, window=win, step=step, deltas=deltas,
             features=F, names=np.array(names))

    def mid_case(name, sig, fs, mw, ms, sw, ss):
        mid, st, names = ref_mt.mid_feature_extraction(sig, fs, mw, ms, sw, ss)
        save(name, kind="mid", signal=sig, fs=fs, mid_window=mw, mid_step=ms, window=sw, step=ss,
             mid=mid, features=st, names=np.array(names))

    def spec_case(name, sig, fs, win, step):
        with contextlib.redirect_stdout(io.StringIO()):
            S, t_ax, f_ax = ref_st.spectrogram(sig, fs, win, step)
        C, ct_ax, cf_ax = ref_st.chromagram(sig, fs, win, step)
        save(name, kind="spec", signal=sig, fs=fs, window=win, step=step, specgram=S,
             spec_time=np.array(t_ax), spec_freq=np.array(f_ax), chromagram=C,
             chroma_time=np.array(ct_ax), chroma_names=np.array(cf_ax))

    # ---- A. in-tree WAVs --------------------------------------------------
    fs, x = wav("pyAudioAnalysis/data/doremi.wav")
    st_case("doremi_800_400", x, fs, 800, 400)                       # BASELINE config 1
    fs, x = wav("pyAudioAnalysis/data/doremi.wav", 3.0)
    st_case("doremi3s_800_800", x, fs, 800, 800)
    st_case("doremi3s_640_640", x, fs, 640, 640)
    st_case("doremi3s_320_160", x, fs, 320, 160)
    st_case("doremi3s_800_400_nodelta", x, fs, 800, 400, deltas=False)
    spec_case("doremi3s_spec_640_640", x, fs, 640, 640)              # tests/cmd_test_00/01.sh shape
    fs, x = wav("pyAudioAnalysis/data/3WORDS.wav", 2.0)
    st_case("3words2s_1102_441", x, fs, 0.025 * fs, 0.010 * fs)     # float args -> int() truncation
    spec_case("3words2s_spec_1102_441", x, fs, 0.025 * fs, 0.010 * fs)
    for stem in ("count", "diarizationExample", "speech_music_sample", "recording1"):
        fs, x = wav("pyAudioAnalysis/data/%s.wav" % stem, 2.0)
        st_case("%s2s_800_400" % stem, x, fs, 800, 400)
    fs, x = wav("pytests/test_data/1_sec_wav.wav")
    st_case("pytest_1sec", x, fs, 0.050 * fs, 0.050 * fs)            # pytests/test_feature_extraction.py:10
    fs, x = wav("pytests/test_data/5_sec_wav.wav")
    mid_case("pytest_5sec_mid", x, fs, 1 * fs, 1 * fs, 0.05 * fs, 0.05 * fs)   # :19

    # ---- B. seeded synthetic ---------------------------------------------
    x = synth_clip(11, 3 * 16000)
    st_case("synth11_800_400", x, 16000, 800, 400)
    mid_case("synth11_mid_1s_1s", x, 16000, 16000, 16000, 800, 400)
    mid_case("synth11_mid_float", x, 16000, 1.0 * 16000, 0.1 * 16000, 800.0, 800.0)
    xs = synth_clip(5, 44100, fs=44100, stereo=True)
    mono = ref_io.stereo_to_mono(xs)                                  # float64 with .5 fractions
    assert mono.dtype == np.float64
    st_case("synth5_stereo_1102_441", mono, 44100, 1102, 441)
    save("synth5_stereo_raw", kind="stereo", stereo=xs, mono=mono)
    spec_case("synth5_spec_1102_441", mono, 44100, 1102, 441)

    # ---- C. degenerate ------------------------------------------------------
    st_case("zeros_2000", np.zeros(2000, dtype=np.int16), 16000, 800, 400)
    x = synth_clip(12, 4000)
    st_case("exact_one_window", x[:800].copy(), 16000, 800, 400)
    st_case("w_plus_s_minus_1", x[:1199].copy(), 16000, 800, 400)
    st_case("constant_dc", np.full(3000, 1234, dtype=np.int16), 16000, 800, 400)
    sq = np.where((np.arange(4000) // 7) % 2 == 0, 32767, -32768).astype(np.int16)
    st_case("square_fullscale", sq, 16000, 800, 400)
    gap = synth_clip(13, 8000).copy()
    gap[2000:6000] = 0                                                 # frames of exact digital silence
    st_case("silent_gap", gap, 16000, 800, 400)
    # fs/4 tone (5,0,-5,0,...): exact zero SAMPLES (sign()=0 half crossings) plus a zeroed span.
    # (A pure Nyquist tone +-5 would put all energy in the dropped bin: the reference's spectrum is
    #  then exactly 0 and every spectral feature is a function of FFT round-off -- not a parity case.)
    alt = np.tile(np.array([5, 0, -5, 0], dtype=np.int16), 1000)
    alt[1000:1400] = 0
    st_case("quarter_tone_with_zeros", alt, 16000, 800, 400)

    print("wrote %d golden files to %s" % (len(written), OUT))
    tot = sum(os.path.getsize(os.path.join(OUT, n + ".npz")) for n in written)
    print("total %.2f MB" % (tot / 1e6))


if __name__ == "__main__" and not any(a in sys.argv for a in ("--dir", "--thumb", "--wide", "--silence", "--round3", "--round4", "--round5", "--round6")):
    main()


def directory_goldens():
    """Row f1/f2 of SURVEY 8f: directory walkers + beat extraction, run by the unmodified reference on a small
    directory of synthetic WAV files (written here with scipy.io.wavfile; the arrays are stored in the golden)."""
    import tempfile
    import scipy.io.wavfile as wavfile
    ref_st, ref_mt, ref_io = load_reference.load()
    files = {
        "a_mono16k_3s.wav": (16000, synth_clip(21, 48000)),
        "b_mono16k_1s2.wav": (16000, synth_clip(22, 19200)),
        "c_stereo16k_2s.wav": (16000, synth_clip(23, 32000, stereo=True)),
        "d_mono8k_2s.wav": (8000, synth_clip(24, 16000, fs=8000)),
        "e_tiny.wav": (16000, synth_clip(25, 1600)),            # < fs/5 samples: skipped (:181-183)
    }
    out = {"kind": "dir", "file_names": np.array(sorted(files)), "mid_window": 1.0, "mid_step": 1.0,
           "short_window": 0.05, "short_step": 0.05}
    with tempfile.TemporaryDirectory() as d:
        for name, (fs, x) in files.items():
            wavfile.write(os.path.join(d, name), fs, x)
            out["wav_fs_" + name] = fs
            out["wav_x_" + name] = x
        open(os.path.join(d, "f_empty.wav"), "wb").close()      # zero bytes: skipped (:168-170)
        with contextlib.redirect_stdout(io.StringIO()):
            f_beat, list_beat, names_beat = ref_mt.directory_feature_extraction(d, 1.0, 1.0, 0.05, 0.05, compute_beat=True)
            f_nobeat, list_nb, names_nb = ref_mt.directory_feature_extraction(d, 1.0, 1.0, 0.05, 0.05, compute_beat=False)
        # directory_feature_extraction_no_avg (:263-309) has no size check: the reference dies on the zero-byte file
        # (scipy.io.wavfile: ValueError "File format b'' not understood") -- recorded, then run without that file
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref_mt.directory_feature_extraction_no_avg(d, 1.0, 1.0, 0.05, 0.05)
            out["noavg_empty_file_error"] = np.array("")
        except Exception as exc:
            out["noavg_empty_file_error"] = np.array(type(exc).__name__)
        os.remove(os.path.join(d, "f_empty.wav"))
        with contextlib.redirect_stdout(io.StringIO()):
            X, Y, flist = ref_mt.directory_feature_extraction_no_avg(d, 1.0, 1.0, 0.05, 0.05)
        out.update(noavg_features=X, noavg_index=Y, noavg_files=np.array([os.path.basename(p) for p in flist]))
        out.update(features_beat=f_beat, files_beat=np.array([os.path.basename(p) for p in list_beat]),
                   names_beat=np.array(names_beat), features_nobeat=f_nobeat,
                   files_nobeat=np.array([os.path.basename(p) for p in list_nb]), names_nobeat=np.array(names_nb))
        # beat extraction alone on a longer clip (0.05 and 0.025 s steps)
        x = synth_clip(26, 10 * 16000)
        st, _ = ref_st.feature_extraction(x, 16000, 800, 800)
        bpm, ratio = ref_mt.beat_extraction(st, 0.05)
        st2, _ = ref_st.feature_extraction(x, 16000, 800, 400)
        bpm2, ratio2 = ref_mt.beat_extraction(st2, 0.025)
        out.update(beat_signal=x, beat_050=np.array([bpm, ratio]), beat_025=np.array([bpm2, ratio2]))
    np.savez_compressed(os.path.join(OUT, "directory_small.npz"), **out)
    print("wrote directory_small.npz: %s files -> %s" % (len(files) + 1, f_beat.shape))


if __name__ == "__main__" and "--dir" in sys.argv:
    directory_goldens()


def thumbnail_goldens():
    """Row f4 of SURVEY 8f: audioSegmentation.self_similarity_matrix / music_thumbnailing run by the unmodified
    reference.  Song inputs are regenerated from their seed (synth.synth_song); a checksum guards the bytes."""
    from synth import synth_song
    ref_st, _, _ = load_reference.load()
    ref_seg = load_reference.load_segmentation()

    def thumb_case(name, seed, seconds, fs, sw, ss, thumb, l1, l2):
        x = synth_song(seed, seconds, fs)
        st, _ = ref_st.feature_extraction(x, fs, fs * sw, fs * ss)
        sim = ref_seg.self_similarity_matrix(st)
        a1, a2, b1, b2, filt = ref_seg.music_thumbnailing(x, fs, sw, ss, thumb, l1, l2)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="thumb", seed=seed, seconds=seconds, fs=fs,
                            short_window=sw, short_step=ss, thumb_size=thumb, limit_1=l1, limit_2=l2,
                            checksum=np.int64(np.sum(x.astype(np.int64) * (np.arange(len(x)) % 251 + 1))),
                            features=st, sim=sim, filtered=filt, pos=np.array([a1, a2, b1, b2]))
        print(name, st.shape, filt.shape, (a1, a2, b1, b2))

    thumb_case("thumb_song40s", 31, 40.0, 16000, 1.0, 0.5, 5.0, 0, 1)
    thumb_case("thumb_song40s_limits", 31, 40.0, 16000, 1.0, 0.5, 5.0, 0.1, 0.9)
    thumb_case("thumb_song30s_half", 32, 30.0, 16000, 0.5, 0.25, 4.0, 0, 1)
    thumb_case("thumb_song24s_8k", 33, 24.0, 8000, 1.0, 0.5, 4.0, 0, 1)

    def sim_case(name, src):
        with np.load(os.path.join(OUT, src + ".npz")) as z:
            F = z["features"]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="sim", source=src, features=F,
                            sim=ref_seg.self_similarity_matrix(F))
        print(name, F.shape)

    sim_case("sim_count2s", "count2s_800_400")
    sim_case("sim_doremi3s_nodelta", "doremi3s_800_400_nodelta")
    sim_case("sim_constant_dc", "constant_dc")           # every row constant -> zero vectors -> NaN off the diagonal
    sim_case("sim_zeros_2000", "zeros_2000")


if __name__ == "__main__" and "--thumb" in sys.argv:
    thumbnail_goldens()


def wide_goldens():
    """Round 2: the rest of SURVEY 8c's list -- the in-tree WAVs round 1 left out (count2, diarizationExample2,
    recording2/3), full-length files where the .npz stays small, and the (800,800) / (640,640) / (320,160) shapes on
    two more files.  Separate entry point so that the round-1 files are not rewritten."""
    ref_st, ref_mt, ref_io = load_reference.load()

    def st_case(name, sig, fs, win, step, deltas=True):
        F, names = ref_st.feature_extraction(sig, fs, win, step, deltas)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="st", signal=sig, fs=fs, window=win, step=step,
                            deltas=deltas, features=F, names=np.array(names))
        print(name, F.shape)

    def spec_case(name, sig, fs, win, step):
        with contextlib.redirect_stdout(io.StringIO()):
            S, t_ax, f_ax = ref_st.spectrogram(sig, fs, win, step)
        C, ct_ax, cf_ax = ref_st.chromagram(sig, fs, win, step)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="spec", signal=sig, fs=fs, window=win, step=step,
                            specgram=S, spec_time=np.array(t_ax), spec_freq=np.array(f_ax), chromagram=C,
                            chroma_time=np.array(ct_ax), chroma_names=np.array(cf_ax))
        print(name, S.shape, C.shape)

    for stem in ("count2", "diarizationExample2", "recording2", "recording3"):
        fs, x = wav("pyAudioAnalysis/data/%s.wav" % stem, 2.0)
        st_case("%s2s_800_400" % stem, x, fs, 800, 400)
    # full-length files (5.9 s, 7 s, 10.5 s, 22.3 s)
    fs, x = wav("pyAudioAnalysis/data/count.wav")
    st_case("count_full_800_400", x, fs, 800, 400)
    st_case("count_full_800_800", x, fs, 800, 800)
    st_case("count_full_640_640", x, fs, 640, 640)
    st_case("count_full_320_160", x, fs, 320, 160, deltas=False)
    fs, x = wav("pyAudioAnalysis/data/speech_music_sample.wav")
    st_case("speech_music_full_800_800", x, fs, 800, 800)
    st_case("speech_music_full_640_640", x, fs, 640, 640)
    st_case("speech_music_full_320_160", x, fs, 320, 160, deltas=False)
    fs, x = wav("pyAudioAnalysis/data/count2.wav")
    st_case("count2_full_800_400", x, fs, 800, 400)
    fs, x = wav("pyAudioAnalysis/data/diarizationExample2.wav")
    st_case("diarizationExample2_full_800_400_nodelta", x, fs, 800, 400, deltas=False)
    # 44.1 kHz speech at the cfg5 shape, longer than the 2 s of round 1; spectrogram + chromagram with a ragged tail
    fs, x = wav("pyAudioAnalysis/data/3WORDS.wav", 5.0)
    st_case("3words5s_1102_441_nodelta", x, fs, 1102, 441, deltas=False)
    fs, x = wav("pyAudioAnalysis/data/doremi.wav")
    spec_case("doremi_full_spec_640_640", x, fs, 640, 640)           # tests/cmd_test_00/01.sh on the whole file


if __name__ == "__main__" and "--wide" in sys.argv:
    wide_goldens()


def silence_goldens():
    """Row f4 (remainder) of SURVEY 8f: audioSegmentation.silence_removal run by the unmodified reference.  Its SVM is
    trained inside the call with scikit-learn (probability=True: libsvm's internal cross-validation draws from NumPy's
    global random state), so the global seed is fixed before every call and stored with the result."""
    ref_st, _, _ = load_reference.load()
    ref_seg = load_reference.load_segmentation()

    def case(name, sig, fs, st_win, st_step, smooth_window, weight, seed=1234):
        np.random.seed(seed)
        segs = ref_seg.silence_removal(sig, fs, st_win, st_step, smooth_window, weight)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="silence", signal=sig, fs=fs, st_win=st_win,
                            st_step=st_step, smooth_window=smooth_window, weight=weight, seed=seed,
                            segments=np.array(segs, dtype=np.float64).reshape(-1, 2))
        print(name, len(segs), segs[:4])

    fs, x = wav("pyAudioAnalysis/data/count.wav")
    case("silence_count_020_020", x, fs, 0.020, 0.020, 1.0, 0.3)                # the CLI's defaults (audioAnalysis.py)
    case("silence_count_050_050", x, fs, 0.050, 0.050, 0.5, 0.5)
    fs, x = wav("pyAudioAnalysis/data/recording1.wav", 20.0)
    case("silence_recording1_20s", x, fs, 0.050, 0.050, 0.5, 0.5)
    g = synth_clip(71, 8 * 16000).copy()
    for a, b in ((0.0, 1.0), (2.5, 3.5), (5.0, 5.6), (7.2, 8.0)):               # digital silence between bursts
        g[int(a * 16000):int(b * 16000)] = 0
    case("silence_synth_gaps", g, 16000, 0.050, 0.025, 0.5, 0.5)


if __name__ == "__main__" and "--silence" in sys.argv:
    silence_goldens()


def round3_goldens():
    """Round 3: outputs of the unmodified reference at the shapes the round-3 kernels own -- 50 ms / 40 ms windows at 44.1 and
    48 kHz (2205, 2400, 1764, 1920: mixed-radix kernel), 50 ms at 8 and 32 kHz (400: 2 RA RB family, 1600), 1024, and the
    float64 mono signal audioBasicIO.stereo_to_mono hands over for a stereo file at 800 / 400.  Short clips keep the files
    small; the seeded synthetic clips are oracle/synth.py's."""
    from synth import synth_clip
    ref_st, ref_mt, ref_io = load_reference.load()

    def st_case(name, sig, fs, win, step, deltas=True):
        F, names = ref_st.feature_extraction(sig, fs, win, step, deltas)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="st", signal=sig, fs=fs, window=win, step=step,
                            deltas=deltas, features=F, names=np.array(names))
        print(name, F.shape)

    def spec_case(name, sig, fs, win, step):
        with contextlib.redirect_stdout(io.StringIO()):
            S, t_ax, f_ax = ref_st.spectrogram(sig, fs, win, step)
        C, ct_ax, cf_ax = ref_st.chromagram(sig, fs, win, step)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="spec", signal=sig, fs=fs, window=win, step=step,
                            specgram=S, spec_time=np.array(t_ax), spec_freq=np.array(f_ax), chromagram=C,
                            chroma_time=np.array(ct_ax), chroma_names=np.array(cf_ax))
        print(name, S.shape, C.shape)

    fs, x = wav("pyAudioAnalysis/data/3WORDS.wav", 2.0)                     # 44.1 kHz speech
    st_case("3words2s_2205_1102", x, fs, 2205, 1102)                          # 50 ms / 25 ms: odd window
    spec_case("3words2s_spec_1764_1764", x, fs, 1764, 1764)                   # the CLI's 40 ms / 40 ms
    x48 = synth_clip(4800, 2 * 48000, 48000)
    st_case("synth48k_2400_1200", x48, 48000, 2400, 1200)
    spec_case("synth48k_spec_1920_960", x48, 48000, 1920, 960)
    x8 = synth_clip(800, 3 * 8000, 8000)
    st_case("synth8k_400_200", x8, 8000, 400, 200)
    x32 = synth_clip(3200, 2 * 32000, 32000)
    st_case("synth32k_1600_800_nodelta", x32, 32000, 1600, 800, deltas=False)
    x16 = synth_clip(1600, 3 * 16000, 16000)
    st_case("synth16k_1024_512", x16, 16000, 1024, 512)
    xs = synth_clip(1601, 3 * 16000, 16000, stereo=True)
    mono = ref_io.stereo_to_mono(xs)                                          # float64, .5 fractions (audioBasicIO.py:167)
    st_case("synth16k_stereo_to_mono_f64_800_400", mono, 16000, 800, 400)
    st_case("synth16k_stereo_to_mono_f64_640_640", mono, 16000, 640, 640, deltas=False)


if __name__ == "__main__" and "--round3" in sys.argv:
    round3_goldens()


def round4_goldens():
    """Round 4: outputs of the unmodified reference at the shapes the three-pass register-FFT kernels (csrc/kernels_tri.hpp)
    took over -- the 50 ms window at 11.025 kHz and the 25 ms window at 22.05 kHz (551 samples, odd: 19 x 29; window and step
    are passed as the FLOATS the reference's callers compute, 0.050 * fs = 551.25 -> int() = 551), 40 ms at 44.1 kHz as
    features, 50 ms at 24 kHz."""
    from synth import synth_clip
    ref_st, ref_mt, ref_io = load_reference.load()

    def st_case(name, sig, fs, win, step, deltas=True):
        F, names = ref_st.feature_extraction(sig, fs, win, step, deltas)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="st", signal=sig, fs=fs, window=win, step=step,
                            deltas=deltas, features=F, names=np.array(names))
        print(name, F.shape)

    def spec_case(name, sig, fs, win, step):
        with contextlib.redirect_stdout(io.StringIO()):
            S, t_ax, f_ax = ref_st.spectrogram(sig, fs, win, step)
        C, ct_ax, cf_ax = ref_st.chromagram(sig, fs, win, step)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="spec", signal=sig, fs=fs, window=win, step=step,
                            specgram=S, spec_time=np.array(t_ax), spec_freq=np.array(f_ax), chromagram=C,
                            chroma_time=np.array(ct_ax), chroma_names=np.array(cf_ax))
        print(name, S.shape, C.shape)

    x11 = synth_clip(1102, 3 * 11025, 11025)
    st_case("synth11k_551_275", x11, 11025, 0.050 * 11025, 0.025 * 11025)     # audioTrainTest.py:28-29 at 11.025 kHz
    spec_case("synth11k_spec_551_275", x11, 11025, 551, 275)
    x22 = synth_clip(2205, 2 * 22050, 22050)
    st_case("synth22k_551_220", x22, 22050, 0.025 * 22050, 0.010 * 22050)
    fs, x = wav("pyAudioAnalysis/data/3WORDS.wav", 2.0)                     # 44.1 kHz speech
    st_case("3words2s_1764_882", x, fs, 1764, 882)
    x24 = synth_clip(2400, 2 * 24000, 24000)
    st_case("synth24k_1200_600_nodelta", x24, 24000, 1200, 600, deltas=False)


if __name__ == "__main__" and "--round4" in sys.argv:
    round4_goldens()


def round5_goldens():
    """Round 5: outputs of the unmodified reference at the power-of-two windows the three-pass register-FFT kernels took over
    (csrc/kernels_tri.hpp: 1024 = 2 x 8 x 8 x 8 had a golden already, synth16k_1024_512) -- 512 / 256 (entropy blocks of 51
    samples: odd, a sample pair straddles block boundaries; the reference's mid-term path too) and 2048 / 1024 on the 44.1 kHz
    speech of the reference's data folder and on a seeded clip, plus their spectrogram / chromagram -- and at the big windows
    music_thumbnailing uses by default (audioSegmentation.py:1137: 1.0 s / 0.5 s) on a seeded 16 kHz clip and at 44.1 kHz."""
    from synth import synth_clip
    ref_st, ref_mt, ref_io = load_reference.load()

    def st_case(name, sig, fs, win, step, deltas=True):
        F, names = ref_st.feature_extraction(sig, fs, win, step, deltas)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="st", signal=sig, fs=fs, window=win, step=step,
                            deltas=deltas, features=F, names=np.array(names))
        print(name, F.shape)

    def mid_case(name, sig, fs, mw, ms, sw, ss):
        mid, st, names = ref_mt.mid_feature_extraction(sig, fs, mw, ms, sw, ss)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="mid", signal=sig, fs=fs, mid_window=mw, mid_step=ms,
                            window=sw, step=ss, mid=mid, features=st, names=np.array(names))
        print(name, mid.shape, st.shape)

    def spec_case(name, sig, fs, win, step):
        with contextlib.redirect_stdout(io.StringIO()):
            S, t_ax, f_ax = ref_st.spectrogram(sig, fs, win, step)
        C, ct_ax, cf_ax = ref_st.chromagram(sig, fs, win, step)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="spec", signal=sig, fs=fs, window=win, step=step,
                            specgram=S, spec_time=np.array(t_ax), spec_freq=np.array(f_ax), chromagram=C,
                            chroma_time=np.array(ct_ax), chroma_names=np.array(cf_ax))
        print(name, S.shape, C.shape)

    x16 = synth_clip(512, 2 * 16000, 16000)
    st_case("synth16k_512_256", x16, 16000, 512, 256)
    mid_case("synth16k_mid_512_256", x16, 16000, 16000, 8000, 512, 256)
    spec_case("synth16k_spec_512_256", x16, 16000, 512, 256)
    fs, x = wav("pyAudioAnalysis/data/3WORDS.wav", 2.0)                     # 44.1 kHz speech
    st_case("3words2s_2048_1024", x, fs, 2048, 1024)
    x44 = synth_clip(2048, 2 * 44100, 44100)
    st_case("synth44k_2048_1024_nodelta", x44, 44100, 2048, 1024, deltas=False)
    spec_case("synth44k_spec_2048_1024", x44, 44100, 2048, 1024)
    fs, x = wav("pyAudioAnalysis/data/doremi.wav", 2.5)                      # 16 kHz
    st_case("doremi_1024_256", x, fs, 1024, 256)
    # music_thumbnailing's default short-term window (audioSegmentation.py:1137): 1 s / 0.5 s
    x1s = synth_clip(16000, 12 * 16000, 16000)
    st_case("synth16k_16000_8000", x1s, 16000, 16000, 8000)
    x1s44 = synth_clip(44100, 6 * 44100, 44100)
    st_case("synth44k_44100_22050_nodelta", x1s44, 44100, 44100, 22050, deltas=False)


if __name__ == "__main__" and "--round5" in sys.argv:
    round5_goldens()


def round6_goldens():
    """Round 6: outputs of the unmodified reference at windows whose FFT length has a prime factor above 13 -- the shapes the
    Bluestein kernel took over (csrc/kernels_blu.hpp; the reference takes any int(window), ShortTermFeatures.py:563-564): the
    prime 1103 and 0.030 x 22050 = 661 (prime) on seeded 22.05 kHz clips (with their digitally silent span), 46 ms at 16 kHz =
    736 = 2^5 x 23 on the reference's own doremi.wav, 202 = 2 x 101 (convolution length 512), the prime 2203 at 44.1 kHz
    (length 4096), their spectrogram / chromagram, and a spectrogram at 158 = 2 x 79 (length 256: too small for the
    reference's mel bank / chroma, its spectrogram works)."""
    from synth import synth_clip
    ref_st, ref_mt, ref_io = load_reference.load()

    def st_case(name, sig, fs, win, step, deltas=True):
        F, names = ref_st.feature_extraction(sig, fs, win, step, deltas)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="st", signal=sig, fs=fs, window=win, step=step,
                            deltas=deltas, features=F, names=np.array(names))
        print(name, F.shape)

    def mid_case(name, sig, fs, mw, ms, sw, ss):
        mid, st, names = ref_mt.mid_feature_extraction(sig, fs, mw, ms, sw, ss)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="mid", signal=sig, fs=fs, mid_window=mw, mid_step=ms,
                            window=sw, step=ss, mid=mid, features=st, names=np.array(names))
        print(name, mid.shape, st.shape)

    def spec_case(name, sig, fs, win, step, chroma=True):
        with contextlib.redirect_stdout(io.StringIO()):
            S, t_ax, f_ax = ref_st.spectrogram(sig, fs, win, step)
        extra = {}
        if chroma:
            C, ct_ax, cf_ax = ref_st.chromagram(sig, fs, win, step)
            extra = dict(chromagram=C, chroma_time=np.array(ct_ax), chroma_names=np.array(cf_ax))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), kind="spec" if chroma else "spec_only", signal=sig, fs=fs, window=win,
                            step=step, specgram=S, spec_time=np.array(t_ax), spec_freq=np.array(f_ax), **extra)
        print(name, S.shape)

    x22 = synth_clip(1103, 3 * 22050, 22050)
    st_case("synth22k_1103_441", x22, 22050, 1103, 441)
    spec_case("synth22k_spec_1103_441", x22[:22050], 22050, 1103, 441)
    x22b = synth_clip(661, 2 * 22050, 22050)
    st_case("synth22k_661_220", x22b, 22050, 661, 220)
    mid_case("synth22k_mid_661_330", x22b, 22050, 22050, 11025, 661, 330)
    spec_case("synth22k_spec_661_220", x22b[:21750], 22050, 661, 220)     # (a truncated chromagram tail frame of 409 >= 330 samples)
    fs, x = wav("pyAudioAnalysis/data/doremi.wav", 2.5)                      # 16 kHz
    st_case("doremi_736_368", x, fs, 736, 368)
    x16 = synth_clip(202, 16000, 16000)
    st_case("synth16k_202_101_nodelta", x16, 16000, 202, 101, deltas=False)
    x44 = synth_clip(2203, 2 * 44100, 44100)
    st_case("synth44k_2203_1100_nodelta", x44, 44100, 2203, 1100, deltas=False)
    xs = synth_clip(5, 44100, 44100, stereo=True)
    st_case("synth44k_stereo_to_mono_f64_1103_441", (xs[:, 1] / 2) + (xs[:, 0] / 2), 44100, 1103, 441)
    x8 = synth_clip(158, 8000, 8000)
    spec_case("synth8k_spec_only_158_79", x8, 8000, 158, 79, chroma=False)
    # 16 ms at 16 kHz: 256 = 2 x 4 x 4 x 8 moved to the three-pass register FFT (entropy blocks of 25 samples + a tail of 6)
    x256 = synth_clip(256, 2 * 16000, 16000)
    st_case("synth16k_256_128", x256, 16000, 256, 128)
    spec_case("synth16k_spec_256_128", x256[:16000], 16000, 256, 128)


if __name__ == "__main__" and "--round6" in sys.argv:
    round6_goldens()

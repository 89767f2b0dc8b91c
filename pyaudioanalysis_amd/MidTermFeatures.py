"""Drop-in for pyAudioAnalysis.MidTermFeatures (reference: pyAudioAnalysis/MidTermFeatures.py).

mid_feature_extraction (:87-127) and its batched many-clip form run on the GPU (HIP only, no CPU path).
Around it, the callers of SURVEY 8f-1/2: beat_extraction (:18-84, a sequential peak detector over 18 short-term rows:
one wave of the GPU's beat_kernel, for a single matrix as for the batched walkers) and the directory walkers
(:140-309), which read the files on the host and push every int16 mono file of one sampling rate through ONE batched
GPU call.
"""
import concurrent.futures
import glob
import os
import time

import numpy as np

from . import ShortTermFeatures, _ffi, audioBasicIO

eps = 0.00000001                          # MidTermFeatures.py:13


def _ratios(mid_window, mid_step, short_window, short_step):
    """Python round() (banker's) exactly where the reference applies it (:100-102).

    feature_extraction int()-truncates short_window/short_step internally (:563-564) but the ratios
    use the caller's (possibly float) values, so they do here too.
    """
    mid_window_ratio = round((mid_window - (short_window - short_step)) / short_step)
    mt_step_ratio = int(round(mid_step / short_step))
    return int(mid_window_ratio), mt_step_ratio


def _mid_names(short_names):
    return [s + "_mean" for s in short_names] + [s + "_std" for s in short_names]   # :113-114


def mid_feature_extraction(signal, sampling_rate, mid_window, mid_step, short_window, short_step):
    """Mid-term feature extraction (reference :87-127).

    RETURNS (mid_features [136 x M], short_features [68 x T], mid_feature_names)
    """
    ratio, step_ratio = _ratios(mid_window, mid_step, short_window, short_step)
    if step_ratio < 1:
        raise ValueError("mid_step / short_step rounds to 0: the reference never terminates "
                         "(MidTermFeatures.py:102,124)")
    window, step = int(short_window), int(short_step)
    kind, sig = _ffi.classify_signal(signal)
    short_names = ShortTermFeatures._feature_names(True)            # deltas always on (:93-95)
    lib = _ffi.lib()
    T = int(lib.paa_num_frames(sig.shape[0], window, step)) if window >= 1 and step >= 1 else 0
    if T < 1:
        raise ValueError("need at least one array to concatenate")  # ShortTermFeatures.py:684
    M = int(lib.paa_num_mid_windows(T, step_ratio))
    st = _ffi.result_array((len(short_names), T))
    mid = np.empty((2 * len(short_names), M), dtype=np.float64)
    if kind == 0:
        rc = lib.paa_mid_features_i16(_ffi.as_i16p(sig), sig.shape[0], float(sampling_rate), window, step,
                                      ratio, step_ratio, _ffi.as_f64p(mid), _ffi.as_f64p(st))
    elif kind == 2:
        rc = lib.paa_mid_features_stereo_i16(_ffi.as_i16p(sig), sig.shape[0], float(sampling_rate), window, step,
                                             ratio, step_ratio, _ffi.as_f64p(mid), _ffi.as_f64p(st))
    else:
        rc = lib.paa_mid_features_f64(_ffi.as_f64p(sig), sig.shape[0], float(sampling_rate), window, step,
                                      ratio, step_ratio, _ffi.as_f64p(mid), _ffi.as_f64p(st))
    _ffi.check(rc)
    return mid, st, _mid_names(short_names)


def mid_feature_extraction_batch(signals, sampling_rate, mid_window, mid_step, short_window, short_step,
                                 return_short=False):
    """Many clips in one launch (all int16, or anything else as float64) -> list of (136, M_c) arrays (+ list of
    (68, T_c)), names."""
    ratio, step_ratio = _ratios(mid_window, mid_step, short_window, short_step)
    if step_ratio < 1:
        raise ValueError("mid_step / short_step rounds to 0: the reference never terminates")
    window, step = int(short_window), int(short_step)
    clips, as_i16 = ShortTermFeatures._batch_clips(signals)
    short_names = ShortTermFeatures._feature_names(True)
    F = len(short_names)
    lib = _ffi.lib()
    lens = np.array([c.shape[0] for c in clips], dtype=np.int64)
    offsets = np.zeros(len(clips) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    packed = np.concatenate(clips) if len(clips) > 1 else clips[0]
    T = np.array([int(lib.paa_num_frames(int(n), window, step)) for n in lens], dtype=np.int64)
    if np.any(T < 1):
        raise ValueError("need at least one array to concatenate")
    M = np.array([int(lib.paa_num_mid_windows(int(t), step_ratio)) for t in T], dtype=np.int64)
    mid_off = np.zeros(len(clips), dtype=np.int64)
    np.cumsum(2 * F * M[:-1], out=mid_off[1:])
    mid = np.empty(int(2 * F * M.sum()), dtype=np.float64)
    st = st_off = None
    if return_short:
        st_off = np.zeros(len(clips), dtype=np.int64)
        np.cumsum(F * T[:-1], out=st_off[1:])
        st = _ffi.result_array((int(F * T.sum()),))
    fn = lib.paa_mid_features_batch_i16 if as_i16 else lib.paa_mid_features_batch_f64
    _ffi.check(fn(
        _ffi.as_i16p(packed) if as_i16 else _ffi.as_f64p(packed), _ffi.as_i64p(offsets), len(clips), float(sampling_rate), window, step,
        ratio, step_ratio, _ffi.as_f64p(mid), _ffi.as_i64p(mid_off),
        _ffi.as_f64p(st) if return_short else None, _ffi.as_i64p(st_off) if return_short else None))
    mids = [mid[int(o):int(o) + 2 * F * int(m)].reshape(2 * F, int(m)) for o, m in zip(mid_off, M)]
    if return_short:
        sts = [st[int(o):int(o) + F * int(t)].reshape(F, int(t)) for o, t in zip(st_off, T)]
        return mids, sts, _mid_names(short_names)
    return mids, _mid_names(short_names)


# ---------------------------------------------------------------------------------------------------------
# beat extraction (reference :18-84 + utilities.peakdet, utilities.py:33-102)
# ---------------------------------------------------------------------------------------------------------
def beat_extraction(short_features, window_size, plot=False):
    """Estimate of the beat rate of a musical signal (reference :18-84): per row of the 18 short-term rows (:30-31) a
    threshold of twice the mean absolute difference, Billauer's peak detector (utilities.py:33-102), a histogram of the
    gaps between successive maxima; the rows' histograms / T are summed, arg-max -> (bpm, confidence).

    Runs on the GPU (beat_kernel through paa_beat_extraction_f64: one wave, lane r scans row r) -- the same kernel the
    batched directory walkers run on matrices that never leave HBM.  `plot` is accepted and ignored (the reference opens
    a matplotlib window with the histogram).

    ARGUMENTS: short_features (n_feats x numOfShortTermWindows), window_size = short-term step in seconds
    RETURNS:   bpm (beats per minute), ratio (confidence)
    """
    feats = np.ascontiguousarray(short_features, dtype=np.float64)
    if feats.ndim != 2:
        raise ValueError("short_features must be a (n_feats, n_frames) matrix")
    if feats.shape[0] < 19:
        raise IndexError("index 18 is out of bounds for axis 0 with size %d" % feats.shape[0])      # :31 reads row 18
    if feats.shape[1] < 1:
        raise ValueError("short_features has no frames")
    out = np.empty(2, dtype=np.float64)
    _ffi.check(_ffi.lib().paa_beat_extraction_f64(_ffi.as_f64p(feats), feats.shape[0], feats.shape[1],
                                                  float(window_size), _ffi.as_f64p(out)))
    return out[0], out[1]


# ---------------------------------------------------------------------------------------------------------
# directory walkers (reference :140-309)
# ---------------------------------------------------------------------------------------------------------
def _read_for_device(file_path):
    """(sampling_rate, signal) of one file; int16 stereo stays interleaved (it is reduced to mono on the device),
    everything else goes through audioBasicIO.stereo_to_mono as in the reference (:182-184)."""
    sampling_rate, signal = audioBasicIO.read_audio_file(file_path)
    if sampling_rate > 0 and not (signal.ndim == 2 and signal.shape[1] == 2 and signal.dtype == np.int16):
        signal = audioBasicIO.stereo_to_mono(signal)
    return sampling_rate, signal


def _read_all(paths):
    """Decode the files on a few host threads (file I/O and sample conversion release the GIL); results keep the
    order of `paths`, an exception of a reader surfaces when its result is taken."""
    if len(paths) < 2:
        return [(lambda p=p: _read_for_device(p)) for p in paths]
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(paths)))
    futures = [pool.submit(_read_for_device, p) for p in paths]
    pool.shutdown(wait=False)
    return [f.result for f in futures]


def _list_audio(folder_path, types):
    files = []
    for pattern in types:
        files.extend(glob.glob(os.path.join(folder_path, pattern)))
    return sorted(files)


def mid_and_beat_batch(signals, sampling_rate, mid_window, mid_step, short_window, short_step,
                       beat_window_seconds=None):
    """Device-resident batch: int16 clips -> list of (136, M_c) mid-term matrices and, when beat_window_seconds
    is given, an (n_clips, 2) array of (bpm, confidence) from the GPU beat kernel.  The short-term matrices
    never leave HBM.  The clips are either all mono (1-D int16), all interleaved stereo ((n, 2) int16) or all float64
    (1-D: what np.double() / stereo_to_mono make of every other file, ShortTermFeatures.py:567): stereo clips are
    uploaded interleaved as they come from the file (4 B per stereo frame); the kernels form L + R in their sample
    loads and scale by 2^-16, which is audioBasicIO.stereo_to_mono (audioBasicIO.py:156-168) followed by the 2^-15
    scaling of :568 -- no pass over the samples on the host."""
    ratio, step_ratio = _ratios(mid_window, mid_step, short_window, short_step)
    if step_ratio < 1:
        raise ValueError("mid_step / short_step rounds to 0: the reference never terminates")
    window, step = int(short_window), int(short_step)
    stereo = len(signals) > 0 and np.asarray(signals[0]).ndim == 2
    if any((np.asarray(s).ndim == 2) != stereo for s in signals):
        raise ValueError("mono and stereo clips cannot share a batch")
    floats = (not stereo) and len(signals) > 0 and np.asarray(signals[0]).dtype != np.int16
    if (not stereo) and any((np.asarray(s).dtype != np.int16) != floats for s in signals):
        raise ValueError("int16 and float64 clips cannot share a batch")
    if stereo:
        clips = [np.ascontiguousarray(s, dtype=np.int16) for s in signals]        # (n, 2): one stereo frame per row
    elif floats:
        clips = [np.ascontiguousarray(np.double(s)) for s in signals]
    else:
        clips = [np.ascontiguousarray(s, dtype=np.int16) for s in signals]
    lib = _ffi.lib()
    lens = np.array([c.shape[0] for c in clips], dtype=np.int64)
    if np.any(lens < window):
        raise ValueError("need at least one array to concatenate")
    offsets = np.zeros(len(clips) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    F = 68
    T = (lens - window) // step + 1
    M = -(-T // step_ratio)
    d_in = _ffi.DeviceBuffer.from_host((np.concatenate(clips) if len(clips) > 1 else clips[0]).reshape(-1))
    plan = _ffi.Plan(offsets, sampling_rate, window, step, deltas=True, sample_kind=2 if stereo else (1 if floats else 0))
    d_st = _ffi.DeviceBuffer(plan.out_doubles * 8)
    plan.execute(d_in, d_st)
    n_mid = plan.mid_doubles(step_ratio)
    d_mid = _ffi.DeviceBuffer(n_mid * 8)
    plan.mid_execute(d_st, ratio, step_ratio, d_mid)
    beats = None
    if beat_window_seconds is not None:
        d_beat = _ffi.DeviceBuffer(len(clips) * 16)
        plan.beat_execute(d_st, beat_window_seconds, d_beat)
        beats = d_beat.to_host(np.float64, 2 * len(clips)).reshape(len(clips), 2)
    flat = d_mid.to_host(np.float64, n_mid)
    mids, pos = [], 0
    for m in M:
        cnt = 2 * F * int(m)
        mids.append(flat[pos:pos + cnt].reshape(2 * F, int(m)))
        pos += cnt
    plan.destroy()
    return mids, beats


def _mid_for_files(entries, mid_window, mid_step, short_window, short_step, want_short):
    """entries: list of (sampling_rate, signal).  Clips that share a sampling rate and a sample layout go through ONE
    batched launch each: int16 mono, interleaved int16 stereo as read from the file (summed on the device), and float64
    -- which is what the reference's stereo_to_mono / np.double() make of every other file (8-bit, 32-bit, float,
    stereo of those; audioBasicIO.py:156-168, ShortTermFeatures.py:567), so those files no longer take the reference's
    one-by-one loop (MidTermFeatures.py:167).  Returns per-entry (mid [136 x M], short [68 x T] or None, names)."""
    out = [None] * len(entries)
    names = _mid_names(ShortTermFeatures._feature_names(True))
    groups = {}
    signals = [None] * len(entries)
    for idx, (fs, sig) in enumerate(entries):
        a = np.asarray(sig)
        if a.dtype == np.int16 and (a.ndim == 1 or (a.ndim == 2 and a.shape[1] == 2)):
            signals[idx] = a
            groups.setdefault((fs, "i16", a.ndim), []).append(idx)
            continue
        mono = np.asarray(audioBasicIO.stereo_to_mono(a))
        if mono.ndim == 1:
            signals[idx] = np.double(mono)
            groups.setdefault((fs, "f64", 1), []).append(idx)
        else:               # more than two channels: the reference fails on these too; keep its single-clip error path
            mid, st, _ = mid_feature_extraction(sig, fs, round(mid_window * fs), round(mid_step * fs),
                                                round(fs * short_window), round(fs * short_step))
            out[idx] = (mid, st if want_short else None)
    for (fs, _, _), members in groups.items():
        # want_short here means "the caller needs the beat": computed on the GPU from the resident short-term
        # matrix, returned in place of the matrix as a (bpm, confidence) pair
        mids, beats = mid_and_beat_batch([signals[i] for i in members], fs, round(mid_window * fs),
                                         round(mid_step * fs), round(fs * short_window), round(fs * short_step),
                                         beat_window_seconds=short_step if want_short else None)
        for k, i in enumerate(members):
            out[i] = (mids[k], tuple(beats[k]) if want_short else None)
    return out, names


def directory_feature_extraction(folder_path, mid_window, mid_step, short_window, short_step, compute_beat=True):
    """One long-term-averaged feature vector per audio file of a folder (reference :140-221).

    Windows and steps are in SECONDS here (the reference multiplies by each file's sampling rate, :187-190).
    RETURNS (features [n_files x 136(+2)], kept file list, feature names)
    """
    mid_term_features = np.array([])
    wav_file_list = _list_audio(folder_path, ('*.wav', '*.aif', '*.aiff', '*.mp3', '*.au', '*.ogg'))
    kept, entries = [], []
    t_start = time.time()
    non_empty = [p for p in wav_file_list if os.stat(p).st_size > 0]
    readers = dict(zip(non_empty, _read_all(non_empty)))
    for i, file_path in enumerate(wav_file_list):
        print("Analyzing file {0:d} of {1:d}: {2:s}".format(i + 1, len(wav_file_list), file_path))
        if file_path not in readers:
            print("   (EMPTY FILE -- SKIPPING)")
            continue
        sampling_rate, signal = readers[file_path]()
        if sampling_rate <= 0:
            continue
        if signal.shape[0] < float(sampling_rate) / 5:
            print("  (AUDIO FILE TOO SMALL - SKIPPING)")
            continue
        kept.append(file_path)
        entries.append((sampling_rate, signal))
    mid_feature_names = []
    total_audio = 0.0
    if entries:
        results, mid_feature_names = _mid_for_files(entries, mid_window, mid_step, short_window, short_step,
                                                    want_short=compute_beat)
        mid_feature_names = list(mid_feature_names)
        added_names = False
        for (fs, signal), (mid, st) in zip(entries, results):
            vec = np.transpose(mid).mean(axis=0)                       # long-term averaging (:199-201)
            if (not np.isnan(vec).any()) and (not np.isinf(vec).any()):
                if compute_beat:
                    # (bpm, confidence) straight from the GPU beat kernel (a host scan only for the single-clip path)
                    beat, beat_conf = st if isinstance(st, tuple) else beat_extraction(st, short_step)
                    vec = np.append(vec, beat)
                    vec = np.append(vec, beat_conf)
                    if not added_names:
                        mid_feature_names += ["bpm", "ratio"]
                        added_names = True
                mid_term_features = vec if len(mid_term_features) == 0 else np.vstack((mid_term_features, vec))
                total_audio += float(len(signal)) / fs
    elapsed = time.time() - t_start
    if total_audio > 0 and elapsed > 0:
        print("Feature extraction complexity ratio: {0:.1f} x realtime".format(total_audio / elapsed))
    return mid_term_features, kept, mid_feature_names


def multiple_directory_feature_extraction(path_list, mid_window, mid_step, short_window, short_step,
                                          compute_beat=False):
    """List of folders -> list of feature matrices, class names (folder names), file lists (reference :224-260)."""
    features, class_names, file_names = [], [], []
    for d in path_list:
        f, fn, _ = directory_feature_extraction(d, mid_window, mid_step, short_window, short_step,
                                                compute_beat=compute_beat)
        if f.shape[0] > 0:
            features.append(f)
            file_names.append(fn)
            if d[-1] == os.sep:
                class_names.append(d.split(os.sep)[-2])
            else:
                class_names.append(d.split(os.sep)[-1])
    return features, class_names, file_names


def directory_feature_extraction_no_avg(folder_path, mid_window, mid_step, short_window, short_step):
    """All mid-term vectors of every file, no averaging (reference :263-309).
    RETURNS (X [sum_M x 136], file index of every row, file list)"""
    wav_file_list = _list_audio(folder_path, ('*.wav', '*.aif', '*.aiff', '*.ogg'))
    entries, index = [], []
    for i, (file_path, reader) in enumerate(zip(wav_file_list, _read_all(wav_file_list))):
        sampling_rate, signal = reader()
        if sampling_rate <= 0:
            continue
        entries.append((sampling_rate, signal))
        index.append(i)
    mid_features = np.array([])
    signal_idx = np.array([])
    if entries:
        results, _ = _mid_for_files(entries, mid_window, mid_step, short_window, short_step, want_short=False)
        for i, (mid, _) in zip(index, results):
            vec = np.transpose(mid)
            if len(mid_features) == 0:
                mid_features = vec
                signal_idx = np.zeros((vec.shape[0],))
            else:
                mid_features = np.vstack((mid_features, vec))
                signal_idx = np.append(signal_idx, i * np.ones((vec.shape[0],)))
    return mid_features, signal_idx, wav_file_list


# ---------------------------------------------------------------------------------------------------------
# file writers (reference :324-377): feature sequences of one file / every WAV of a folder -> .npy (.csv)
# ---------------------------------------------------------------------------------------------------------
def mid_feature_extraction_to_file(file_path, mid_window, mid_step, short_window, short_step, output_file,
                                   store_short_features=False, store_csv=False, plot=False):
    """Read one audio file, extract its mid-term (and optionally short-term) feature sequences on the GPU and
    save them as <output_file>_mt.npy / _st.npy (and .csv, one row per window) -- no long-term averaging."""
    sampling_rate, signal = audioBasicIO.read_audio_file(file_path)
    if not (signal.ndim == 2 and signal.shape[1] == 2 and signal.dtype == np.int16):
        signal = audioBasicIO.stereo_to_mono(signal)          # int16 stereo is reduced to mono on the device
    mid_features, short_features, _ = mid_feature_extraction(signal, sampling_rate,
                                                             round(sampling_rate * mid_window),
                                                             round(sampling_rate * mid_step),
                                                             round(sampling_rate * short_window),
                                                             round(sampling_rate * short_step))
    outputs = [("_mt", "Mid-term", mid_features)]
    if store_short_features:
        outputs.insert(0, ("_st", "Short-term", short_features))
    for suffix, label, matrix in outputs:
        np.save(output_file + suffix, matrix)
        if plot:
            print(label + " np file: " + output_file + suffix + ".npy saved")
        if store_csv:
            np.savetxt(output_file + suffix + ".csv", matrix.T, delimiter=",")
            if plot:
                print(label + " CSV file: " + output_file + suffix + ".csv saved")


def mid_feature_extraction_file_dir(folder_path, mid_window, mid_step, short_window, short_step,
                                    store_short_features=False, store_csv=False, plot=False):
    """mid_feature_extraction_to_file for every *.wav of a folder; outputs are written next to the inputs."""
    for wav in glob.glob(folder_path + os.sep + '*.wav'):
        mid_feature_extraction_to_file(wav, mid_window, mid_step, short_window, short_step, wav,
                                       store_short_features, store_csv, plot)

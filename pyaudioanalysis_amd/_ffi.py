"""ctypes binding of libpaa_hip.so (include/paa_hip.h).  No CPU fallback: every compute call
raises when the library or a HIP device is missing."""
import ctypes as C
import os
import threading

import numpy as np

from . import _build

PAA_OK = 0
ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_OOM = -1, -2, -3, -4
ERR_TOO_SHORT, ERR_CHROMA_VALUE, ERR_CHROMA_INDEX, ERR_MEL_INDEX, ERR_COMM = -5, -6, -7, -8, -9
COMM_ID_BYTES = 128

_lock = threading.Lock()
_lib = None

c_i16p = C.POINTER(C.c_int16)
c_f64p = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)


class HipLibraryError(RuntimeError):
    """libpaa_hip.so is missing / unbuildable, or the HIP runtime reported an error."""


_SIGNATURES = {
    "paa_version": (C.c_char_p, []),
    "paa_last_error": (C.c_char_p, []),
    "paa_device_count": (C.c_int, []),
    "paa_init": (C.c_int, [C.c_int]),
    "paa_shutdown": (None, []),
    "paa_device_bus_id": (C.c_int, [C.c_char_p, C.c_int]),
    "paa_dev_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "paa_dev_free": (C.c_int, [C.c_void_p]),
    "paa_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "paa_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "paa_memcpy_d2h_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "paa_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "paa_dev_sync": (C.c_int, []),
    "paa_timer_start": (C.c_int, []),
    "paa_timer_stop": (C.c_int, [C.POINTER(C.c_float)]),
    "paa_prof_enable": (C.c_int, [C.c_int]),
    "paa_prof_read": (C.c_int, [C.POINTER(C.c_double), c_i64p]),
    "paa_num_frames": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "paa_num_mid_windows": (C.c_int64, [C.c_int64, C.c_int64]),
    "paa_spectrogram_rows": (C.c_int64, [C.c_int64, C.c_int, C.c_int, c_i64p]),
    "paa_chromagram_rows": (C.c_int64, [C.c_int64, C.c_int, C.c_int, c_i64p]),
    "paa_st_features_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, c_f64p]),
    "paa_st_features_f64": (C.c_int, [c_f64p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, c_f64p]),
    "paa_st_features_stereo_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, c_f64p]),
    "paa_mid_features_stereo_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                              c_f64p, c_f64p]),
    "paa_mid_features_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                       c_f64p, c_f64p]),
    "paa_mid_features_f64": (C.c_int, [c_f64p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                       c_f64p, c_f64p]),
    "paa_spectrogram_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, c_f64p]),
    "paa_spectrogram_f64": (C.c_int, [c_f64p, C.c_int64, C.c_double, C.c_int, C.c_int, c_f64p]),
    "paa_chromagram_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, c_f64p]),
    "paa_chromagram_f64": (C.c_int, [c_f64p, C.c_int64, C.c_double, C.c_int, C.c_int, c_f64p]),
    "paa_spectrogram_stereo_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, c_f64p]),
    "paa_chromagram_stereo_i16": (C.c_int, [c_i16p, C.c_int64, C.c_double, C.c_int, C.c_int, c_f64p]),
    "paa_st_features_batch_f64": (C.c_int, [c_f64p, c_i64p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int,
                                            c_f64p, c_i64p]),
    "paa_mid_features_batch_f64": (C.c_int, [c_f64p, c_i64p, C.c_int64, C.c_double, C.c_int, C.c_int,
                                             C.c_int64, C.c_int64, c_f64p, c_i64p, c_f64p, c_i64p]),
    "paa_st_features_batch_i16": (C.c_int, [c_i16p, c_i64p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int,
                                            c_f64p, c_i64p]),
    "paa_mid_features_batch_i16": (C.c_int, [c_i16p, c_i64p, C.c_int64, C.c_double, C.c_int, C.c_int,
                                             C.c_int64, C.c_int64, c_f64p, c_i64p, c_f64p, c_i64p]),
    "paa_plan_create": (C.c_int, [c_i64p, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_void_p)]),
    "paa_plan_create_mode": (C.c_int, [c_i64p, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)]),
    "paa_plan_destroy": (C.c_int, [C.c_void_p]),
    "paa_plan_total_frames": (C.c_int64, [C.c_void_p]),
    "paa_plan_out_doubles": (C.c_int64, [C.c_void_p]),
    "paa_plan_out_offsets": (C.c_int, [C.c_void_p, c_i64p]),
    "paa_plan_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "paa_plan_mid_doubles": (C.c_int64, [C.c_void_p, C.c_int64]),
    "paa_plan_mid_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "paa_plan_beat_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "paa_plan_kernel_name": (C.c_char_p, [C.c_void_p]),
    "paa_beat_extraction_f64": (C.c_int, [c_f64p, C.c_int, C.c_int64, C.c_double, c_f64p]),
    "paa_dev_expand_deltas": (C.c_int, [C.c_void_p, c_i64p, C.c_int64, C.c_void_p]),
    "paa_self_similarity_f64": (C.c_int, [c_f64p, C.c_int, C.c_int64, c_f64p]),
    "paa_dev_self_similarity": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "paa_thumbnail_rows": (C.c_int64, [C.c_int64, C.c_int]),
    "paa_thumbnail_f64": (C.c_int, [c_f64p, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, c_f64p,
                                    c_i64p]),
    "paa_dev_thumbnail_filter": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double,
                                           C.c_void_p, c_i64p]),
    "paa_svm_binary_proba_f64": (C.c_int, [c_f64p, C.c_int, C.c_int64, c_f64p, c_f64p, c_f64p, c_f64p, C.c_int, C.c_double,
                                           C.c_double, C.c_double, C.c_double, c_f64p]),
    "paa_comm_unique_id": (C.c_int, [C.c_void_p]),
    "paa_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "paa_comm_destroy": (C.c_int, []),
    "paa_comm_gather_f64": (C.c_int, [C.c_void_p, c_i64p, C.c_int, C.c_void_p]),
    "paa_comm_gatherv_f64": (C.c_int, [C.c_void_p, c_i64p, c_i64p, C.c_int, C.c_void_p]),
    "paa_comm_barrier": (C.c_int, []),
    "paa_debug_mel_bank": (C.c_int, [C.c_double, C.c_int, c_f64p]),
    "paa_debug_dct": (C.c_int, [c_f64p]),
    "paa_debug_chroma": (C.c_int, [C.c_double, C.c_int, C.c_int, c_i32p, c_f64p, c_i32p]),
    "paa_debug_phase_cycles": (C.c_int, [C.POINTER(C.c_uint64)]),
    "paa_debug_wave_trace": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "paa_debug_lane_peak": (C.c_int, []),
    "paa_debug_fft_plan": (C.c_int, [C.c_int, c_i32p, c_i32p]),
    "paa_debug_comm_marker_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "paa_debug_tri_plan": (C.c_int, [C.c_int, C.c_double, c_i32p, c_i32p, C.c_void_p, C.c_int]),
    "paa_debug_wg_plan": (C.c_int, [C.c_int, c_i32p, C.POINTER(C.c_uint16), C.c_int]),
    "paa_debug_wgs_plan": (C.c_int, [C.c_int, c_i32p, c_i32p, C.c_int]),
    "paa_debug_wgr_tables": (C.c_int, [C.c_double, C.c_int, c_i32p, c_i32p, c_i32p, c_i32p, c_f64p]),
    "paa_debug_wgr_runs": (C.c_int, [c_i64p, C.c_int64, C.c_int, c_i32p, C.c_int64, c_i64p]),
    "paa_debug_blu_plan": (C.c_int, [C.c_int, C.c_double, c_i32p, c_i32p, C.c_void_p, C.c_int]),
    "paa_debug_lane_jobs": (C.c_int, [c_i32p, c_i32p, c_i32p, C.c_int, c_i32p]),
    "paa_debug_mix_plan": (C.c_int, [C.c_int, c_i32p, c_i32p, C.POINTER(C.c_uint16), C.c_int, c_i32p, c_i32p]),
    "paa_debug_run_plan": (C.c_int, [c_i64p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p,
                                     c_i64p, c_i32p]),
    "paa_debug_run_plan_shrink": (C.c_int, [c_i64p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p,
                                     c_i64p, c_i32p]),
    "paa_debug_balanced_runs": (C.c_int64, [c_i64p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p,
                                            C.c_int64]),
}
EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


def library_path():
    """The in-tree library; PAA_HIP_LIBRARY names another build of the same sources (the sanitizer build of
    tests/test_sanitizer_cpu.py), which is then loaded as is."""
    return os.environ.get("PAA_HIP_LIBRARY") or _build.LIB


def lib():
    """Load (building first if the sources are newer) libpaa_hip.so; raises HipLibraryError."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if path == _build.LIB and _build.is_stale():
            try:
                _build.hipcc_path()
            except RuntimeError as exc:       # no hipcc on this host: only a prebuilt library can serve
                if not os.path.exists(path):
                    raise HipLibraryError(
                        "libpaa_hip.so is not built and cannot be built here (%s). "
                        "pyaudioanalysis_amd has no CPU path." % exc)
            else:
                try:
                    _build.build()
                except Exception as exc:      # a compile error must never fall back to a stale binary
                    raise HipLibraryError("libpaa_hip.so is older than its sources and the rebuild failed:\n%s" % exc)
        try:
            handle = C.CDLL(path)
        except OSError as exc:
            raise HipLibraryError("cannot load %s: %s" % (path, exc))
        for name, (restype, argtypes) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise HipLibraryError("%s does not export %s" % (path, name))
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
        return _lib


def last_error():
    msg = lib().paa_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a negative return code to the exception TYPE the reference raises in that situation."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == ERR_TOO_SHORT:
        raise ValueError(msg or "need at least one array to concatenate")      # ShortTermFeatures.py:684
    if rc == ERR_CHROMA_VALUE:
        raise ValueError(msg)                                                   # :293
    if rc in (ERR_CHROMA_INDEX, ERR_MEL_INDEX):
        raise IndexError(msg)                                                   # :291 / :230
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_OOM:
        raise MemoryError(msg)
    raise HipLibraryError(msg or "HIP error %d" % rc)


def device_count():
    n = lib().paa_device_count()
    return max(n, 0)


def init(device_id=0):
    check(lib().paa_init(int(device_id)))


def as_i16p(a):
    return a.ctypes.data_as(c_i16p)


def as_f64p(a):
    return a.ctypes.data_as(c_f64p)


def as_i64p(a):
    return a.ctypes.data_as(c_i64p)


def classify_signal(signal):
    """Return (kind, contiguous array): kind 0 = int16 PCM, 1 = float64.

    The reference converts everything with np.double() (ShortTermFeatures.py:567); int16 is kept as
    int16 (2 B/sample over PCIe and HBM), every other dtype goes through the same np.double().
    """
    a = np.asarray(signal)
    if a.ndim == 2 and a.shape[1] == 2 and a.dtype == np.int16:
        # extension over the reference (which needs stereo_to_mono first): interleaved stereo int16 goes to the
        # device as is and is summed there (kind 2)
        return 2, np.ascontiguousarray(a)
    if a.ndim != 1:
        a = a.reshape(-1) if a.ndim == 0 else a
        if a.ndim != 1:
            raise ValueError("signal must be one-dimensional (the reference does not support %d-D input)" % a.ndim)
    if a.dtype == np.int16:
        return 0, np.ascontiguousarray(a)
    return 1, np.ascontiguousarray(np.double(a))


class DeviceBuffer:
    """A hipMalloc'd buffer owned by Python (bench / pipelines)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().paa_dev_alloc(self.nbytes, C.byref(p)))
        self.ptr = p

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        check(lib().paa_memcpy_h2d(buf.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return buf

    def to_host(self, dtype, count, offset_bytes=0, wait_comm=True):
        """wait_comm=False: ordered behind the compute stream only (a buffer that is at most the SOURCE of queued gathers)."""
        out = np.empty(int(count), dtype=dtype)
        src = C.c_void_p(self.ptr.value + int(offset_bytes))
        fn = lib().paa_memcpy_d2h if wait_comm else lib().paa_memcpy_d2h_compute
        check(fn(out.ctypes.data_as(C.c_void_p), src, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            lib().paa_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Plan:
    """Device-resident batch plan (paa_plan_*): samples and results stay in HBM."""

    def __init__(self, offsets, fs, window, step, deltas=True, sample_kind=0, mode=0):
        """mode 0: short-term features; 1: spectrogram rows; 2: chromagram rows (full-length frames, see paa_hip.h)."""
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n_clips = len(self.offsets) - 1
        h = C.c_void_p()
        if mode == 0:
            check(lib().paa_plan_create(as_i64p(self.offsets), self.n_clips, int(sample_kind), float(fs), int(window),
                                        int(step), 1 if deltas else 0, C.byref(h)))
        else:
            check(lib().paa_plan_create_mode(as_i64p(self.offsets), self.n_clips, int(sample_kind), float(fs),
                                             int(window), int(step), int(mode), C.byref(h)))
        self.handle = h
        self.mode = int(mode)
        self.F = (68 if deltas else 34) if mode == 0 else (int(window) // 2 if mode == 1 else 12)
        self.total_frames = int(lib().paa_plan_total_frames(h))
        self.out_doubles = int(lib().paa_plan_out_doubles(h))
        self.kernel_name = lib().paa_plan_kernel_name(h).decode()

    def out_offsets(self):
        o = np.empty(self.n_clips, dtype=np.int64)
        check(lib().paa_plan_out_offsets(self.handle, as_i64p(o)))
        return o

    def execute(self, d_packed, d_out):
        check(lib().paa_plan_execute(self.handle, d_packed.ptr, d_out.ptr))

    def mid_doubles(self, mid_step_ratio):
        return int(lib().paa_plan_mid_doubles(self.handle, int(mid_step_ratio)))

    def mid_execute(self, d_st, mid_ratio, mid_step_ratio, d_mid):
        check(lib().paa_plan_mid_execute(self.handle, d_st.ptr, int(mid_ratio), int(mid_step_ratio), d_mid.ptr))

    def beat_execute(self, d_st, window_size, d_beat):
        check(lib().paa_plan_beat_execute(self.handle, d_st.ptr, float(window_size), d_beat.ptr))

    def destroy(self):
        if self.handle is not None:
            lib().paa_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def sync():
    check(lib().paa_dev_sync())


# ---- recycled result arrays ------------------------------------------------------------------------------
# A fresh 39 MB NumPy array (one hour of features) costs ~2 ms of mmap/munmap plus ~1 ms of first-touch page faults during
# the device-to-host copy -- as much as the copy itself.  Result arrays >= 1 MB are therefore views of pooled storage.
# Liveness is tracked explicitly, not through reference counts: every hand-out wraps the storage in a fresh *token* array
# (np.frombuffer over a memoryview: its base is not an ndarray, so NumPy's base-collapsing stops at the token and every
# view, slice or reshape the caller derives from the result keeps the TOKEN alive); a weakref.finalize on the token marks
# the storage idle when the last such view is gone.  The caller owns what it gets for as long as it keeps any view of it,
# exactly the reference's contract; the arrays are non-owning views (flags.owndata is False, in-place resize is refused).
import weakref

_POOL_LOCK = threading.Lock()
_POOL = []                      # [storage (uint8 ndarray, owns the memory), idle flag]
_POOL_MAX_BYTES = 1 << 30
_POOL_MIN_BYTES = 1 << 20


def _retire(entry):
    entry[1] = True             # (list item assignment: atomic under the GIL; finalizers may run on any thread)


def result_array(shape, dtype=np.float64):
    """Uninitialised C-contiguous array of `shape`, from the pool when an idle block fits."""
    n = int(np.prod(shape))
    nbytes = n * np.dtype(dtype).itemsize
    if nbytes < _POOL_MIN_BYTES:
        return np.empty(shape, dtype=dtype)
    with _POOL_LOCK:
        best = None
        for entry in _POOL:
            if entry[1] and entry[0].nbytes >= nbytes and (best is None or entry[0].nbytes < best[0].nbytes):
                best = entry
        if best is None or best[0].nbytes > 2 * nbytes + (1 << 22):
            best = [np.empty(nbytes, dtype=np.uint8), True]
            _POOL.append(best)
            total = sum(e[0].nbytes for e in _POOL)
            k = 0
            while total > _POOL_MAX_BYTES and k < len(_POOL):       # drop idle blocks, oldest first
                if _POOL[k][1] and _POOL[k] is not best:
                    total -= _POOL[k][0].nbytes
                    del _POOL[k]
                else:
                    k += 1
        best[1] = False
        token = np.frombuffer(memoryview(best[0]), dtype=np.uint8)
        fin = weakref.finalize(token, _retire, best)
        fin.atexit = False
        return token[:nbytes].view(dtype).reshape(shape)

/*
 * paa_hip.h -- C ABI of libpaa_hip.so, the MI355X (gfx950) short-term / mid-term audio
 * feature extractor that is a drop-in for ONE path of tyiannak/pyAudioAnalysis.
 *
 * The reference has no FFI layer: its boundary is the Python signature.  Each entry point
 * below names the reference interface it replaces (paths relative to
 * /root/reference/pyAudioAnalysis).  Plain pointers and sizes only; no torch / numpy types.
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - return value 0 (PAA_OK) = success, < 0 = error code; paa_last_error() gives the text
 *     (thread local).  There is NO CPU fallback: without a HIP device every compute call
 *     returns PAA_ERR_HIP.
 *   - `window` / `step` are in samples and already int()-truncated by the caller
 *     (ShortTermFeatures.py:563-564); num_fft = window / 2 (integer division, :575).
 *   - host buffers are caller-owned and only touched during the call.  `*_dev_*` entry points
 *     take HIP device pointers (from paa_dev_alloc or any hipMalloc) and are asynchronous on
 *     the library's stream; paa_dev_sync() waits for them.
 *   - feature matrices are float64, C-contiguous, FEATURE-major [F][T] (F = 34, or 68 with
 *     deltas), exactly the reference's return layout (:684); spectrogram / chromagram are
 *     TIME-major [T'][num_fft] / [T''][12] (:413, :347).
 *   - a batch is a packed sample buffer plus n_clips+1 sample offsets; clip c's result is the
 *     [F][T_c] slab starting at out + out_offsets[c] (in doubles).
 */
#ifndef PAA_HIP_H
#define PAA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAA_OK                 0
#define PAA_ERR_ARG          (-1)  /* bad argument (null pointer, window < 2, step < 1, ...)          */
#define PAA_ERR_UNSUPPORTED  (-2)  /* window beyond the LDS envelope of the kernels                    */
#define PAA_ERR_HIP          (-3)  /* HIP runtime error / no device                                    */
#define PAA_ERR_OOM          (-4)  /* device or host allocation failed                                 */
#define PAA_ERR_TOO_SHORT    (-5)  /* fewer samples than one window: reference raises ValueError :684  */
#define PAA_ERR_CHROMA_VALUE (-6)  /* chroma slot > num_fft: reference raises ValueError :293          */
#define PAA_ERR_CHROMA_INDEX (-7)  /* chroma slot == num_fft: reference raises IndexError :291         */
#define PAA_ERR_MEL_INDEX    (-8)  /* mel filter bin >= num_fft: reference raises IndexError :230-231  */
#define PAA_ERR_COMM         (-9)  /* RCCL error                                                       */

#define PAA_N_BASE_FEATURES 34

/* ---- library / device management ------------------------------------------------------ */
const char *paa_version(void);
const char *paa_last_error(void);
int  paa_device_count(void);             /* >= 0, or PAA_ERR_HIP                                   */
int  paa_init(int device_id);            /* select the device for this process (one process per GPU) */
void paa_shutdown(void);                 /* free cached tables, scratch and the stream              */
/* PCI bus id of the selected device ("0000:75:00.0", NUL-terminated): names the physical device whatever
 * HIP_VISIBLE_DEVICES says; used to refuse two ranks of one RCCL job on one device                  */
int  paa_device_bus_id(char *out, int capacity);
int  paa_dev_alloc(size_t bytes, void **out_ptr);
int  paa_dev_free(void *ptr);
int  paa_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);   /* synchronous */
int  paa_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes);    /* queued on the library stream */
int  paa_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);   /* synchronous */
/* the same, ordered behind the kernels of the compute stream only -- not behind gathers still queued on the communication
 * stream.  For buffers that are at most the SOURCE of a gather (distributed.py's restart shards are saved with it while the
 * exchange may still be waiting for a peer); a gather DESTINATION must be read with paa_memcpy_d2h                    */
int  paa_memcpy_d2h_compute(void *dst_host, const void *src_dev, size_t bytes);
int  paa_dev_sync(void);
/* elapsed GPU milliseconds between two points of the library stream (HIP events) */
int  paa_timer_start(void);
int  paa_timer_stop(float *ms);

/* HIP-event timing of the feature kernel inside paa_plan_execute (off by default): every_nth = 1 brackets every
 * launch with an event pair, n > 1 every n-th one (an event pair is an ordering point on the stream, so the
 * recorder itself slows a stream of back-to-back launches by ~4 % at n = 1), 0 switches it off;
 * paa_prof_read returns the accumulated milliseconds / number of timed launches and resets them */
int  paa_prof_enable(int every_nth);
int  paa_prof_read(double *total_ms, int64_t *launches);

/* ---- shape helpers ----------------------------------------------------------------------- */
/* T = floor((n - window)/step) + 1, 0 if n < window            (ShortTermFeatures.py:608)   */
int64_t paa_num_frames(int64_t n_samples, int window, int step);
/* M = ceil(T / mid_step_ratio)                                  (MidTermFeatures.py:116-124) */
int64_t paa_num_mid_windows(int64_t n_frames, int64_t mid_step_ratio);
/* rows allocated by spectrogram: int((n-window)/step)+1; rows actually filled in *filled     */
int64_t paa_spectrogram_rows(int64_t n_samples, int window, int step, int64_t *filled);
/* rows allocated by chromagram: int((n-step-window)/step)+1; filled rows in *filled          */
int64_t paa_chromagram_rows(int64_t n_samples, int window, int step, int64_t *filled);

/* ---- ShortTermFeatures.feature_extraction (ShortTermFeatures.py:543-685) ---------------- */
/* signal: int16 PCM as scipy.io.wavfile returns it (audioBasicIO.py:99), or float64
 * (after stereo_to_mono, audioBasicIO.py:167).  Both are scaled by 1/2^15 (:568).
 * out: [F][T] doubles, F = 34 * (deltas ? 2 : 1).                                         */
int paa_st_features_i16(const int16_t *signal, int64_t n, double fs, int window, int step,
                        int deltas, double *out);
int paa_st_features_f64(const double *signal, int64_t n, double fs, int window, int step,
                        int deltas, double *out);

/* interleaved stereo int16 (L0 R0 L1 R1 ..., n frames): audioBasicIO.stereo_to_mono (audioBasicIO.py:156-168) is
 * fused into the kernels' sample loads (L + R summed exactly as it is fetched, scaled by 2^-16): the mono signal is
 * never materialised; stereo files cost 4 B/sample over PCIe and HBM instead of the 8 B of the float64 mono copy the
 * reference makes on the host                                                                              */
int paa_st_features_stereo_i16(const int16_t *interleaved, int64_t n, double fs, int window, int step,
                               int deltas, double *out);
int paa_mid_features_stereo_i16(const int16_t *interleaved, int64_t n, double fs, int window, int step,
                                int64_t mid_ratio, int64_t mid_step_ratio, double *mid_out, double *st_out);

/* ---- MidTermFeatures.mid_feature_extraction (MidTermFeatures.py:87-127) ----------------- */
/* mid_ratio / mid_step_ratio are computed by the caller with Python round() (:100-102).
 * st_out: [68][T] (deltas always on, :93-95), may be NULL; mid_out: [136][M].               */
int paa_mid_features_i16(const int16_t *signal, int64_t n, double fs, int window, int step,
                         int64_t mid_ratio, int64_t mid_step_ratio, double *mid_out, double *st_out);
int paa_mid_features_f64(const double *signal, int64_t n, double fs, int window, int step,
                         int64_t mid_ratio, int64_t mid_step_ratio, double *mid_out, double *st_out);

/* ---- ShortTermFeatures.spectrogram / chromagram (:389-452 / :324-386) ------------------- */
/* out has paa_spectrogram_rows() x (window/2) doubles; unfilled trailing rows are zeroed.   */
int paa_spectrogram_i16(const int16_t *signal, int64_t n, double fs, int window, int step, double *out);
int paa_spectrogram_f64(const double *signal, int64_t n, double fs, int window, int step, double *out);
/* out has paa_chromagram_rows() x 12 doubles.  The reference may FFT a truncated last frame
 * (:349-355); that frame is evaluated on the device as a direct DFT of its true length.      */
int paa_chromagram_i16(const int16_t *signal, int64_t n, double fs, int window, int step, double *out);
int paa_chromagram_f64(const double *signal, int64_t n, double fs, int window, int step, double *out);
/* interleaved stereo int16 (n frames): audioAnalysis.fileSpectrogramWrapper / fileChromagramWrapper call stereo_to_mono
 * first (audioAnalysis.py:66-81, audioBasicIO.py:156-168); here the channels are summed on the device (4 B/frame over PCIe
 * instead of the 8 B of the float64 mono copy)                                                                  */
int paa_spectrogram_stereo_i16(const int16_t *interleaved, int64_t n, double fs, int window, int step, double *out);
int paa_chromagram_stereo_i16(const int16_t *interleaved, int64_t n, double fs, int window, int step, double *out);

/* ---- batched many-clip path (what MidTermFeatures.directory_feature_extraction :140-221
 *      does sequentially per file)                                                          */
int paa_st_features_batch_i16(const int16_t *packed, const int64_t *offsets, int64_t n_clips,
                              double fs, int window, int step, int deltas,
                              double *out, const int64_t *out_offsets);
/* float64 clips (what stereo_to_mono / np.double() hand the reference for stereo, 8-bit, 32-bit or float files,
 * audioBasicIO.py:167, ShortTermFeatures.py:567): the same batch, 8 B/sample                 */
int paa_st_features_batch_f64(const double *packed, const int64_t *offsets, int64_t n_clips,
                              double fs, int window, int step, int deltas,
                              double *out, const int64_t *out_offsets);
int paa_mid_features_batch_f64(const double *packed, const int64_t *offsets, int64_t n_clips,
                               double fs, int window, int step,
                               int64_t mid_ratio, int64_t mid_step_ratio,
                               double *mid_out, const int64_t *mid_out_offsets,
                               double *st_out, const int64_t *st_out_offsets);
/* mid_out_offsets index [136][M_c] slabs; st_out / st_out_offsets may be NULL               */
int paa_mid_features_batch_i16(const int16_t *packed, const int64_t *offsets, int64_t n_clips,
                               double fs, int window, int step,
                               int64_t mid_ratio, int64_t mid_step_ratio,
                               double *mid_out, const int64_t *mid_out_offsets,
                               double *st_out, const int64_t *st_out_offsets);

/* ---- device-resident plans (bench / pipelines: samples and results stay in HBM) --------- */
typedef struct paa_plan paa_plan_t;
/* offsets: n_clips+1 HOST sample offsets into the packed device buffer.  sample_kind 0 = int16,
 * 1 = float64, 2 = interleaved stereo int16 (offsets count stereo frames of 4 bytes; L + R is formed in the kernels'
 * loads and scaled by 2^-16 = stereo_to_mono followed by :568).  The plan owns the tables, the tile list and the
 * per-clip statistics.                                                                                      */
int paa_plan_create(const int64_t *offsets, int64_t n_clips, int sample_kind, double fs,
                    int window, int step, int deltas, paa_plan_t **out_plan);
/* spectrogram (mode 1) / chromagram (mode 2) rows kept in HBM (ShortTermFeatures.py:389-452, :324-386); mode 0 = features
 * without deltas.  paa_plan_out_doubles() gives the size of the row block, paa_plan_total_frames() the full-length frames. */
int paa_plan_create_mode(const int64_t *offsets, int64_t n_clips, int sample_kind, double fs, int window, int step,
                         int mode, paa_plan_t **out_plan);
int paa_plan_destroy(paa_plan_t *plan);
int64_t paa_plan_total_frames(const paa_plan_t *plan);
int64_t paa_plan_out_doubles(const paa_plan_t *plan);     /* sum_c F*T_c                      */
/* out_offsets (HOST, n_clips) receives the slab start of every clip when non-NULL           */
int paa_plan_out_offsets(const paa_plan_t *plan, int64_t *out_offsets);
/* asynchronous on the library stream: clip statistics -> features (+deltas) into d_out       */
int paa_plan_execute(paa_plan_t *plan, const void *d_packed, double *d_out);
/* optional mid-term statistics of an executed plan (deltas must be on): d_mid gets
 * [136][M_c] slabs back to back; returns total doubles via paa_plan_mid_doubles             */
int64_t paa_plan_mid_doubles(const paa_plan_t *plan, int64_t mid_step_ratio);
int paa_plan_mid_execute(paa_plan_t *plan, const double *d_st, int64_t mid_ratio,
                         int64_t mid_step_ratio, double *d_mid);
/* MidTermFeatures.beat_extraction (MidTermFeatures.py:18-84) for every clip of an executed plan:
 * d_beat receives [n_clips][2] = (bpm, confidence); window_size = short-term step in seconds          */
int paa_plan_beat_execute(paa_plan_t *plan, const double *d_st, double window_size, double *d_beat);
/* the same for ONE short-term matrix in host memory (the reference's own signature, MidTermFeatures.py:18): feats is
 * [n_rows][n_frames] feature-major (n_rows >= 19: rows 0..18 are read, :30-31); bpm_ratio receives (bpm, confidence) */
int paa_beat_extraction_f64(const double *feats, int n_rows, int64_t n_frames, double window_size, double *bpm_ratio);
/* name of the feature kernel the plan dispatches ("st_fast_800", "st_generic", ...)          */
const char *paa_plan_kernel_name(const paa_plan_t *plan);

/* delta rows re-formed on the device (ShortTermFeatures.py:668-680): d_base holds the [34][T_c] base-feature slabs of
 * n_clips clips back to back (frames[c] = T_c, HOST array), d_out receives their [68][T_c] slabs back to back -- rows
 * 34..67 are the differences of consecutive columns, column 0 zeros, bit-identical to what a 68-row plan stores.  A sharded
 * job gathers the 34 base rows over xGMI and the root completes the matrices (half the bytes on the links).  Asynchronous:
 * queued behind the gathers on the communication stream when a communicator exists, else on the library stream          */
int paa_dev_expand_deltas(const double *d_base, const int64_t *frames, int64_t n_clips, double *d_out);

/* ---- self-similarity matrix / music thumbnailing ----------------------------- */
/* audioSegmentation.self_similarity_matrix (audioSegmentation.py:40-55): rows standardised like scikit-learn's
 * StandardScaler, sim[i][j] = 1 - cosine distance of columns i, j (SciPy pdist semantics: clipped cosine, exact 1
 * on the diagonal, NaN for zero vectors).  feats is [n_dims][n_vec] row-major -- the layout feature_extraction
 * returns; sim is [n_vec][n_vec].  Host buffers:                                                */
int paa_self_similarity_f64(const double *feats, int n_dims, int64_t n_vec, double *sim);
/* the same on device buffers (asynchronous on the library stream); ld = row pitch of d_feats in doubles, so the
 * output of paa_plan_execute for a one-clip plan can be passed as is                            */
int paa_dev_self_similarity(const double *d_feats, int n_dims, int64_t n_vec, int64_t ld, double *d_sim);
/* matrix part of audioSegmentation.music_thumbnailing (:1141-1165): moving sum of m_filter cells along the
 * diagonals (convolve2d with eye(m_filter), 'valid'), cells with |i-j| < band, i > j or outside
 * [int(limit_1 R), int(limit_2 R)) set to the global minimum, then the arg-max (first maximum in row-major
 * order).  R = paa_thumbnail_rows(n_vec, m_filter) = n_vec - m_filter + 1; filt is [R][R]; pos2 (HOST) receives
 * (row, column).  n_vec < m_filter is rejected with PAA_ERR_ARG.                                */
int64_t paa_thumbnail_rows(int64_t n_vec, int m_filter);
int paa_thumbnail_f64(const double *feats, int n_dims, int64_t n_vec, int m_filter, double band, double limit_1,
                      double limit_2, double *filt, int64_t *pos2);
/* device buffers in and out; synchronises the library stream before returning pos2.  d_sim must be symmetric bit for
 * bit, as paa_dev_self_similarity writes it: the diagonal sums are formed for j >= i only (the cells below the
 * diagonal are masked, and the minimum over one triangle is the minimum over the matrix)       */
int paa_dev_thumbnail_filter(const double *d_sim, int64_t n_vec, int m_filter, double band, double limit_1,
                             double limit_2, double *d_filt, int64_t *pos2);

/* ---- silence_removal's per-frame SVM loop (audioSegmentation.py:744-748) ------------------------------
 * P(class index 1) of a TRAINED binary probabilistic scikit-learn SVC for every column of feats [n_dims][n_frames]
 * (host, feature-major like the short-term matrix): (x - mean) / scale (:746), decision value from the support vectors
 * (support_vectors [n_sv][n_dims], dual_coef [n_sv], intercept; gamma > 0: RBF kernel, gamma <= 0: linear), Platt
 * sigmoid with (prob_a, prob_b) and libsvm's two-class multiclass_probability -- svm.predict_proba(..)[0][1] per frame.
 * Training (:739) stays with scikit-learn.                                                                  */
int paa_svm_binary_proba_f64(const double *feats, int n_dims, int64_t n_frames, const double *mean, const double *scale,
                             const double *support_vectors, const double *dual_coef, int n_sv, double intercept,
                             double gamma, double prob_a, double prob_b, double *prob1);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI --------------------------------------- */
#define PAA_COMM_ID_BYTES 128
int paa_comm_unique_id(void *id_out /* PAA_COMM_ID_BYTES, rank 0 only */);
/* one process per GPU: PAA_ERR_COMM when another live rank of the same job (same id) already uses the same physical
 * device on this node (ncclCommInitRank would hang); multi-node jobs and one-visible-device-per-rank launchers are fine */
int paa_comm_init(int world_size, int rank, const void *id);
int paa_comm_destroy(void);
/* gather variable-sized double blocks to rank `root`: counts[world] doubles per rank (host);
 * d_recv (root only) receives them back to back in rank order.  Asynchronous on the stream. */
int paa_comm_gather_f64(const double *d_send, const int64_t *counts, int root, double *d_recv);
/* the same with explicit placement: rank r's block lands at d_recv + displs[r] (in doubles) on the root, so that a job
 * cut into chunks can gather chunk k while chunk k+1 is being computed and still end with one rank-major buffer   */
int paa_comm_gatherv_f64(const double *d_send, const int64_t *counts, const int64_t *displs, int root, double *d_recv);
int paa_comm_barrier(void);

/* ---- introspection for tests (host tables built by the reference's rules) ----------------- */
/* dense mel bank [40][num_fft], dct [13][40]; chroma gather list: returns the number of
 * entries, fills src/weight/slot (capacity entries each) in ascending slot order.            */
int paa_debug_mel_bank(double fs, int num_fft, double *out_dense);
int paa_debug_dct(double *out_13x40);
int paa_debug_chroma(double fs, int num_fft, int capacity, int32_t *src, double *weight, int32_t *slot);
/* per-phase cycle totals of the fast kernel (diagnostic builds with -DPAA_F800_TIMING; zeros otherwise) */
int paa_debug_wave_trace(uint64_t *out, int max_waves);
int paa_debug_lane_peak(void);     /* most host-buffer calls in flight at once since the last query */
int paa_debug_phase_cycles(uint64_t *out16);
/* radix plan chosen for a window: returns number of passes, fills radices (capacity 32)      */
int paa_debug_fft_plan(int window, int32_t *radices, int32_t *fft_len);
/* file name (inside PAA_COMM_MARKER_DIR / TMPDIR) of the one-process-per-GPU marker that `rank` of the job `unique_id`
 * drops for the selected device before RCCL is called (comm_rccl.hpp); needs a device                                */
int paa_debug_comm_marker_name(const void *unique_id, int rank, char *out, int capacity);
/* three-pass register-FFT kernels (csrc/kernels_tri.hpp), host only: shape8 = {R1, R2, R3, packed | group pitch of the second exchange << 8, plane row pitch P, waves per
 * workgroup | H1 << 8 | H2 << 16 (lanes that share a prime butterfly of pass 1 / 2), pass-3 lane jobs, LDS bytes}, offsets6 = SEVEN
 * byte offsets {tw2, p3, g_tw1, g_post, table_bytes, total_bytes, split tables} into the
 * table blob (spectrogram mode: no mel / chroma lists), which is copied to `blob` when that is not NULL.  Returns the blob
 * size, 0 when the window goes to another kernel                                                                     */
int paa_debug_tri_plan(int window, double fs, int32_t *shape8, int32_t *offsets6, unsigned char *blob, int capacity);
/* workgroup-per-frame kernels (csrc/kernels_wg.hpp), host only: info32[48] = {complex points, bins, passes, r0 (0: whole transform in one
 * workgroup's LDS, else r0 sub-transforms whose first pass runs from the samples), elements per (sub-)transform, elements between pad
 * slots, threads, LDS bytes, permutation in LDS, feature kernel stages the row, its LDS bytes, then (radix, span, twiddle stride) per
 * pass}; perm[k] = padded LDS position of output k of the (sub-)transform.  Returns 1, 0 when the window goes to another path        */
int paa_debug_wg_plan(int window, int32_t *info32, uint16_t *perm, int perm_capacity);
/* Real-input split of the 12 x 3675- / 6 x 3675-sample windows (csrc/kernels_wgs.hpp: 44 100 and 22 050 samples -- the 1 s window
 * audioSegmentation.py:1134-1138 passes to feature_extraction at 44.1 / 22.05 kHz), host only: info16 = {r0, points per sub-transform Q,
 * R1, R2, R3 (three register passes), row pitch of the exchange buffer, threads per workgroup, LDS bytes, task types per frame, bins
 * of a spectrum row the feature kernel keeps in LDS (the mel filters' range), its LDS bytes, natural blocks of 64 r0 bins per row (the
 * roll-off is located block by block), threads of the feature kernel}; bin_of[W / 2] (may
 * be null) = the bin that element idx of a UNIT-MAJOR spectrum row holds (the transform kernel stores |X| unit after unit: sub-transform
 * q = 1 .. r0/2 - 1 delivers the bins q + r0 kappa and their mirrors, the last one the bins (r0/2) j).  Returns 1, 0 when the window goes
 * to another kernel                                                                                                                  */
int paa_debug_wgs_plan(int window, int32_t *info16, int32_t *bin_of, int capacity);
/* Host tables of the fused three-pass kernel of the 1 s windows (csrc/kernels_wgr.hpp: 16 000 / 8 000 samples -- the windows
 * audioSegmentation.py:1134-1138 passes to feature_extraction), host only.  mel_job[512][4] = per thread {first bin, index of its weight
 * in the mel table, stride, number of bins}: the thread's share of ONE mel filter (ShortTermFeatures.py:236-254; its bins are first
 * bin + j stride); mel_fil[40][2] = per filter {first thread, threads}; ch_n[12], ch_src[12][64], ch_w[12][64] = the chroma gather lists
 * (:277-321), one entry per lane.  Returns the shape id (1: 20 x 20 x 20, 2: 10 x 20 x 20), 0 when the window goes to another kernel,
 * -1 when a table cannot be held (the plan then keeps csrc/kernels_wg.hpp)                                                         */
int paa_debug_wgr_tables(double fs, int window, int32_t *mel_job, int32_t *mel_fil, int32_t *ch_n, int32_t *ch_src, double *ch_w);
/* ... and its runs of consecutive frames (one workgroup walks runs b, b + grid, ...): per-clip frame counts -> (clip, t0, cnt)
 * triples; *n_runs = their number (runs3 may be NULL to query it)                                                                */
int paa_debug_wgr_runs(const int64_t *frames, int64_t n_clips, int num_cu, int32_t *runs3, int64_t capacity, int64_t *n_runs);
/* Host side of the Bluestein kernel (kernels_blu.hpp: windows whose FFT length has a prime factor above 13; replaces
 * scipy.fftpack.fft at ShortTermFeatures.py:617 for them): info8 = {log2 M, R0, R1, R2, waves per workgroup, LDS bytes,
 * table_bytes, total_bytes}, offsets3 = byte offsets of {conj chirp [W], FFT(b) / M in pass order [M], pass twiddles} in the
 * blob.  Returns the blob size, 0 when the window goes to another kernel (blob may be NULL to query the size). */
int paa_debug_blu_plan(int window, double fs, int32_t *info8, int32_t *offsets3, unsigned char *blob, int capacity);
/* the 64 lane jobs {start, n, woff, ctl} the three-pass kernels cut the sums of n_owners <= 64 owners (40 mel filters / 12 pitch
 * classes) into: owner k has cnt[k] consecutive entries from first[k] (weights from wfirst[k]); a job's ctl = position of the piece
 * in its owner's run of lanes | (lanes k < n_owners: the lane that ends up with owner k's total) << 8 (csrc/kernels_tri.hpp)   */
int paa_debug_lane_jobs(const int32_t *first, const int32_t *wfirst, const int32_t *cnt, int n_owners, int32_t *jobs256);
/* mixed-radix kernel (csrc/kernels_mix.hpp): radix schedule of its in-place DIF transform and the position that holds
 * Z[k] afterwards (perm: fft_len entries); returns the number of passes, 0 when the window goes to another kernel   */
int paa_debug_mix_plan(int window, int32_t *radices, int32_t *fft_len, uint16_t *perm, int perm_capacity,
                       int32_t *waves, int32_t *tw_global);
/* the run-length choice of paa_plan_create for clips of frames[c] frames (host only): runs are multiples of `quantum`
 * frames within [min_run, max_run], cost `halo` extra frames each, a workgroup takes wg_runs of them and num_cu
 * workgroups run at a time.  Returns the cap, the number of runs and the longest run          */
int paa_debug_run_plan(const int64_t *frames, int64_t n_clips, int quantum, int min_run, int max_run, int halo, int wg_runs,
                       int num_cu, int32_t *run_cap, int64_t *n_runs, int32_t *longest);
/* the same for kernels whose runs after a clip's first are `shrink` frames shorter (halo inside the first iteration) */
int paa_debug_run_plan_shrink(const int64_t *frames, int64_t n_clips, int quantum, int min_run, int max_run, int shrink,
                              int wg_runs, int num_cu, int32_t *run_cap, int64_t *n_runs, int32_t *longest);
/* the run lengths a plan that fills LESS than one round of a one-workgroup-per-CU kernel gets instead of equal runs (the hot
 * kernel: num_cu x wg_runs runs whose iteration counts differ by at most one; csrc/lib_plan.hpp: balanced_runs).  Writes them clip
 * after clip to lens[capacity]; returns their number, 0 when the equal runs of `run_cap` stay                              */
int64_t paa_debug_balanced_runs(const int64_t *frames, int64_t n_clips, int run_cap, int quantum, int shrink, int wg_runs,
                                int num_cu, int min_run, int32_t *lens, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* PAA_HIP_H */

// Three-pass register-FFT feature kernels for the LARGE windows the reference uses by default at 44.1 / 48 kHz
// (50 ms = 2205 / 2400 samples, audioTrainTest.py:28-29, audioAnalysis.py:46; 40 ms = 1764 / 1920, audioAnalysis.py:71,80):
// any step, int16 / interleaved stereo int16 / float64 samples, features / spectrogram / chromagram.
//
// One wave = one run of consecutive frames of one clip, ONE frame per iteration with all 64 lanes on it.  The frame never
// sits in LDS as a complex buffer: the transform of length N = R1 R2 R3 runs in registers, and the data moves between the
// passes through ONE plane of doubles (real parts, then imaginary parts) that is the frame's own spectrum slot:
//
//   load    : lane j (< L1 = R2 R3, one or two "jobs" per lane) fetches z[j + L1 r], r < R1, straight from HBM / L2 (per load
//             instruction the lanes read one contiguous span) and removes the clip mean.  Even windows are packed two real
//             samples per complex point (N = W / 2, PairLoad); odd windows run a REAL-input transform of length N = W whose
//             first pass only produces the (R1 + 1) / 2 non-redundant outputs -- half the work of the complex transform the
//             mixed-radix kernel (kernels_mix.hpp) spends on them
//   time    : the same registers give energy, the ten entropy-block energies and the sign changes (block membership is
//             static per register row; the sample before a lane's first one is in the lane below: DPP wave_shr:1)
//   pass 1  : radix-R1 codelet per job, outputs times W_N^(j q1) (table in global memory: L1 / L2 hits)
//   exchange: element (j, q1) at plane[q1 P + j]
//   pass 2  : lane (q1, b): radix-R2 over a of B[R3 a + b][q1], outputs times W_L1^(b q2)
//   exchange: element (q1, b, q2) at plane[q1 P + q2 R3 + b]
//   pass 3  : radix-R3 over b for job (q1, q2): Z[q1 + R1 (q2 + R2 k3)].  Packed windows: a lane takes a PAIR of jobs whose
//             outputs are Z[k] and Z[N - k], so the real-FFT recombination and |X| happen in registers; real-input windows:
//             |Z[k]| goes to bin k or N - k.  The magnitudes overwrite the plane: it becomes the frame's spectrum.
//   features: kernels_mix.hpp's spectral stage on the two spectrum slots (current / previous frame alternate)
//   store   : lane = feature row; a row's values wait in eight registers until a 64-byte aligned chunk is complete, which
//             is stored whole and non-temporally (kernels_fast.hpp: store_chunk_pieces)
//
// Per-wave LDS: two spectrum slots + 0.7 KB (window 2400: 19.9 KB against the 34 KB of the in-place transform), so SEVEN
// or EIGHT waves share a CU -- two per SIMD -- and every kernel instance stays below 256 registers.
// The codelets keep the property that R equal inputs give exact zeros in the non-DC outputs (device_common.hpp: dft5), so a
// digitally silent frame has the exact spectrum [2 |c|, 0, 0, ...] the reference's pocketfft produces.
//
// Replaces the while loop at ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321) and the loops of spectrogram
// (:415-422) / chromagram (:349-359) for these windows.
#pragma once
#include <algorithm>
#include <type_traits>
#include <vector>

#include "kernels_mix.hpp"
#include "prime_tables.hpp"        // PrimeTab

namespace paa {
namespace tri {
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
using f800::g_phase_cycles;
using f800::g_wave_trace;
#endif

// ---- shapes ---------------------------------------------------------------------------------------------------
// number of pass-3 lane jobs of a packed shape: unordered pairs {job, partner job}, job (q1, q2) <-> Z[q1 + R1 q2 + ...]
constexpr int pair_count(int R1, int R2, int R3) {
    const int N = R1 * R2 * R3;
    int pairs = 0;
    for (int q1 = 0; q1 < R1; ++q1)
        for (int q2 = 0; q2 < R2; ++q2) {
            const int k = q1 + R1 * q2, m = (N - k) % N;
            const int p1 = m % R1, p2 = (m / R1) % R2;
            if (p1 > q1 || (p1 == q1 && p2 >= q2)) ++pairs;         // count each unordered pair once
        }
    return pairs;
}

template <int R1_, int R2_, int R3_, bool PACKED_, int P_, int NW_, int R3P_ = R3_, int H1_ = 1, int H2_ = 1, int NWR_ = NW_>
struct Shape {
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_, P = P_, NW = NW_;
    // waves per workgroup of the spectrogram / chromagram instances (no feature stage: fewer registers, so a third wave per SIMD fits
    // where the feature instance needs more than 168)
    static constexpr int NWR = NWR_;
    // prime first / second radix of a real-input shape: the O(R^2) butterfly of a pass-1 (pass-2) job is shared by H1 (H2) lanes,
    // each forming a subset of the outputs (SplitSel below) -- the lanes a prime shape leaves idle take a share of the FMAs
    static constexpr int H1 = H1_, H2 = H2_;
    // second exchange: element (q1, b, q2) sits at plane[q1 P + q2 R3P + b]; R3P = R3 + 1 for the radix-8 last pass (the jobs of a
    // pass-3 read group then hit 32 different bank pairs: q2 R3 = 8 q2 repeats every four jobs)
    static constexpr int R3P = R3P_;
    static constexpr bool PACKED = PACKED_;
    static constexpr int N = R1 * R2 * R3;                  // transform length
    static constexpr int W = PACKED ? 2 * N : N, NF = W / 2;
    static constexpr int L1 = R2 * R3;                      // pass-1 jobs
    static constexpr int NQ1 = PACKED ? R1 : (R1 + 1) / 2;  // pass-1 outputs that are needed
    static constexpr int NJ = (L1 + 63) / 64;               // pass-1 jobs per lane
    static constexpr int J2 = NQ1 * R3;                     // pass-2 jobs (one per lane)
    // 16 x 16 x 4 (2048 samples): 127 pairs + the two self-paired jobs 0 and R1 R2 / 2 = 129 lane jobs -- a third pass-3 round for one
    // job.  Job 0 needs three of its lane's four result slots (bins N3 & N - N3, 2 N3, and bin 0, which is two additions of its own),
    // job R1 R2 / 2 two (its pairs k3 <-> 3 - k3): the two share lane 0 (FOLD) and the pass has 128 jobs = two rounds
    static constexpr bool FOLD = PACKED && R3 == 4 && (R1 * R2) % 2 == 0 && pair_count(R1, R2, R3) % 64 == 1;
    static constexpr int NJOB3 = PACKED ? pair_count(R1, R2, R3) - (FOLD ? 1 : 0) : NQ1 * R2;
    static constexpr int NR3 = (NJOB3 + 63) / 64;           // pass-3 rounds
    static constexpr int PLANE = NQ1 * P;
    static constexpr int C0 = (NF + 63) / 64;
    static constexpr int C = (C0 % 2) ? C0 : C0 + 1;        // bins per lane in the feature stage (odd: conflict-free at stride C)
    // doubles per spectrum slot: the exchange plane, and 64 lanes x C bins for the feature stage -- the bins past NF are kept
    // at zero (re-zeroed after every transform where the plane reaches into them), so the sweeps need no bounds masks
    // (+ one double at index NF, where the last pass parks the results that no bin takes)
    static constexpr int NFD = NF + 1;
    static constexpr int PLANED = PLANE + ((H1_ > 1 || H2_ > 1) ? 1 : 0);        // split passes: + a dummy double for dropped outputs
    static constexpr int SLOT00 = (PLANED > NFD ? PLANED : NFD) > 64 * C ? (PLANED > NFD ? PLANED : NFD) : 64 * C;
    // the time-domain partials are summed through an LDS transpose of 11 x 65 doubles inside the frame's spectrum slot (free at
    // that point); a slot that is only a little smaller is padded to that size (both slots: 2 x the padding) when that is cheaper
    // than a scratch of its own -- 1024-sample windows: 12.3 instead of 15.7 KB per wave = ten waves per CU instead of eight
    static constexpr int SLOT0 = (SLOT00 < 11 * 65 && 2 * (11 * 65 - SLOT00) < 11 * 65) ? 11 * 65 : SLOT00;
    static constexpr int SLOT = (SLOT0 + 1) & ~1;
    static constexpr int LT = W / 10, LB = NF / 10;         // entropy blocks (samples / bins)
    static constexpr int SPL = PACKED ? 2 : 1;              // samples per loaded element
    // ... in a scratch of its own for the small windows
    static constexpr int TSCR = (SLOT >= 11 * 65) ? 0 : 11 * 65;
    static constexpr int WAVE_DOUBLES = 2 * SLOT + 48 + 40 + 12 + TSCR;   // two slots, fv[48], msp[40], bnd[12], scratch
    static constexpr int WAVE_DOUBLES_ROWS = SLOT + 16;                   // spectrogram / chromagram instances: no previous spectrum,
                                                                          // no feature staging -- one slot, up to sixteen waves per CU
    static_assert(P >= L1 && P >= (R2 - 1) * R3P + R3 && R3P >= R3, "plane rows hold L1 elements (first exchange) and R2 groups of R3 (second)");
    static_assert(J2 * H2 <= 64, "one pass-2 job (part) per lane");
    static_assert(H1 == 1 || (!PACKED && L1 * H1 <= 64), "split first pass: real input, all parts of all jobs in one wave");
    static_assert(H2 == 1 || !PACKED, "split second pass: real-input shapes");
    static_assert(!PACKED || NJ == 1, "packed shapes: one pass-1 job per lane");
    static_assert(PACKED || (R1 % 2 == 1), "real-input shapes: odd first radix");
    // packed shapes: a pass-3 table entry is PE x 8 ushorts = {plane offset of job A, of job B, (bin of X[k], bin of X[N - k]) x R3}
    static constexpr int PE = PACKED ? (2 + 2 * R3 + 7) / 8 : 1;
    static_assert(PACKED || (L1 < 64 ? L1 : 64) * SPL <= LT, "real-input shapes: a register row may contain at most one entropy-block boundary");
    static_assert(C <= LB, "a lane's bins may contain at most one entropy-block boundary");
    static_assert(R3 > 1 || !PACKED, "two-pass shapes (R3 = 1): real input only (Z[k] and Z[N - k] would sit in different lanes)");
    static_assert(R3 <= 8, "pass 3 has codelets up to radix 8");
    static_assert(SLOT * 8 < 65536, "pass-3 table entries are 16-bit byte offsets into the slot");
};

// shared (per workgroup) LDS tables + the global tables behind them in the same device blob
struct TriLayout {
    int off_tw2;                    // double2 [R2][R3]: W_L1^(b q2)
    int off_p3;                     // PE x 8 ushorts [64 NR3], BYTE offsets into the slot.  Packed: plane elements of job A, of job B, then per
                                    // output k3 < R3 where |X[k]| and |X[N - k]| go (8 NF = "nowhere": a parking double).  Real input, three
                                    // passes: plane elements of the job, then where its R3 (<= 5) magnitudes go.  Two passes (R3 = 1):
                                    // ushort [R2][64]: where lane q1's magnitude q2 goes
    int off_mello, off_melcnt, off_meloff, off_melw, off_dct, off_chstart, off_chsrc, off_chw;
    int off_split;                  // split prime passes: int [H1][8] then int [H2][16]: per part and output slot the true output index
                                    // (bits 0-7), conjugate (bit 8), enabled (bit 9: the first part that produces an index writes it)
    int off_sync;                   // pacing of the waves of a SIMD: SIMD id [16], progress in half frames [16] (ints)
    int waves;                      // waves per workgroup of this launch: the shape's maximum (Shape::NW / NWR, which also sets the
                                    // register budget) or fewer when the table blob of this (fs, window) leaves less LDS
    int table_bytes;                // LDS part, multiple of 16
    int off_g_tw1;                  // global part: double2 [NQ1][L1]: W_N^(j q1)
    int off_g_post;                 // packed: double2 [64 NR3][R3]: W_W^(kA + N3 k3)
    int off_g_meljob, off_g_chjob;  // int4 [64] each: the lane jobs of the mel sums / the chroma gather (LaneJob below)
    int total_bytes;
    double f0, rf0, r_half_fs, f0sq;                    // fs / W, its reciprocal, 2 / fs, f0^2 (host-computed: scalar registers)
};

// ---- codelets: run() transforms v[] in place, X[q] ends at v[pos(q)] ------------------------------------------------
template <int R> struct Cd;
template <> struct Cd<2> {
    static __device__ __forceinline__ void run(double2 *v) { dft2(v); }
    static constexpr int pos(int q) { return q; }
};
template <> struct Cd<3> {
    static __device__ __forceinline__ void run(double2 *v) { dft3(v); }
    static constexpr int pos(int q) { return q; }
};
template <> struct Cd<5> {
    static __device__ __forceinline__ void run(double2 *v) { dft5(v); }
    static constexpr int pos(int q) { return q; }
};
template <> struct Cd<4> {
    static __device__ __forceinline__ void run(double2 *v) { ct::dft4r(v[0], v[1], v[2], v[3]); }
    static constexpr int pos(int q) { return q; }
};
template <> struct Cd<8> {
    static __device__ __forceinline__ void run(double2 *v) { ct::Dft<8>::run<1>(v); }
    static constexpr int pos(int q) { return ct::Dft<8>::pos(q); }
};
template <> struct Cd<16> {
    static __device__ __forceinline__ void run(double2 *v) { ct::Dft<16>::run<1>(v); }
    static constexpr int pos(int q) { return ct::Dft<16>::pos(q); }
};
template <> struct Cd<20> {
    static __device__ __forceinline__ void run(double2 *v) { ct::Dft<20>::run<1>(v); }
    static constexpr int pos(int q) { return ct::Dft<20>::pos(q); }
};
// 21 = 3 x 7, coprime: prime-factor transform, no twiddles.  Element (n1, n2) sits at v[(7 n1 + 3 n2) % 21] (= natural order);
// X[q] ends at v[(7 (q % 3) + 3 (q % 7)) % 21].
template <> struct Cd<21> {
    static __device__ __forceinline__ void run(double2 *v) {
#pragma unroll
        for (int n2 = 0; n2 < 7; ++n2) {
            double2 t[3] = {v[(3 * n2) % 21], v[(7 + 3 * n2) % 21], v[(14 + 3 * n2) % 21]};
            dft3(t);
            v[(3 * n2) % 21] = t[0]; v[(7 + 3 * n2) % 21] = t[1]; v[(14 + 3 * n2) % 21] = t[2];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1) {
            double2 t[7];
#pragma unroll
            for (int n2 = 0; n2 < 7; ++n2) t[n2] = v[(7 * k1 + 3 * n2) % 21];
            mix::dft_prime<7>(t);
#pragma unroll
            for (int n2 = 0; n2 < 7; ++n2) v[(7 * k1 + 3 * n2) % 21] = t[n2];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static constexpr int pos(int q) { return (7 * (q % 3) + 3 * (q % 7)) % 21; }
};

// ---- odd primes: O(R^2) butterflies from the sums / differences of the pairs (x_j, x_{R-j}), in the pivot form of
// mix::dft_prime (R equal inputs give exact zeros in the non-DC outputs); cos / sin from the half tables of prime_tables.hpp
template <int R> struct PT {
    static constexpr int H = (R - 1) / 2;
    static constexpr double c(int m) { return (m % R == 0) ? 1.0 : reg::PrimeTab<R>::c[((m % R) <= H ? (m % R) : R - (m % R)) - 1]; }
    static constexpr double s(int m) {
        return (m % R == 0) ? 0.0 : ((m % R) <= H ? reg::PrimeTab<R>::s[(m % R) - 1] : -reg::PrimeTab<R>::s[R - (m % R) - 1]);
    }
};
template <int R>
__device__ __forceinline__ void cdft_prime(double2 *v) {          // complex, in place, natural order
    constexpr int H = (R - 1) / 2;
    double2 sm[H], df[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) { sm[j - 1] = cadd(v[j], v[R - j]); df[j - 1] = csub(v[j], v[R - j]); }
    const double2 x0 = v[0];
    double2 tot = x0;
#pragma unroll
    for (int j = 0; j < H; ++j) tot = cadd(tot, sm[j]);
    const double2 base = make_double2(fma(-0.5, sm[H - 1].x, x0.x), fma(-0.5, sm[H - 1].y, x0.y));
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) sm[j] = csub(sm[j], sm[H - 1]);      // rel_j = s_j - s_H
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        double ar = base.x, ai = base.y, br = 0.0, bi = 0.0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const double c = PT<R>::c(j * q), sn = PT<R>::s(j * q);
            if (j < H) { ar = fma(c, sm[j - 1].x, ar); ai = fma(c, sm[j - 1].y, ai); }
            br = fma(sn, df[j - 1].x, br);
            bi = fma(sn, df[j - 1].y, bi);
        }
        v[q] = make_double2(ar + bi, ai - br);             // X[q] = A - i B, X[R - q] = A + i B
        v[R - q] = make_double2(ar - bi, ai + br);
        __builtin_amdgcn_sched_barrier(0);
    }
    v[0] = tot;
}
template <int R>
__device__ __forceinline__ void rdft_prime(const double *x, double2 *a) {      // real input -> a[q] = X[q], q = 0 .. H
    constexpr int H = (R - 1) / 2;
    double sm[H], df[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) { sm[j - 1] = x[j] + x[R - j]; df[j - 1] = x[j] - x[R - j]; }
    double tot = x[0];
#pragma unroll
    for (int j = 0; j < H; ++j) tot += sm[j];
    const double base = fma(-0.5, sm[H - 1], x[0]);
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) sm[j] -= sm[H - 1];
    a[0] = make_double2(tot, 0.0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        double ar = base, br = 0.0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            if (j < H) ar = fma(PT<R>::c(j * q), sm[j - 1], ar);
            br = fma(PT<R>::s(j * q), df[j - 1], br);
        }
        a[q] = make_double2(ar, -br);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// ---- prime butterflies shared by H lanes ("parts"): every part runs the SAME code -- the outputs q of SplitSel<R, H>::q with
// the same compile-time cos / sin constants -- on its inputs taken in the order n -> (n ginv_h) mod R, which makes output q of
// part h the true output (g_h q) mod R (X'[q] = sum_n x[n ginv] W^(n q) = sum_m x[m] W^(m g q)); the multipliers g_h are chosen so
// that the parts' outputs cover every index once or twice (second copies are dropped by the host tables).  Real input:
// X[R - t] = conj X[t], so a part's output q stands for index min(t, R - t), conjugated when t > R / 2.  Same pivot form as
// rdft_prime / cdft_prime: R equal inputs give exact zeros in the non-DC outputs.
template <int R, int H> struct SplitSel;
template <> struct SplitSel<29, 3> {            // {1,2,3,6,9} x {1,5,11} covers +-1 .. +-14 (5 of the 14 outputs / pairs per part)
    static constexpr int NQ = 5;
    static constexpr int q(int i) { constexpr int t[5] = {1, 2, 3, 6, 9}; return t[i]; }
    static constexpr int g(int h) { constexpr int t[3] = {1, 5, 11}; return t[h]; }
};
template <> struct SplitSel<19, 3> {            // {1,2,4} x {1,7,8} covers +-1 .. +-9
    static constexpr int NQ = 3;
    static constexpr int q(int i) { constexpr int t[3] = {1, 2, 4}; return t[i]; }
    static constexpr int g(int h) { constexpr int t[3] = {1, 7, 8}; return t[h]; }
};
template <int R, int H> struct SplitNQ { static constexpr int value = SplitSel<R, H>::NQ; };
template <int R> struct SplitNQ<R, 1> { static constexpr int value = 0; };
constexpr int mod_inverse(int g, int R) {
    for (int x = 1; x < R; ++x)
        if ((g * x) % R == 1) return x;
    return 1;
}
// real input (in the part's order) -> a[0] = X[0] (real), a[1 + i] = X'[q_i]
template <int R, typename SEL>
__device__ __forceinline__ void rdft_prime_sel(const double *x, double2 *a) {
    constexpr int H = (R - 1) / 2;
    double sm[H], df[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) { sm[j - 1] = x[j] + x[R - j]; df[j - 1] = x[j] - x[R - j]; }
    double tot = x[0];
#pragma unroll
    for (int j = 0; j < H; ++j) tot += sm[j];
    const double base = fma(-0.5, sm[H - 1], x[0]);
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) sm[j] -= sm[H - 1];
    a[0] = make_double2(tot, 0.0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < SEL::NQ; ++i) {
        constexpr int dummy = 0; (void)dummy;
        const int q = SEL::q(i);
        double ar = base, br = 0.0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            if (j < H) ar = fma(PT<R>::c(j * q), sm[j - 1], ar);
            br = fma(PT<R>::s(j * q), df[j - 1], br);
        }
        a[1 + i] = make_double2(ar, -br);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// one real plane of a complex column (in the part's order): tot = sum, A[i] = sum_n x_n cos(2 pi n q_i / R), B[i] = sum_n x_n
// sin(2 pi n q_i / R) -- the complex butterfly is this, once on the real and once on the imaginary parts (X[q] = (A_r + B_i) +
// i (A_i - B_r), X[R - q] = (A_r - B_i) + i (A_i + B_r)): the same operations in the same order as cdft_prime_sel, with half of its
// registers live (the exchange delivers the planes one after the other anyway)
template <int R, typename SEL>
__device__ __forceinline__ void prime_sel_plane(const double *x, double &tot_out, double *A, double *B) {
    constexpr int H = (R - 1) / 2;
    double sm[H], df[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) { sm[j - 1] = x[j] + x[R - j]; df[j - 1] = x[j] - x[R - j]; }
    double tot = x[0];
#pragma unroll
    for (int j = 0; j < H; ++j) tot += sm[j];
    const double base = fma(-0.5, sm[H - 1], x[0]);
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) sm[j] -= sm[H - 1];
    tot_out = tot;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < SEL::NQ; ++i) {
        const int q = SEL::q(i);
        double ar = base, br = 0.0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            if (j < H) ar = fma(PT<R>::c(j * q), sm[j - 1], ar);
            br = fma(PT<R>::s(j * q), df[j - 1], br);
        }
        A[i] = ar;
        B[i] = br;
        __builtin_amdgcn_sched_barrier(0);
    }
}
// complex input (in the part's order) -> o[0] = X[0], o[1 + 2 i] = X'[q_i], o[2 + 2 i] = X'[R - q_i]
template <int R, typename SEL>
__device__ __forceinline__ void cdft_prime_sel(const double2 *v, double2 *o) {
    constexpr int H = (R - 1) / 2;
    double2 sm[H], df[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) { sm[j - 1] = cadd(v[j], v[R - j]); df[j - 1] = csub(v[j], v[R - j]); }
    const double2 x0 = v[0];
    double2 tot = x0;
#pragma unroll
    for (int j = 0; j < H; ++j) tot = cadd(tot, sm[j]);
    const double2 base = make_double2(fma(-0.5, sm[H - 1].x, x0.x), fma(-0.5, sm[H - 1].y, x0.y));
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) sm[j] = csub(sm[j], sm[H - 1]);
    o[0] = tot;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < SEL::NQ; ++i) {
        const int q = SEL::q(i);
        double ar = base.x, ai = base.y, br = 0.0, bi = 0.0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const double c = PT<R>::c(j * q), sn = PT<R>::s(j * q);
            if (j < H) { ar = fma(c, sm[j - 1].x, ar); ai = fma(c, sm[j - 1].y, ai); }
            br = fma(sn, df[j - 1].x, br);
            bi = fma(sn, df[j - 1].y, bi);
        }
        o[1 + 2 * i] = make_double2(ar + bi, ai - br);         // X[q] = A - i B
        o[2 + 2 * i] = make_double2(ar - bi, ai + br);         // X[R - q] = A + i B
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <> struct Cd<19> {
    static __device__ __forceinline__ void run(double2 *v) { cdft_prime<19>(v); }
    static constexpr int pos(int q) { return q; }
};
template <> struct Cd<29> {
    static __device__ __forceinline__ void run(double2 *v) { cdft_prime<29>(v); }
    static constexpr int pos(int q) { return q; }
};
template <> struct Cd<10> {
    static __device__ __forceinline__ void run(double2 *v) { ct::Dft<10>::run<1>(v); }
    static constexpr int pos(int q) { return ct::Dft<10>::pos(q); }
};

// real-input first pass: x[R] real (natural order) -> a[q] = X[q], q = 0 .. (R - 1) / 2 (the rest are conjugates)
template <int R> struct RCd;
// three real inputs: X0 = a + b + c (real), X1 = (a - (b + c) / 2) - i h (b - c); equal inputs give X1 = 0 exactly
__device__ __forceinline__ void rdft3(double a, double b, double c, double &s0, double2 &x1) {
    const double h = 0.86602540378443864676;
    const double t = b + c;
    s0 = a + t;
    x1 = make_double2(fma(-0.5, t, a), -(h * (b - c)));
}
// seven real inputs -> y[0] (real) .. y[3]; the pivot form of mix::dft_prime (equal inputs: exact zeros in y[1..3])
__device__ __forceinline__ void rdft7(const double *s, double2 *y) {
    typedef mix::PrimeTab<7> TB;
    const double sm1 = s[1] + s[6], sm2 = s[2] + s[5], sm3 = s[3] + s[4];
    const double df1 = s[1] - s[6], df2 = s[2] - s[5], df3 = s[3] - s[4];
    const double tot = ((s[0] + sm1) + sm2) + sm3;
    const double base = fma(-0.5, sm3, s[0]);
    const double rel1 = sm1 - sm3, rel2 = sm2 - sm3;
    y[0] = make_double2(tot, 0.0);
#pragma unroll
    for (int q = 1; q <= 3; ++q) {
        double ar = base, br = 0.0;
        ar = fma(TB::c[(1 * q) % 7], rel1, ar);
        ar = fma(TB::c[(2 * q) % 7], rel2, ar);
        br = fma(TB::s[(1 * q) % 7], df1, br);
        br = fma(TB::s[(2 * q) % 7], df2, br);
        br = fma(TB::s[(3 * q) % 7], df3, br);
        y[q] = make_double2(ar, -br);                    // X[q] = A - i B
    }
}
template <> struct RCd<29> {
    static __device__ __forceinline__ void run(const double *x, double2 *a) { rdft_prime<29>(x, a); }
};
template <> struct RCd<19> {
    static __device__ __forceinline__ void run(const double *x, double2 *a) { rdft_prime<19>(x, a); }
};
template <> struct RCd<21> {
    static __device__ __forceinline__ void run(const double *x, double2 *a) {
        double s0[7];
        double2 c1[7];
#pragma unroll
        for (int n2 = 0; n2 < 7; ++n2) rdft3(x[(3 * n2) % 21], x[(7 + 3 * n2) % 21], x[(14 + 3 * n2) % 21], s0[n2], c1[n2]);
        __builtin_amdgcn_sched_barrier(0);
        double2 y0[4];
        rdft7(s0, y0);                                   // k1 = 0: real radix 7
        __builtin_amdgcn_sched_barrier(0);
        mix::dft_prime<7>(c1);                           // k1 = 1: complex radix 7 (k1 = 2 is its conjugate, mirrored)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const int k1 = q % 3, k2 = q % 7;
            if (k1 == 0) a[q] = (k2 <= 3) ? y0[k2] : make_double2(y0[7 - k2].x, -y0[7 - k2].y);
            else if (k1 == 1) a[q] = c1[k2];
            else a[q] = make_double2(c1[(7 - k2) % 7].x, -c1[(7 - k2) % 7].y);
        }
    }
};

// lane l receives the value of lane l - 1 (lane 0: `first`)
__device__ __forceinline__ int wave_shr1(int v, int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false);       // wave_shr:1
}
template <typename T> __device__ __forceinline__ int load_int(const T *p);
template <> __device__ __forceinline__ int load_int<int16_t>(const int16_t *p) { return (int)(*p); }
template <> __device__ __forceinline__ int load_int<stereo16>(const stereo16 *p) { return stereo_word_sum(*reinterpret_cast<const int *>(p)); }
template <> __device__ __forceinline__ int load_int<double>(const double *) { return 0; }

// a feature row's pending values: h[i] = the frame at position i of the row's current 64-byte chunk
struct RowChunk {
    double h[8];
};
// frame t of the row starting at `row` (frames [lo, hi) are this wave's): insert, store the chunk when it is complete or
// the run ends
__device__ __forceinline__ void row_put(RowChunk &rc, double *row, int t, int lo, int hi, double v) {
    const int a = (int)(((reinterpret_cast<uintptr_t>(row) >> 3) + (unsigned)t) & 7);
#pragma unroll
    for (int i = 0; i < 8; ++i) rc.h[i] = (a == i) ? v : rc.h[i];
    const bool last = (t == hi - 1);
    if (a == 7 || last) {
        const int g = t - a;                               // first frame of the chunk
        f800::store_chunk_pieces(row + g, rc.h, lo - g, (a == 7) ? 8 : a + 1);
    }
}


// ---- mel sums and chroma gather on all 64 lanes (round 5).  Until then lane m < 40 walked mel filter m (the widest filter set the
// trip count: 8 % of the 1024 kernel) and lane c < 12 its pitch class (NF / 12 entries each: 3-6 %).  Now every OWNER (filter,
// class) is cut into lane jobs of at most `m` consecutive entries on consecutive lanes of ONE 16-lane row (host: lane_jobs);
// a lane sums its piece, a segmented scan inside the row (row_shr DPP, a lane adds its left neighbours of the same owner only)
// leaves the owner's total in its last lane, and the owner's lane (lane m < 40 / lane c < 12) fetches it.  Deterministic order.
struct LaneJob {
    int start;        // mel: first bin; chroma: first entry of the gather list
    int n;            // entries of this piece
    int woff;         // mel: index of the first weight
    int ctl;          // bits 0-3: position of the piece in its owner's run of lanes; bits 8-13: (owner lanes only) lane that ends up with the owner's total
};
// inclusive sum over the lanes [lane - head, lane] (all inside one 16-lane row)
__device__ __forceinline__ double seg_row_scan(double v, int head) {
    double t;
    t = dpp_mov<0x111>(v); v += (head >= 1) ? t : 0.0;
    // after step k a lane holds the sum of min(head, 2^k - 1) + 1 lanes ending at itself: the next step adds the partial sum 2^k lanes to
    // the left when that lane belongs to the same owner (its own partial then covers exactly the missing lanes or stops at the owner's first)
    t = dpp_mov<0x112>(v); v += (head >= 2) ? t : 0.0;
    t = dpp_mov<0x114>(v); v += (head >= 4) ? t : 0.0;
    t = dpp_mov<0x118>(v); v += (head >= 8) ? t : 0.0;
    return v;
}
__device__ __forceinline__ double lane_fetch(double v, int src_lane) {
    const int lo = __shfl(__double2loint(v), src_lane, 64), hi = __shfl(__double2hiint(v), src_lane, 64);
    return __hiloint2double(hi, lo);
}
// mel band energies: returns (in lane m < 40) sum_k X[k] w_m[k]
__device__ __forceinline__ double mel_sums_balanced(const Tabs &tb, const double *cur, const int4 job, int lane) {
    const double *w = tb.mel_w + job.z;
    const double *x = cur + job.x;
    double a0 = 0.0, a1 = 0.0;
    int i = 0;
    for (; i + 4 <= job.y; i += 4) {
        double xb[4], wb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xb[u] = x[i + u]; wb[u] = w[i + u]; }
        a0 = fma(xb[0], wb[0], a0); a1 = fma(xb[1], wb[1], a1);
        a0 = fma(xb[2], wb[2], a0); a1 = fma(xb[3], wb[3], a1);
    }
    for (; i < job.y; ++i) a0 = fma(x[i], w[i], a0);
    const double tot = seg_row_scan(a0 + a1, job.w & 15);
    return lane_fetch(tot, (job.w >> 8) & 63);
}
// chroma numerators: returns (in lane c < 12) sum over the class's entries of X[src]^2 w
__device__ __forceinline__ double chroma_sums_balanced(const Tabs &tb, const double *cur, const int4 job, int lane) {
    const int *src = tb.ch_src + job.x;
    const double *w = tb.ch_w + job.x;
    double a0 = 0.0, a1 = 0.0;
    int i = 0;
    for (; i + 4 <= job.y; i += 4) {
        int sb[4];
        double xb[4], wb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sb[u] = src[i + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xb[u] = cur[sb[u]]; wb[u] = w[i + u]; }
        a0 = fma(xb[0] * xb[0], wb[0], a0); a1 = fma(xb[1] * xb[1], wb[1], a1);
        a0 = fma(xb[2] * xb[2], wb[2], a0); a1 = fma(xb[3] * xb[3], wb[3], a1);
    }
    for (; i < job.y; ++i) { const double xv = cur[src[i]]; a0 = fma(xv * xv, w[i], a0); }
    const double tot = seg_row_scan(a0 + a1, job.w & 15);
    return lane_fetch(tot, (job.w >> 8) & 63);
}
// host: cut K owners with cnt[k] entries starting at first[k] (weights at wfirst[k]) into 64 lane jobs.  The smallest piece
// length m for which every owner's ceil(cnt / m) <= 16 lanes fit, unsplit, into the four 16-lane rows (first fit, longest first).
inline void lane_jobs(int K, const int *first, const int *wfirst, const int *cnt, LaneJob *jobs) {
    int best_m = 0;
    std::vector<int> row_of((size_t)K, 0), pos_of((size_t)K, 0), lanes((size_t)K, 1);
    int max_cnt = 1;
    for (int k = 0; k < K; ++k) max_cnt = std::max(max_cnt, cnt[k]);
    for (int m = 1; m <= max_cnt; ++m) {
        std::vector<int> order((size_t)K);
        bool ok = true;
        for (int k = 0; k < K; ++k) {
            order[(size_t)k] = k;
            lanes[(size_t)k] = std::max(1, (cnt[k] + m - 1) / m);
            ok = ok && lanes[(size_t)k] <= 16;
        }
        if (!ok) continue;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lanes[(size_t)a] > lanes[(size_t)b]; });
        int used[4] = {0, 0, 0, 0};
        for (int k : order) {
            int r = 0;
            while (r < 4 && used[r] + lanes[(size_t)k] > 16) ++r;
            if (r == 4) { ok = false; break; }
            row_of[(size_t)k] = r; pos_of[(size_t)k] = used[r];
            used[r] += lanes[(size_t)k];
        }
        if (ok) { best_m = m; break; }
    }
    for (int l = 0; l < 64; ++l) jobs[l] = LaneJob{0, 0, 0, 0};
    if (!best_m) return;          // (cannot happen for K <= 64: m = max_cnt gives one lane per owner)
    for (int k = 0; k < K; ++k) {
        const int nl = lanes[(size_t)k], l0 = 16 * row_of[(size_t)k] + pos_of[(size_t)k];
        int done = 0;
        for (int i = 0; i < nl; ++i) {
            const int n = (cnt[k] - done + (nl - i) - 1) / (nl - i);          // nearly equal pieces, the longer ones first
            LaneJob &j = jobs[l0 + i];
            j.start = first[k] + done; j.woff = wfirst[k] + done; j.n = n; j.ctl = (j.ctl & ~15) | i;
            done += n;
        }
        jobs[k].ctl = (jobs[k].ctl & 0xff) | ((l0 + nl - 1) << 8);          // owner k is read by lane k
    }
}

// ---- the 34 base features of one frame into fv[0..33] (ShortTermFeatures.py:626-667): all 64 lanes on one spectrum,
// lane i owns bins [C i, C i + C) in registers for both sweeps; spectral-entropy blocks from the cumulative energy at the
// block boundaries (kernels_ct.hpp's scheme with 64-lane scans)
template <typename SH>
__device__ __forceinline__ void tri_features(const TriLayout &L, const Tabs &tb, const TimeFeat &tf, const double *cur,
                                             const double *prv, bool first_frame, double *fv, double *msp, double *bnd,
                                             const int4 *g_meljob, const int4 *g_chjob, int lane) {
    constexpr int NF = SH::NF, C = SH::C, LB = SH::LB, W = SH::W;
    const int4 mjob = g_meljob[lane], cjob = g_chjob[lane];          // (L2; used after the two sweeps)
    const double f0 = L.f0, rf0 = L.rf0, r_half_fs = L.r_half_fs, f0sq = L.f0sq;
    const int kb = C * lane;
    double Xc[C], Xv[C];
#pragma unroll
    for (int m = 0; m < C; ++m) { Xc[m] = cur[kb + m]; Xv[m] = prv[kb + m]; }       // (bins past NF are zeros: see Shape::SLOT)
    double sXa = 0.0, sXb = 0.0, sMa = 0.0, sMb = 0.0, sVa = 0.0, sVb = 0.0, mx = 0.0, csa = 0.0, csb = 0.0;
#pragma unroll
    for (int m = 0; m + 1 < C; m += 2) {
        const double X0 = Xc[m], X1 = Xc[m + 1];
        sXa += X0; sXb += X1;
        sVa += Xv[m]; sVb += Xv[m + 1];
        sMa = fma((double)m, X0, sMa); sMb = fma((double)(m + 1), X1, sMb);
        csa = fma(X0, X0, csa); csb = fma(X1, X1, csb);
        mx = fmax(mx, fmax(X0, X1));
    }
    {
        const double X0 = Xc[C - 1];          // (C is odd)
        sXa += X0; sVa += Xv[C - 1]; sMa = fma((double)(C - 1), X0, sMa); csa = fma(X0, X0, csa); mx = fmax(mx, X0);
    }
    const double cs = csa + csb;
    const double run_incl = wscan_incl(cs);
    const double run_excl = run_incl - cs;
    const double sP = readlane63(run_incl);                  // sum X^2 over all bins
    const double base_k = (double)(kb + 1);
    double sX = sXa + sXb;
    double sIX = f0 * fma(base_k, sX, sMa + sMb);            // sum (k + 1) f0 X
    double sXp = sVa + sVb;
    sX = wsum(sX); sXp = wsum(sXp);
    sIX = wsum(sIX); mx = wmax_nonneg(mx);
    const double sXe = sX + (double)NF * kEps;               // np.sum(X + eps) (:118-119)
    sXp += (double)NF * kEps;
    // spectral entropy (:85-107): cumulative energy at the block boundaries j LB, j = 0 .. 10, written by the lane whose
    // bins contain the boundary (boundary 10 = the total when the blocks tile the spectrum)
    // -- and the roll-off (:127-140: first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2)) from the same running energy: it never
    // decreases, so the first bin that qualifies is the number of bins that do not (a bin below NF always qualifies -- the
    // last one reaches the total, bin 0 does when the total is 0 -- so the zero bins past NF never decide)
    int below = 0;
    {
        const int jb = (kb + LB - 1) / LB;                   // first boundary at or after the lane's first bin
        const int mb = jb * LB - kb;
        const double thr = 0.90 * sP;
        double run = run_excl, cumb = run_excl;
#pragma unroll
        for (int m = 0; m < C; ++m) {
            cumb = (m == mb) ? run : cumb;
            run = fma(Xc[m], Xc[m], run);
            below += (run + kEps > thr) ? 0 : 1;
        }
        if (mb < C && jb <= 10 && jb * LB < NF) bnd[jb] = cumb;
        if (10 * LB == NF && lane == 0) bnd[10] = sP;
    }
    wsync();
    double ent_f;
    {
        const int ib = min(lane, 9);
        const double sf = fast_div(bnd[ib + 1] - bnd[ib], sP + kEps);
        ent_f = wsum((lane < 10) ? -(sf * fast_log2(sf + kEps)) : 0.0);
    }
    // centroid, spread, flux (:57-82, :110-124)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double rden = fast_div(1.0, den);
    const double cen = (sIX * r) * rden;
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    const double cb = base_k - cen * rf0;
    double sSa = 0.0, sSb = 0.0, sFa = 0.0, sFb = 0.0;
#pragma unroll
    for (int m = 0; m + 1 < C; m += 2) {
        const double d0 = cb + (double)m, d1 = cb + (double)(m + 1);
        sSa = fma(d0 * d0, Xc[m], sSa);
        sSb = fma(d1 * d1, Xc[m + 1], sSb);
        const double f0d = Xc[m] * rX - Xv[m] * rXp, f1d = Xc[m + 1] * rX - Xv[m + 1] * rXp;
        sFa = fma(f0d, f0d, sFa);
        sFb = fma(f1d, f1d, sFb);
    }
    {
        const double d0 = cb + (double)(C - 1);
        sSa = fma(d0 * d0, Xc[C - 1], sSa);
        const double f0d = Xc[C - 1] * rX - Xv[C - 1] * rXp;
        sFa = fma(f0d, f0d, sFa);
    }
    double sSp = (sSa + sSb) * (f0sq * r), sFl = sFa + sFb;
    sSp = wsum(sSp);
    sFl = wsum(sFl);
    const double spread = fast_sqrt(sSp * rden);
    const int first = mix::wmin_nonneg_i((below < C) ? kb + below : 0x7fffffff);
    // MFCC (:236-254): mel band energies on all 64 lanes (mel_sums_balanced), log10 in lane m < 40, then the 13 x 40 DCT on 52 lanes
#ifndef PAA_TRI_ABLATE
#define PAA_TRI_ABLATE 0          // timing builds of scripts/rounds/r05 only: 1 = no mel sums, 2 = no chroma gather
#endif
#ifndef PAA_TRI_LANE_JOBS
#define PAA_TRI_LANE_JOBS 1       // 0: A/B build of scripts/rounds/r05/gpu_r05ai.sh -- lane m < 40 walks mel filter m, lane c < 12 its pitch class
#endif
#if PAA_TRI_LANE_JOBS
    {
        const double e = (PAA_TRI_ABLATE & 1) ? 0.0 : mel_sums_balanced(tb, cur, mjob, lane);
        if (lane < 40) msp[lane] = fast_log10(e + kEps);
    }
    // chroma (:277-321)
    double chroma = (PAA_TRI_ABLATE & 2) ? 0.0 : chroma_sums_balanced(tb, cur, cjob, lane);
    chroma = (sP == 0.0) ? chroma / kEps : fast_div(chroma, sP);
#else
    (void)mjob; (void)cjob;
    if (lane < 40) {
        const int lo = tb.mel_lo[lane], cnt = (PAA_TRI_ABLATE & 1) ? 0 : tb.mel_cnt[lane];
        const double *w = tb.mel_w + tb.mel_off[lane];
        double a0 = 0.0, a1 = 0.0;
        int i = 0;
        for (; i + 8 <= cnt; i += 8) {                   // eight bins and weights in flight
            double xb[8], wb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xb[u] = cur[lo + i + u]; wb[u] = w[i + u]; }
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 = fma(xb[u], wb[u], a0); a1 = fma(xb[u + 1], wb[u + 1], a1); }
        }
        for (; i + 2 <= cnt; i += 2) {
            a0 = fma(cur[lo + i], w[i], a0);
            a1 = fma(cur[lo + i + 1], w[i + 1], a1);
        }
        if (i < cnt) a0 = fma(cur[lo + i], w[i], a0);
        msp[lane] = fast_log10((a0 + a1) + kEps);
    }
    const double chroma = (PAA_TRI_ABLATE & 2) ? 0.0 : mix::chroma_class_batched(tb, cur, sP, lane);
#endif
    wsync();
    {
        // lane 4 q + part: ten terms of DCT row q; the four parts meet through two quad permutes
        const int q = min(lane >> 2, 12), part = lane & 3;
        const double *dm = tb.dct + q * tb.dct_stride + 10 * part;
        const double *mv = msp + 10 * part;
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int n = 0; n < 10; n += 2) {
            c0 = fma(dm[n], mv[n], c0);
            c1 = fma(dm[n + 1], mv[n + 1], c1);
        }
        double cc = c0 + c1;
        cc += dpp_mov<PAA_DPP_X1>(cc);
        cc += dpp_mov<PAA_DPP_X2>(cc);
        if (lane < 52 && part == 0) fv[8 + q] = cc;
    }
    {   // population std of the 12 chroma values (:667): lanes 0..11 of the first row
        const double cv = (lane < 12) ? chroma : 0.0;
        const double mean = group_sum(cv) / 12.0;
        const double d = (lane < 12) ? cv - mean : 0.0;
        const double var = group_sum(d * d) / 12.0;
        if (lane < 12) fv[21 + lane] = chroma;
        if (lane == 0) {
            fv[0] = ((double)tf.zc * 0.5) * (1.0 / (double)(W - 1));
            fv[1] = tf.e_tot * (1.0 / (double)W);
            fv[2] = tf.ent_e;
            fv[3] = cen * r_half_fs;
            fv[4] = spread * r_half_fs;
            fv[5] = ent_f;
            fv[6] = first_frame ? 0.0 : sFl;      // first frame: previous spectrum = itself (:624-625)
            fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first * (1.0 / (double)NF);
            fv[33] = fast_sqrt(var);
        }
    }
    wsync();
}

// MODE 0: short-term features (DELTAS: 68 rows), 1: spectrogram rows, 2: chromagram rows
template <typename SH, typename T, int MODE, int DELTAS>
__global__ __launch_bounds__(64 * (MODE == 0 ? SH::NW : SH::NWR), ((MODE == 0 ? SH::NW : SH::NWR) + 3) / 4) void st_tri_kernel(PlanDev P, TriLayout L,
                                                                               const unsigned char *__restrict__ blob,
                                                                               const T *__restrict__ sig,
                                                                               const ClipDev *__restrict__ clips,
                                                                               const ClipNorm *__restrict__ norms,
                                                                               const Tile *__restrict__ tiles, int n_tiles,
                                                                               double *__restrict__ out) {
    constexpr int R1 = SH::R1, R2 = SH::R2, R3 = SH::R3, N = SH::N, W = SH::W, NF = SH::NF, L1 = SH::L1, NQ1 = SH::NQ1;
    constexpr int NJ = SH::NJ, J2 = SH::J2, NR3 = SH::NR3, PP = SH::P, SLOT = SH::SLOT, LT = SH::LT;
    constexpr int N3 = R1 * R2;
    const int NW = L.waves;
    constexpr bool PACKED = SH::PACKED;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.table_bytes / 16; n += 64 * NW) dst4[n] = src4[n];
    }
    __syncthreads();       // the only workgroup-wide barrier
    Tabs tb;
    tb.tw = nullptr; tb.post = nullptr;
    tb.mel_lo = reinterpret_cast<const int *>(smem + L.off_mello);
    tb.mel_cnt = reinterpret_cast<const int *>(smem + L.off_melcnt);
    tb.mel_off = reinterpret_cast<const int *>(smem + L.off_meloff);
    tb.mel_w = reinterpret_cast<const double *>(smem + L.off_melw);
    tb.dct = reinterpret_cast<const double *>(smem + L.off_dct);
    tb.dct_stride = 41;
    tb.ch_start = reinterpret_cast<const int *>(smem + L.off_chstart);
    tb.ch_src = reinterpret_cast<const int *>(smem + L.off_chsrc);
    tb.ch_w = reinterpret_cast<const double *>(smem + L.off_chw);
    const double2 *t_tw2 = reinterpret_cast<const double2 *>(smem + L.off_tw2);
    const uint4 *t_p3 = reinterpret_cast<const uint4 *>(smem + L.off_p3);
    const double2 *g_tw1 = reinterpret_cast<const double2 *>(blob + L.off_g_tw1);
    const double2 *g_post = reinterpret_cast<const double2 *>(blob + L.off_g_post);
    const int4 *g_meljob = reinterpret_cast<const int4 *>(blob + L.off_g_meljob), *g_chjob = reinterpret_cast<const int4 *>(blob + L.off_g_chjob);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_id = blockIdx.x * NW + wave;
    // the two waves of a SIMD are paced against each other (kernels_fast.hpp: the hardware issues oldest-first, so unpaced the
    // older wave runs ahead and the younger one ends alone on the SIMD): a wave publishes its progress twice per frame and
    // raises its priority when it is behind its partner.  pace[0..15] = SIMD id, pace[16..31] = progress (up to 16 waves per
    // workgroup; with three waves on a SIMD a wave paces itself against one of the other two).
    volatile int *pace = reinterpret_cast<volatile int *>(smem + L.off_sync);
    int partner = wave;
    {
        const int my_simd = (int)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3u);     // HW_REG_HW_ID[5:4]
        if ((threadIdx.x & 63) == 0) {
            pace[wave] = my_simd;
            pace[16 + wave] = (tile_id < n_tiles) ? 0 : 0x7fffffff;
        }
        __syncthreads();
        for (int w = 0; w < NW; ++w) partner = (w != wave && pace[w] == my_simd) ? w : partner;
        partner = __builtin_amdgcn_readfirstlane(partner);
    }
    if (tile_id >= n_tiles) return;
    int n_done = 0;
#ifndef PAA_TRI_PACING
#define PAA_TRI_PACING 1
#endif
#define PAA_TRI_PACE(half_)                                                                            \
    if (PAA_TRI_PACING) {                                                                              \
        const int mine_ = 2 * n_done + (half_);                                                        \
        if (lane == 0) pace[16 + wave] = mine_;                                                         \
        const int other_ = __builtin_amdgcn_readfirstlane(pace[16 + partner]);                          \
        const int d_ = mine_ - other_;                                                                 \
        if (d_ < 0) __builtin_amdgcn_s_setprio(3);                                                     \
        else if (d_ > 0) __builtin_amdgcn_s_setprio(0);                                                \
        else __builtin_amdgcn_s_setprio(1);                                                            \
    }
    double *slots = reinterpret_cast<double *>(smem + L.table_bytes) + (size_t)wave * (MODE == 0 ? SH::WAVE_DOUBLES : SH::WAVE_DOUBLES_ROWS);
    double *fv = slots + ((MODE == 0) ? 2 * SLOT : SLOT);          // (row instances never touch fv / msp / bnd)
    double *msp = fv + 48;
    double *bnd = msp + 40;

    const Tile tl = tiles[tile_id];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = wave_clip_norm<T>(P, c, norms, tl.clip, (int)(threadIdx.x & 63));
    const T *x0 = sig + c.sample_off + P.frame_origin;
    const long long Tc = c.T;
    double *oc = out + c.out_off;
    const double sc = sample_scale<T>();
    // (wave-uniform: pinned into scalar registers, or they would live in vector registers through the whole loop)
    const double mean = f800::uni(nm.mean), inv = f800::uni(nm.inv);
    const double mscale = f800::uni((PACKED ? 0.5 : 1.0) * inv * (1.0 / (double)NF));      // X / len(X) (:621); y = d * inv; E, O carry 1/2
    const double inv2 = f800::uni(inv * inv);               // energies of y = d * inv
    constexpr bool INT_T = !std::is_same<T, double>::value;
    SignRule sr = {0, 0, 0};
    if (INT_T && MODE == 0) sr = sign_rule<T>(nm.mean);
    // (the packed int16 path of the even windows forms its codes from nm.zb / nm.mu_whole: the same rule in 16-bit arithmetic)
    const int zc_shift = !INT_T ? 0 : (PACKED && std::is_same<T, int16_t>::value) ? (nm.mu_whole ? 0 : 1) : sr.sh;

    const int hneed = (MODE == 0) ? (DELTAS ? 2 : 1) : 0;
    const int h = min(hneed, tl.t0);
    const int tend = tl.t0 + tl.cnt;
    double vprev = 0.0;
    int odd = 0;
    RowChunk rc, rcd;
#pragma unroll
    for (int i = 0; i < 8; ++i) { rc.h[i] = 0.0; rcd.h[i] = 0.0; }
    const int lane_id = threadIdx.x & 63;
    PAA_T0()
    for (int t = tl.t0 - h; t < tend; ++t, odd ^= 1) {
        // everything derived from the lane number is recomputed per frame: hoisted out of the loop, the per-lane plane
        // addresses and table pointers of all passes are some 60 loop-invariant registers (they ended up in scratch)
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        PAA_TRI_PACE(0)
        // pass-2 job of this lane: (q1, b); idle lanes shadow the last job (their plane writes are masked)
        constexpr int H1 = SH::H1, H2 = SH::H2;
        int part2 = 0;
        if constexpr (H2 > 1) {
#pragma unroll
            for (int h = 1; h < H2; ++h) part2 = (lane >= h * J2) ? h : part2;
        }
        const int m2 = min(lane - part2 * J2, J2 - 1);
        const int q1_2 = m2 / R3, b_2 = m2 - R3 * q1_2;
        const bool act2 = lane < J2 * H2;
        const int *t_split = reinterpret_cast<const int *>(smem + L.off_split);
        // a part reads its inputs in the order n -> (n ginv) mod R (SplitSel); R <= 29: (n ginv) * ceil(2^16 / R) >> 16 is the quotient
        int ginv2 = 1;
        if constexpr (H2 > 1) {
#pragma unroll
            for (int h = 1; h < H2; ++h) ginv2 = (part2 == h) ? mod_inverse(SplitSel<R2, H2>::g(h), R2) : ginv2;
        }
        auto col2 = [&](int k) {
            if constexpr (H2 > 1) {
                const int v = k * ginv2;
                return v - R2 * ((v * ((65536 + R2 - 1) / R2)) >> 16);
            } else {
                return k;
            }
        };
        double *cur = slots + ((MODE == 0 && odd) ? SLOT : 0);
        const double *prv = slots + ((MODE == 0 && !odd) ? SLOT : 0);
        const T *xf = x0 + (long long)t * P.S;
        const bool want = (MODE == 0) && ((t >= tl.t0) || (DELTAS && t == tl.t0 - 1));

        // ---------------- load, time domain (ShortTermFeatures.py:22-51), pass 1: a[u][q1] = A_j[q1] W_N^(j q1), j = lane + 64 u
        double2 a[NJ][NQ1];
        TimeFeat tf;
        tf.e_tot = 0.0; tf.ent_e = 0.0; tf.zc = 0;
        {
            double eb[11];                 // ten entropy blocks + the tail the reference leaves out of them (:37-41)
#pragma unroll
            for (int b = 0; b < 11; ++b) eb[b] = 0.0;
            int zcv = 0, carry = 0;        // per-lane sign-change count, the sign carried from the slot before
            // wave totals of the partials (before pass 1: the partials' registers are free for the codelets).  The eleven block
            // sums go through LDS transposed -- lane l writes its partials at [b][l], lane 4 b + p adds 16 of block b's values,
            // two quad permutes join the four parts -- 70 instructions instead of eleven 20-instruction wave reductions; the
            // slot is free here (it becomes the exchange plane after pass 1)
            auto finish_time = [&]() {
                if (MODE == 0 && want) {
                    double *sc11 = (SH::TSCR > 0) ? bnd + 12 : cur;      // [11][65]
                    wsync();                                  // (the previous frame's readers of this slot are done)
                    // row stride 65 and part p starting 8 (p >> 1) elements into its sixteen: at every step the 32 lanes of a read
                    // group (eight blocks x four parts) touch 32 different banks
#pragma unroll
                    for (int b = 0; b < 11; ++b) sc11[65 * b + lane] = eb[b];
                    wsync();
                    const int bq = min(lane >> 2, 10), part = lane & 3;
                    const double *rowp = sc11 + 65 * bq + 16 * part;
                    const double *half_a = rowp + 8 * (part >> 1), *half_b = rowp + 8 * ((part >> 1) ^ 1);
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc += half_a[i];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc += half_b[i];
                    acc += dpp_mov<PAA_DPP_X1>(acc);
                    acc += dpp_mov<PAA_DPP_X2>(acc);
                    acc = (lane < 44) ? acc * inv2 : 0.0;     // lanes 4 b .. 4 b + 3: energy of block b (b = 10: the tail)
                    const double tot = wsum((part == 0) ? acc : 0.0);
                    tf.e_tot = tot;
                    // (integer sign codes of a clip whose mean is not a whole count are 1 / 2: their differences count double)
                    tf.zc = wsum_i(zcv) << zc_shift;
                    const double s = fast_div(acc, tot + kEps);
                    tf.ent_e = wsum((lane < 40 && part == 0) ? -(s * fast_log2(s + kEps)) : 0.0);
                    wsync();
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (PACKED) {
                const bool act1 = lane < L1;
                const int jj = act1 ? lane : L1 - 1;
                double2 v[R1];
                // int16 PCM: the raw pair words are kept for the sign changes (packed 16-bit arithmetic, below)
                constexpr bool RAW16 = std::is_same<T, int16_t>::value;
                constexpr bool ST16 = std::is_same<T, stereo16>::value;
                int wr[RAW16 ? R1 : 1];
                int w0[ST16 ? R1 : 1], w1[ST16 ? R1 : 1];     // stereo: the two raw frames of a pair (summed L + R row by row, below)
                // idle lanes shadow the last job with scale and mean 0: their samples are exact zeros (no energy, nothing to mask)
                const double scl = act1 ? sc : 0.0, meanl = act1 ? mean : 0.0;
                const T *xb = xf + 2 * jj;       // (one address per lane: the rows are immediate offsets of the loads)
#pragma unroll
                for (int r = 0; r < R1; ++r) {
                    if constexpr (RAW16) {
                        typedef int w32 __attribute__((aligned(2)));
                        wr[r] = *reinterpret_cast<const w32 *>(xb + 2 * L1 * r);
                        v[r] = make_double2(fma((double)(short)(wr[r] & 0xffff), scl, -meanl), fma((double)(wr[r] >> 16), scl, -meanl));
                    } else if constexpr (ST16) {
                        typedef int vec2 __attribute__((ext_vector_type(2), aligned(4)));
                        const vec2 s2 = *reinterpret_cast<const vec2 *>(xb + 2 * L1 * r);
                        w0[r] = s2.x; w1[r] = s2.y;
                        if (MODE != 0)
                            v[r] = make_double2(fma((double)stereo_word_sum(w0[r]), scl, -meanl), fma((double)stereo_word_sum(w1[r]), scl, -meanl));
                    } else {
                        const double2 x = ct::PairLoad<T>::get(xb + 2 * L1 * r);
                        v[r] = make_double2(fma(x.x, scl, -meanl), fma(x.y, scl, -meanl));
                    }
                }
                // sign(x / 2^15 - mean) of an int16 sample = sign(x - mu), mu = mean 2^15, as the sign CODE of device_common.hpp in
                // packed saturating 16-bit arithmetic (kernels_fast.hpp): code = clamp(sat(x - (floor(mu) - 1)), lo, 2), lo = 0 when
                // mu is a whole number (a sample can sit on the mean), 1 otherwise (differences count double: the shift at the end)
                typedef f800::s16x2 s16x2;
                const short zb1_ = (short)max(nm.zb - 1, -32768), lo_ = nm.mu_whole ? (short)0 : (short)1;
                const s16x2 zc_b = {zb1_, zb1_}, zc_lo = {lo_, lo_}, zc_two = {2, 2};
                int zacc = 0;
                unsigned carryw = 0;
                PAA_TICK(0)
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < R1; ++r) {
                        int x0 = 0, x1 = 0;
                        if constexpr (ST16) {
                            x0 = stereo_word_sum(w0[r]); x1 = stereo_word_sum(w1[r]);
                            v[r] = make_double2(fma((double)x0, scl, -meanl), fma((double)x1, scl, -meanl));
                        }
                        const double d0 = v[r].x, d1 = v[r].y;
                        const double e = fma(d0, d0, d1 * d1);
                        // the row holds samples [n0, n0 + 2 L1), lane p the pair (n0 + 2 p, + 1); block(n) = min(n / LT, 10) (10: the tail
                        // the reference leaves out of the blocks).  Everything below is static per register row: a row inside one block
                        // is a plain add, a block boundary at an even sample splits the LANES, one at an odd sample (LT odd: power-of-two
                        // windows such as 512) also splits the pair of the lane it falls into
                        const int n0 = 2 * L1 * r, n1 = n0 + 2 * L1;
                        const int jfirst = (n0 / LT < 10) ? n0 / LT : 10, jlast = ((n1 - 1) / LT < 10) ? (n1 - 1) / LT : 10;
#pragma unroll
                        for (int jb = 0; jb < 11; ++jb) {
                            if (jb < jfirst || jb > jlast) continue;
                            const int lo_s = ((jb * LT > n0) ? jb * LT : n0) - n0;                               // first sample of block jb in the row
                            const int hi_s = ((jb < 10 && (jb + 1) * LT < n1) ? (jb + 1) * LT : n1) - n0;        // one past its last
                            if (lo_s == 0 && hi_s == 2 * L1) {
                                eb[jb] += e;
                            } else if (lo_s % 2 == 0 && hi_s % 2 == 0) {
                                eb[jb] += (lane >= lo_s / 2 && lane < hi_s / 2) ? e : 0.0;
                            } else {
                                const int s_even = 2 * lane, s_odd = 2 * lane + 1;
                                const double part0 = (s_even >= lo_s && s_even < hi_s) ? d0 * d0 : 0.0;
                                eb[jb] += (s_odd >= lo_s && s_odd < hi_s) ? fma(d1, d1, part0) : part0;
                            }
                        }
                        if constexpr (RAW16) {
                            // both samples of the pair at once: {s_even, s_odd} against {s of the sample before the pair, s_even}
                            const s16x2 cur2 = __builtin_bit_cast(s16x2, wr[r]);
                            const s16x2 sg = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_elementwise_sub_sat(cur2, zc_b), zc_lo), zc_two);
                            const unsigned sgw = __builtin_bit_cast(unsigned, sg);
                            if (r == 0) carryw = (unsigned)__builtin_amdgcn_readfirstlane((int)sgw) << 16;      // the first sample has no left one
                            const unsigned leftw = (unsigned)wave_shr1((int)sgw, (int)carryw);         // the lane below (lane 0: the row above)
                            const unsigned shw = __builtin_amdgcn_alignbit(sgw, leftw, 16);
                            asm("v_sad_u16 %0, %1, %2, %0" : "+v"(zacc) : "v"(sgw), "v"(shw));      // both halves' |code - code'|
                            carryw = (unsigned)__builtin_amdgcn_readlane((int)sgw, L1 - 1);
                        } else {
                            const int sa = ST16 ? sgn1(x0, sr) : sgn1(d0), sb = ST16 ? sgn1(x1, sr) : sgn1(d1);
                            if (r == 0) carry = __builtin_amdgcn_readfirstlane(sa);       // the frame's first sample has no left one
                            const int left = wave_shr1(sb, carry);
                            sad_acc(zcv, sb, sa);
                            sad_acc(zcv, sa, left);
                            carry = __builtin_amdgcn_readlane(sb, L1 - 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                zcv = act1 ? (RAW16 ? zacc : zcv) : 0;
                finish_time();
                PAA_TICK(1)
                Cd<R1>::run(v);
#pragma unroll
                for (int q0 = 0; q0 < R1; q0 += 4) {
                    double2 wl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (q0 + k < R1 && q0 + k > 0) wl[k] = g_tw1[(q0 + k) * L1 + jj];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (q0 + k < R1) a[0][q0 + k] = (q0 + k > 0) ? cmul(v[Cd<R1>::pos(q0 + k)], wl[k]) : v[Cd<R1>::pos(0)];
                    __builtin_amdgcn_sched_barrier(0);      // (all twiddles requested up front cost 80 registers)
                }
            } else {
                double xr[NJ][R1];
                int xi[INT_T ? NJ : 1][INT_T ? R1 : 1];        // integer samples (stereo: L + R), converted row by row below
                double scl[NJ], meanl[NJ];                     // idle lanes shadow the last job with scale and mean 0: exact zeros
                // split first pass: lane = part L1 + job; part 0 holds its rows in natural order (it also owns the time-domain sums),
                // parts 1 .. fetch row (r ginv) mod R1 into register row r
                int part1 = 0, ginv1 = 1;
                if constexpr (H1 > 1) {
#pragma unroll
                    for (int h = 1; h < H1; ++h) {
                        part1 = (lane >= h * L1) ? h : part1;
                        ginv1 = (lane >= h * L1) ? mod_inverse(SplitSel<R1, H1>::g(h), R1) : ginv1;
                    }
                }
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    const int j = (H1 > 1) ? lane - part1 * L1 : lane + 64 * u;
                    const bool in1 = (H1 > 1) ? lane < H1 * L1 : j < L1;
                    const int jj = in1 ? j : L1 - 1;
                    scl[u] = in1 ? sc : 0.0;
                    meanl[u] = in1 ? mean : 0.0;
                    const T *xb = xf + jj;       // (one address per lane: the rows are immediate offsets of the loads)
#pragma unroll
                    for (int r = 0; r < R1; ++r) {
                        int row = r;
                        if constexpr (H1 > 1) {
                            const int v = r * ginv1;
                            row = v - R1 * ((v * ((65536 + R1 - 1) / R1)) >> 16);
                        }
                        if constexpr (INT_T) {
                            xi[u][r] = load_int<T>(xb + L1 * row);
                            if (MODE != 0) xr[u][r] = fma((double)xi[u][r], scl[u], -meanl[u]);
                        } else {
                            xr[u][r] = fma(load_sample<T>(xb + L1 * row), scl[u], -meanl[u]);
                        }
                    }
                }
                PAA_TICK(0)
                if (MODE == 0) {
                    int zcu[NJ];
#pragma unroll
                    for (int u = 0; u < NJ; ++u) zcu[u] = 0;
#pragma unroll
                    for (int r = 0; r < R1; ++r)
#pragma unroll
                        for (int u = 0; u < NJ; ++u) {
                            if constexpr (INT_T) xr[u][r] = fma((double)xi[u][r], scl[u], -meanl[u]);
                            const double d = xr[u][r];
                            const double e = (H1 > 1 && lane >= L1) ? 0.0 : d * d;       // (split first pass: part 0 owns the sums)
                            const int n0 = L1 * r + 64 * u;                     // sample of lane 0
                            const int jlo = (n0 / LT < 10) ? n0 / LT : 10;
                            const int jth = (jlo >= 10) ? 64 : (jlo + 1) * LT - n0;
                            if (jth >= 64) {
                                eb[jlo] += e;
                            } else {
                                eb[jlo] += (lane < jth) ? e : 0.0;
                                eb[jlo + 1] += (lane >= jth) ? e : 0.0;
                            }
                            int sg;
                            if constexpr (INT_T) sg = sgn1(xi[u][r], sr);
                            else sg = sgn1(d);
                            if (r == 0 && u == 0) carry = __builtin_amdgcn_readfirstlane(sg);
                            const int left = wave_shr1(sg, carry);      // sample n - 1: the lane below / the last lane of the slot before
                            sad_acc(zcu[u], sg, left);
                            carry = __builtin_amdgcn_readlane(sg, (L1 - 1 - 64 * u < 63) ? L1 - 1 - 64 * u : 63);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                    for (int u = 0; u < NJ; ++u) zcv += (lane + 64 * u < L1) ? zcu[u] : 0;
                }
                finish_time();
                PAA_TICK(1)
                if constexpr (H1 > 1) {
                    // every part forms DC + the SplitSel outputs of its permuted rows; slot s stands for the true output index t_s
                    // (conjugated when (g q) mod R1 > R1 / 2), times W_N^(j t_s)
                    typedef SplitSel<R1, H1> S1;
                    const int jj = (lane < H1 * L1) ? lane - part1 * L1 : L1 - 1;
                    rdft_prime_sel<R1, S1>(xr[0], a[0]);
                    int code1[1 + S1::NQ];
                    double2 wl1[1 + S1::NQ];
#pragma unroll
                    for (int sl = 1; sl <= S1::NQ; ++sl) {
                        code1[sl] = t_split[8 * part1 + sl];
                        wl1[sl] = g_tw1[(code1[sl] & 0xff) * L1 + jj];
                    }
#pragma unroll
                    for (int sl = 1; sl <= S1::NQ; ++sl) {
                        const double2 z = make_double2(a[0][sl].x, (code1[sl] & 0x100) ? -a[0][sl].y : a[0][sl].y);
                        a[0][sl] = cmul(z, wl1[sl]);
                    }
                } else
#pragma unroll
                for (int u = 0; u < NJ; ++u) {
                    const int j = lane + 64 * u;
                    const int jj = (j < L1) ? j : L1 - 1;
                    RCd<R1>::run(xr[u], a[u]);
#pragma unroll
                    for (int q0 = 1; q0 < NQ1; q0 += 4) {
                        double2 wl[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (q0 + k < NQ1) wl[k] = g_tw1[(q0 + k) * L1 + jj];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (q0 + k < NQ1) a[u][q0 + k] = cmul(a[u][q0 + k], wl[k]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        constexpr int NS1 = (H1 > 1) ? 1 + SplitNQ<R1, H1>::value : NQ1;      // pass-1 output slots a lane holds
#pragma unroll
        for (int u = 0; u < NJ; ++u)
#pragma unroll
            for (int q = 0; q < NS1; ++q) asm volatile("" : "+v"(a[u][q].x), "+v"(a[u][q].y));
        PAA_TICK(2)
        wsync();               // the previous frame's readers of this slot are done

        // ---------------- exchange 1: element (j, q1) at plane[q1 PP + j]; pass-2 lane (q1, b) reads B[R3 a + b][q1]
        double2 c2[R2];
        // split second pass: the butterfly runs plane by plane as the exchange delivers them (real parts, then imaginary parts)
        constexpr int NP2 = (H2 > 1) ? SplitNQ<R2, H2>::value : 1;
        double p2_tot_r = 0.0, p2_tot_i = 0.0, p2_Ar[NP2], p2_Br[NP2], p2_Ai[NP2], p2_Bi[NP2];
        {
            double *pl = cur;
            int w1[NS1];           // split first pass: where slot s of this lane goes (second copies and idle lanes: a dummy double behind the plane)
            if constexpr (H1 > 1) {
                int part1 = 0;
#pragma unroll
                for (int h = 1; h < H1; ++h) part1 = (lane >= h * L1) ? h : part1;
                const int j = lane - part1 * L1;
#pragma unroll
                for (int sl = 0; sl < NS1; ++sl) {
                    const int code = t_split[8 * part1 + sl];
                    w1[sl] = ((code & 0x200) && lane < H1 * L1) ? (code & 0xff) * PP + j : SH::PLANE;
                }
#pragma unroll
                for (int sl = 0; sl < NS1; ++sl) pl[w1[sl]] = a[0][sl].x;
            } else {
#pragma unroll
                for (int u = 0; u < NJ; ++u)
                    if (lane + 64 * u < L1) {
#pragma unroll
                        for (int q = 0; q < NQ1; ++q) pl[q * PP + lane + 64 * u] = a[u][q].x;
                    }
            }
            wsync();
            if constexpr (H2 > 1) {
                double cx[R2];
#pragma unroll
                for (int k = 0; k < R2; ++k) cx[k] = pl[q1_2 * PP + R3 * col2(k) + b_2];
                prime_sel_plane<R2, SplitSel<R2, H2>>(cx, p2_tot_r, p2_Ar, p2_Br);
            } else {
#pragma unroll
            for (int k = 0; k < R2; ++k) c2[k].x = pl[q1_2 * PP + R3 * col2(k) + b_2];
            }
            wsync();
            if constexpr (H1 > 1) {
#pragma unroll
                for (int sl = 0; sl < NS1; ++sl) pl[w1[sl]] = a[0][sl].y;
            } else {
#pragma unroll
                for (int u = 0; u < NJ; ++u)
                    if (lane + 64 * u < L1) {
#pragma unroll
                        for (int q = 0; q < NQ1; ++q) pl[q * PP + lane + 64 * u] = a[u][q].y;
                    }
            }
            wsync();
            if constexpr (H2 > 1) {
                double cx[R2];
#pragma unroll
                for (int k = 0; k < R2; ++k) cx[k] = pl[q1_2 * PP + R3 * col2(k) + b_2];
                prime_sel_plane<R2, SplitSel<R2, H2>>(cx, p2_tot_i, p2_Ai, p2_Bi);
            } else {
#pragma unroll
            for (int k = 0; k < R2; ++k) c2[k].y = pl[q1_2 * PP + R3 * col2(k) + b_2];
            }
            wsync();
        }
        PAA_TICK(3)
        // ---------------- pass 2: radix R2 over a, outputs times W_L1^(b q2)
        constexpr int NS2 = (H2 > 1) ? 1 + 2 * SplitNQ<R2, H2>::value : R2;        // pass-2 outputs a lane holds
        double2 o2[(H2 > 1) ? NS2 : 1];
        int code2[(H2 > 1) ? NS2 : 1];
        if constexpr (H2 > 1) {
            // every part forms DC + the SplitSel output pairs of its permuted column; slot s stands for the true output index of the
            // host table (bit 9: this lane is the one that delivers it), times W_L1^(b t)
            o2[0] = make_double2(p2_tot_r, p2_tot_i);
#pragma unroll
            for (int i = 0; i < NP2; ++i) {
                o2[1 + 2 * i] = make_double2(p2_Ar[i] + p2_Bi[i], p2_Ai[i] - p2_Br[i]);          // X[q] = A - i B
                o2[2 + 2 * i] = make_double2(p2_Ar[i] - p2_Bi[i], p2_Ai[i] + p2_Br[i]);          // X[R - q] = A + i B
            }
#pragma unroll
            for (int sl = 0; sl < NS2; ++sl) code2[sl] = t_split[8 * (H1 > 1 ? H1 : 1) + 16 * part2 + sl];
            if constexpr (R3 > 1) {
#pragma unroll
                for (int s0 = 1; s0 < NS2; s0 += 4) {
                    double2 wl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (s0 + k < NS2) wl[k] = t_tw2[(code2[s0 + k] & 0xff) * R3 + b_2];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (s0 + k < NS2) o2[s0 + k] = cmul(o2[s0 + k], wl[k]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int sl = 0; sl < NS2; ++sl) asm volatile("" : "+v"(o2[sl].x), "+v"(o2[sl].y));
        } else {
        Cd<R2>::run(c2);
#pragma unroll
        for (int q0 = 1; q0 < (R3 > 1 ? R2 : 0); q0 += 4) {
            double2 wl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q0 + k < R2) wl[k] = t_tw2[(q0 + k) * R3 + b_2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q0 + k < R2) c2[Cd<R2>::pos(q0 + k)] = cmul(c2[Cd<R2>::pos(q0 + k)], wl[k]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (pinned: the optimiser would sink the imaginary halves of these products into the masked plane store that uses
        // them, keeping the twiddles and the codelet's outputs alive side by side -- 80 registers)
#pragma unroll
        for (int q = 0; q < R2; ++q) asm volatile("" : "+v"(c2[q].x), "+v"(c2[q].y));
        }
        PAA_TICK(4)
        // ---------------- exchange 2: element (q1, b, q2) at plane[q1 PP + q2 R3 + b]; pass 3 + |X| / num_fft (:617-621)
        if constexpr (R3 == 1) {
            // two-pass shapes: lane q1 holds Z[q1 + R1 q2] already; bin k or its mirror N - k (real input) -- the host table says
            // where (idle lanes and the results no bin takes: the parking double at index NF); no predicates around the stores
            const unsigned short *t_st = reinterpret_cast<const unsigned short *>(t_p3);
            unsigned char *plb = reinterpret_cast<unsigned char *>(cur);
            unsigned so[NS2];
#pragma unroll
            for (int q = 0; q < NS2; ++q) so[q] = t_st[q * 64 + lane];
#pragma unroll
            for (int q = 0; q < NS2; ++q) {
                double2 z;
                if constexpr (H2 > 1) z = o2[q];          // (split second pass: slot q of this lane; the table knows its bin)
                else z = c2[Cd<R2>::pos(q)];
                const double mg = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * mscale;
                *reinterpret_cast<double *>(plb + so[q]) = mg;
            }
        } else if constexpr (PACKED) {
            constexpr int PE = SH::PE;
            double2 dA[NR3][R3], dB[NR3][R3];
            // host table, PE x 8 ushorts per lane job: byte offsets {plane elements of job A, of job B, (|X[k]|, |X[N - k]|) x R3}
            unsigned pw32[NR3][4 * PE];
#pragma unroll
            for (int u = 0; u < NR3; ++u)
#pragma unroll
                for (int i = 0; i < PE; ++i) {
                    const uint4 q4 = t_p3[(lane + 64 * u) * PE + i];
                    pw32[u][4 * i] = q4.x; pw32[u][4 * i + 1] = q4.y; pw32[u][4 * i + 2] = q4.z; pw32[u][4 * i + 3] = q4.w;
                }
            double *pl = cur;
            unsigned char *plb = reinterpret_cast<unsigned char *>(cur);
            if (act2) {
#pragma unroll
                for (int q = 0; q < R2; ++q) pl[q1_2 * PP + q * SH::R3P + b_2] = c2[Cd<R2>::pos(q)].x;
            }
            wsync();
            // (R3 = 2 / 4 with unpadded groups: a job's R3 plane elements are consecutive and 16-byte aligned -- read as double2: half
            // the LDS cycles of four-way conflicting 8-byte reads)
            constexpr bool VEC3 = (R3 == 2 || R3 == 4) && SH::R3P == R3 && PP % 2 == 0 && SH::SLOT % 2 == 0 && SH::WAVE_DOUBLES % 2 == 0;
#pragma unroll
            for (int u = 0; u < NR3; ++u) {
                const double *pa = reinterpret_cast<const double *>(plb + (pw32[u][0] & 0xffffu));
                const double *pb = reinterpret_cast<const double *>(plb + (pw32[u][0] >> 16));
                if constexpr (VEC3) {
#pragma unroll
                    for (int b = 0; b < R3; b += 2) {
                        const double2 va = *reinterpret_cast<const double2 *>(pa + b), vb = *reinterpret_cast<const double2 *>(pb + b);
                        dA[u][b].x = va.x; dA[u][b + 1].x = va.y; dB[u][b].x = vb.x; dB[u][b + 1].x = vb.y;
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < R3; ++b) { dA[u][b].x = pa[b]; dB[u][b].x = pb[b]; }
                }
            }
            wsync();
            if (act2) {
#pragma unroll
                for (int q = 0; q < R2; ++q) pl[q1_2 * PP + q * SH::R3P + b_2] = c2[Cd<R2>::pos(q)].y;
            }
            wsync();
#pragma unroll
            for (int u = 0; u < NR3; ++u) {
                const double *pa = reinterpret_cast<const double *>(plb + (pw32[u][0] & 0xffffu));
                const double *pb = reinterpret_cast<const double *>(plb + (pw32[u][0] >> 16));
                if constexpr (VEC3) {
#pragma unroll
                    for (int b = 0; b < R3; b += 2) {
                        const double2 va = *reinterpret_cast<const double2 *>(pa + b), vb = *reinterpret_cast<const double2 *>(pb + b);
                        dA[u][b].y = va.x; dA[u][b + 1].y = va.y; dB[u][b].y = vb.x; dB[u][b + 1].y = vb.y;
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < R3; ++b) { dA[u][b].y = pa[b]; dB[u][b].y = pb[b]; }
                }
            }
            wsync();
            PAA_TICK(5)
#pragma unroll
            for (int u = 0; u < NR3; ++u) {
                // host table: where each result goes (a self-paired job meets each of its pairs {k, N - k} twice: only the smaller
                // index is a bin, the other copy -- like everything idle lanes of the last round compute -- is parked at index NF);
                // no predicates, no branches around the stores.  Job 0 (lane 0 of round 0) is (q1, q2) = (0, 0): its partner is itself
                double2 pw[R3];
#pragma unroll
                for (int k3 = 0; k3 < R3; ++k3) pw[k3] = g_post[(lane + 64 * u) * R3 + k3];
                Cd<R3>::run(dA[u]);
                Cd<R3>::run(dB[u]);
#pragma unroll
                for (int k3 = 0; k3 < R3; ++k3) {
                    double2 zk = dA[u][Cd<R3>::pos(k3)];
                    double2 zm = dB[u][Cd<R3>::pos(R3 - 1 - k3)];
                    if (u == 0) {
                        if constexpr (SH::FOLD) {
                            // lane 0 of round 0: job 0 (dA) and the self-paired job R1 R2 / 2 (dB) share the four slots (the host
                            // table holds their bins and post-twiddles): slot 0 <- B's pair (0, 3), slot 1 <- A's (1, 3), slot 2 <-
                            // A's (2, 2), slot 3 <- B's (1, 2); bin 0 follows below
                            const double2 zk0 = (k3 == 0) ? dB[u][Cd<R3>::pos(0)] : (k3 == 3) ? dB[u][Cd<R3>::pos(1)] : dA[u][Cd<R3>::pos(k3)];
                            const double2 zm0 = (k3 == 0) ? dB[u][Cd<R3>::pos(3)] : (k3 == 1) ? dA[u][Cd<R3>::pos(3)]
                                              : (k3 == 2) ? dA[u][Cd<R3>::pos(2)] : dB[u][Cd<R3>::pos(2)];
                            zk = make_double2((lane == 0) ? zk0.x : zk.x, (lane == 0) ? zk0.y : zk.y);
                            zm = make_double2((lane == 0) ? zm0.x : zm.x, (lane == 0) ? zm0.y : zm.y);
                        } else {
                            const double2 z0 = dA[u][Cd<R3>::pos((R3 - k3) % R3)];
                            zm = make_double2((lane == 0) ? z0.x : zm.x, (lane == 0) ? z0.y : zm.y);
                        }
                    }
                    // 2E = Z[k] + conj Z[N-k],  2O = -i (Z[k] - conj Z[N-k]);  X[k] = E + w^k O,  X[N-k] = conj(E - w^k O)
                    const double2 e = make_double2(zk.x + zm.x, zk.y - zm.y);
                    const double2 o = make_double2(zk.y + zm.y, zm.x - zk.x);
                    const double2 wo = cmul(pw[k3], o);
                    const double xr_ = e.x + wo.x, xi_ = e.y + wo.y, yr_ = e.x - wo.x, yi_ = e.y - wo.y;
                    const double mk = mag_sqrt(fma(xr_, xr_, xi_ * xi_)) * mscale;
                    const double mm = mag_sqrt(fma(yr_, yr_, yi_ * yi_)) * mscale;
                    const unsigned st = pw32[u][1 + k3];
                    *reinterpret_cast<double *>(plb + (st & 0xffffu)) = mk;
                    *reinterpret_cast<double *>(plb + (st >> 16)) = mm;
                }
                if constexpr (SH::FOLD) {
                    // bin 0: Z[0] pairs with itself and w^0 = 1, so E + w O = (Re Z[0] + Im Z[0]) x 2 (same form as the slots above)
                    if (u == 0 && lane == 0) {
                        const double2 z0 = dA[0][Cd<R3>::pos(0)];
                        const double xr0 = (z0.x + z0.x) + (z0.y + z0.y);
                        *reinterpret_cast<double *>(plb) = mag_sqrt(xr0 * xr0) * mscale;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            double2 d3[NR3][R3];
            uint4 pe[NR3];           // host table: byte offsets of the job's plane elements and of its R3 magnitudes' bins
#pragma unroll
            for (int u = 0; u < NR3; ++u) pe[u] = t_p3[lane + 64 * u];
            double *pl = cur;
            unsigned char *plb = reinterpret_cast<unsigned char *>(cur);
            int w2[(H2 > 1) ? NS2 : 1];        // split second pass: where slot s goes (second copies, idle lanes: the dummy double behind the plane)
            if constexpr (H2 > 1) {
#pragma unroll
                for (int sl = 0; sl < NS2; ++sl)
                    w2[sl] = ((code2[sl] & 0x200) && act2) ? q1_2 * PP + (code2[sl] & 0xff) * SH::R3P + b_2 : SH::PLANE;
#pragma unroll
                for (int sl = 0; sl < NS2; ++sl) pl[w2[sl]] = o2[sl].x;
            } else if (act2) {
#pragma unroll
                for (int q = 0; q < R2; ++q) pl[q1_2 * PP + q * SH::R3P + b_2] = c2[Cd<R2>::pos(q)].x;
            }
            wsync();
#pragma unroll
            for (int u = 0; u < NR3; ++u) {
                const double *pa = reinterpret_cast<const double *>(plb + (pe[u].x & 0xffffu));
#pragma unroll
                for (int b = 0; b < R3; ++b) d3[u][b].x = pa[b];
            }
            wsync();
            if constexpr (H2 > 1) {
#pragma unroll
                for (int sl = 0; sl < NS2; ++sl) pl[w2[sl]] = o2[sl].y;
            } else if (act2) {
#pragma unroll
                for (int q = 0; q < R2; ++q) pl[q1_2 * PP + q * SH::R3P + b_2] = c2[Cd<R2>::pos(q)].y;
            }
            wsync();
#pragma unroll
            for (int u = 0; u < NR3; ++u) {
                const double *pa = reinterpret_cast<const double *>(plb + (pe[u].x & 0xffffu));
#pragma unroll
                for (int b = 0; b < R3; ++b) d3[u][b].y = pa[b];
            }
            wsync();
            PAA_TICK(5)
#pragma unroll
            for (int u = 0; u < NR3; ++u) {
                const unsigned w4[4] = {pe[u].x, pe[u].y, pe[u].z, pe[u].w};
                Cd<R3>::run(d3[u]);
#pragma unroll
                for (int k3 = 0; k3 < R3; ++k3) {
                    const double2 z = d3[u][Cd<R3>::pos(k3)];
                    const double mg = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * mscale;
                    // bin k, or its mirror N - k (|X[N - k]| = |X[k]| for real input), or the parking double: entry 1 + k3
                    const unsigned wd = w4[(1 + k3) / 2];
                    *reinterpret_cast<double *>(plb + (((1 + k3) & 1) ? (wd >> 16) : (wd & 0xffffu))) = mg;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // bins NF .. 64 C of the slot are zeros for the feature sweeps (the exchange planes may have written there)
        if (MODE == 0) {
            for (int k = NF + lane; k < 64 * SH::C; k += kWave) cur[k] = 0.0;
        }
        wsync();
        PAA_TRI_PACE(1)
        PAA_TICK(6)

        if (MODE == 1) {            // spectrogram row (ShortTermFeatures.py:422)
            double *row = oc + (long long)t * NF;
            for (int k = lane; k < NF; k += kWave) __builtin_nontemporal_store(cur[k], row + k);
        } else if (MODE == 2) {     // chromagram row (:356-359)
            double p = 0.0;
            for (int k = lane; k < NF; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
            p = wsum(p);
#if PAA_TRI_LANE_JOBS
            double chv = chroma_sums_balanced(tb, cur, g_chjob[lane], lane);
            chv = (p == 0.0) ? chv / kEps : fast_div(chv, p);
#else
            const double chv = chroma_class(tb, cur, p, lane);
#endif
            if (lane < 12) oc[(long long)t * 12 + lane] = chv;
        } else if (want) {
            tri_features<SH>(L, tb, tf, cur, (t == 0) ? cur : prv, t == 0, fv, msp, bnd, g_meljob, g_chjob, lane);
            PAA_TICK(7)
            const double v = (lane < kBase) ? fv[lane] : 0.0;
            if (t >= tl.t0 && lane < kBase) {
                row_put(rc, oc + (long long)lane * Tc, t, tl.t0, tend, v);
                if (DELTAS) row_put(rcd, oc + (long long)(kBase + lane) * Tc, t, tl.t0, tend, (t == 0) ? 0.0 : v - vprev);
            }
            vprev = v;
        }
        wsync();
        ++n_done;
        PAA_TICK(10)
    }
#undef PAA_TRI_PACE
    if (lane_id == 0) pace[16 + wave] = 0x7fffffff;
    {
        const int lane = lane_id;
        (void)lane;
        PAA_TEND()
    }
}

// ---- host: shapes, LDS layout + table blob, launch --------------------------------------------------------------------
typedef Shape<20, 20, 3, true, 60, 7> S2400;        // 50 ms at 48 kHz: 1200 complex points
typedef Shape<21, 21, 5, false, 105, 7> S2205;      // 50 ms at 44.1 kHz: 2205 real points (odd window)
typedef Shape<21, 21, 2, true, 42, 8> S1764;        // 40 ms at 44.1 kHz (audioAnalysis.py:71,80): 882 complex points
typedef Shape<20, 16, 3, true, 49, 8> S1920;        // 40 ms at 48 kHz: 960 complex points
typedef Shape<20, 20, 2, true, 40, 8> S1600;        // 50 ms at 32 kHz: 800 complex points
typedef Shape<20, 10, 3, true, 30, 8> S1200;        // 50 ms at 24 kHz / 25 ms at 48 kHz: 600 complex points
typedef Shape<29, 19, 1, false, 19, 12, 1, 3, 3, 16> S551;   // 50 ms at 11.025 kHz / 25 ms at 22.05 kHz: 551 real points, two passes; both prime
                                                         // butterflies shared by three lanes (57 / 45 lanes busy instead of 19 / 15)
typedef Shape<19, 29, 2, false, 58, 12, 2, 1, 3, 16> S1102;  // 25 ms at 44.1 kHz (BASELINE config 5) / 50 ms at 22.05 kHz: 1102 real points;
                                                         // radix 19 first (58 lanes), then radix 29 shared by three lanes (20 jobs: 60 lanes)
// power-of-two windows (what callers outside the reference's 50 ms default pass most often, ShortTermFeatures.py:563-564 takes any
// window): every pass on all 64 lanes where the factorisation allows it.  Plane row pitches from scripts/dev/tri_model.py's LDS
// model (ds_write_b64: 16-lane groups mod 16 doubles, ds_read_b64: 32-lane groups mod 32): the radix-8 shapes use row pitch 72
// and group pitch 9 in the second exchange -- the two exchanges then cost 48 + 80 LDS cycles per plane against 48 + 64 conflict-free
// (row pitch 64, group pitch 8: 312; 69 / 8, the first version: 176, measured conflict ratio 0.36); P = 68 is the best pitch of 16 x 16 x 4
#ifndef PAA_NW_1024
#define PAA_NW_1024 12              // (A/B builds of scripts/rounds/r05: 8 / 10 / 11 waves per workgroup; 12 where the tables leave room)
#endif
typedef Shape<8, 8, 8, true, 72, PAA_NW_1024, 9, 1, 1, 16> S1024;          // 512 complex points: 64 x radix 8, three times
typedef Shape<16, 16, 4, true, 68, 7, 4, 1, 1, 16> S2048;   // 2048 samples: 1024 complex points; with the folded pass 3 the row instances need 108 registers: sixteen waves per CU
typedef Shape<4, 8, 8, true, 72, 12, 9, 1, 1, 16> S512;           // 256 complex points (odd entropy blocks: 51 samples)
// 16 ms at 16 kHz (round 6; st_mix until then: 1.4e8 frames/s): 128 complex points = 4 x 4 x 8 -- half of the lanes carry a pass-1 / pass-2 job,
// nine a pair of radix-8 pass-3 jobs; what a frame costs is the time-domain and feature stages, which are on all 64 lanes
typedef Shape<4, 4, 8, true, 40, 12, 9, 1, 1, 16> S256;

struct TriLaunch {
    int shape = -1;                 // index into the shape list above
    int waves = 0;
    size_t lds = 0;
    const char *name = "";
    TriLayout layout;
};

inline int tri_shape_of(int window) {
    switch (window) {
        case 2400: return 0;
        case 2205: return 1;
        case 1764: return 2;
        case 1920: return 3;
        case 1600: return 4;
        case 1200: return 5;
        case 551: return 6;
        case 1102: return 7;
        case 1024: return 8;
        case 2048: return 9;
        case 512: return 10;
        case 256: return 11;
        default: return -1;
    }
}
// PAA_TRI_SHAPE(SH) is expanded once per shape, in tri_shape_of's order
#define PAA_TRI_SHAPES(X) X(0, S2400) X(1, S2205) X(2, S1764) X(3, S1920) X(4, S1600) X(5, S1200) X(6, S551) X(7, S1102) \
    X(8, S1024) X(9, S2048) X(10, S512) X(11, S256)

template <typename SH>
inline void tri_fill(double fs, int mode, const MelTable *mel, const ChromaTable *chroma, TriLaunch &tl, std::vector<unsigned char> &blob) {
    constexpr int R1 = SH::R1, R2 = SH::R2, R3 = SH::R3, N = SH::N, L1 = SH::L1, NQ1 = SH::NQ1, NR3 = SH::NR3;
    TriLayout &L = tl.layout;
    memset(&L, 0, sizeof(L));
    const size_t n_melw = mel ? mel->w.size() : 0, n_ch = chroma ? chroma->src.size() : 0;
    int off = 0;
    auto take = [&off](size_t bytes) { const int o = off; off += (int)((bytes + 15) / 16 * 16); return o; };
    L.off_tw2 = take((size_t)R2 * R3 * 16);
    L.off_p3 = take(R3 == 1 ? (size_t)R2 * 64 * 2 : (size_t)64 * NR3 * 16 * SH::PE);
    L.off_mello = take(40 * 4);
    L.off_melcnt = take(40 * 4);
    L.off_meloff = take(40 * 4);
    L.off_melw = take(std::max<size_t>(n_melw, 1) * 8);
    L.off_dct = take(13 * 41 * 8);
    L.off_chstart = take(13 * 4);
    L.off_chsrc = take(std::max<size_t>(n_ch, 1) * 4);
    L.off_chw = take(std::max<size_t>(n_ch, 1) * 8);
    L.off_split = take((size_t)(8 * SH::H1 + 16 * SH::H2) * 4);
    L.off_sync = take(32 * 4);
    L.table_bytes = off;
    L.off_g_tw1 = take((size_t)NQ1 * L1 * 16);
    L.off_g_post = take(SH::PACKED ? (size_t)64 * NR3 * R3 * 16 : 16);
    L.off_g_meljob = take(64 * 16);
    L.off_g_chjob = take(64 * 16);
    L.total_bytes = off;
    L.f0 = fs / (2.0 * (double)SH::NF);
    L.rf0 = 1.0 / L.f0;
    L.r_half_fs = 1.0 / (fs / 2.0);
    L.f0sq = L.f0 * L.f0;
    blob.assign((size_t)L.total_bytes, 0);
    unsigned char *b = blob.data();
    const long double two_pi = 6.283185307179586476925286766559005768L;
    auto put_w = [&](int o, size_t idx, long long num, long long den) {      // exp(-2 pi i num / den)
        const long double ang = -two_pi * (long double)(num % den) / (long double)den;
        double *d = reinterpret_cast<double *>(b + o) + 2 * idx;
        d[0] = (double)cosl(ang);
        d[1] = (double)sinl(ang);
    };
    for (int q2 = 0; q2 < R2; ++q2)
        for (int bb = 0; bb < R3; ++bb) put_w(L.off_tw2, (size_t)q2 * R3 + bb, (long long)bb * q2, L1);
    for (int q1 = 0; q1 < NQ1; ++q1)
        for (int j = 0; j < L1; ++j) put_w(L.off_g_tw1, (size_t)q1 * L1 + j, (long long)j * q1, N);
    if (SH::PACKED) {
        // pass-3 lane jobs: unordered pairs {job A, job B} with Z[N - k] of A's outputs in B (same enumeration as pair_count)
        unsigned short *pt = reinterpret_cast<unsigned short *>(b + L.off_p3);
        constexpr int E = 8 * SH::PE, NK = (E - 2) / 2;            // ushorts per entry, (bin, mirror bin) slots of an entry
        int p = 0;
        for (int q1 = 0; q1 < R1; ++q1)
            for (int q2 = 0; q2 < R2; ++q2) {
                const int k = q1 + R1 * q2, m = (N - k) % N;
                const int p1 = m % R1, p2 = (m / R1) % R2;
                if (!(p1 > q1 || (p1 == q1 && p2 >= q2))) continue;
                const bool self = (p1 == q1 && p2 == q2);
                if (SH::FOLD && self && k != 0) continue;          // the second self-paired job rides in entry 0 (below)
                if (SH::FOLD && k == 0) {
                    // entry 0: job 0 (A) + the self-paired job kf = R1 R2 / 2 (B); slots: B's pair (kf, N - kf), A's (N3, N - N3),
                    // A's 2 N3 alone, B's pair (kf + N3, N - kf - N3); bin 0 is written by the kernel itself
                    const int kf = R1 * R2 / 2, N3h = R1 * R2;
                    pt[0] = 0;
                    pt[1] = (unsigned short)(8 * ((kf % R1) * SH::P + (kf / R1) * SH::R3P));
                    const int bins[4][2] = {{kf, N - kf}, {N3h, N - N3h}, {2 * N3h, SH::NF}, {kf + N3h, N - kf - N3h}};
                    for (int k3 = 0; k3 < NK; ++k3) {
                        pt[2 + 2 * k3] = (unsigned short)(8 * (k3 < 4 ? bins[k3][0] : SH::NF));
                        pt[3 + 2 * k3] = (unsigned short)(8 * (k3 < 4 ? bins[k3][1] : SH::NF));
                    }
                    for (int k3 = 0; k3 < 4; ++k3) put_w(L.off_g_post, (size_t)k3, (long long)bins[k3][0], 2LL * N);
                    ++p;
                    continue;
                }
                pt[E * p] = (unsigned short)(8 * (q1 * SH::P + q2 * SH::R3P));
                pt[E * p + 1] = (unsigned short)(8 * (p1 * SH::P + p2 * SH::R3P));
                for (int k3 = 0; k3 < NK; ++k3) {
                    const int kk = k + R1 * R2 * k3;
                    const bool put_k = k3 < R3 && (!self || 2 * kk <= N);                    // X[k]
                    const bool put_m = k3 < R3 && kk != 0 && (!self || 2 * kk < N);          // X[N - k]
                    pt[E * p + 2 + 2 * k3] = (unsigned short)(8 * (put_k ? kk : SH::NF));
                    pt[E * p + 3 + 2 * k3] = (unsigned short)(8 * (put_m ? N - kk : SH::NF));
                }
                for (int k3 = 0; k3 < R3; ++k3) put_w(L.off_g_post, (size_t)p * R3 + k3, (long long)k + (long long)R1 * R2 * k3, 2LL * N);
                ++p;
            }
        // idle lanes of the last round: valid plane offsets (0), every result parked
        for (; p < 64 * NR3; ++p)
            for (int k3 = 0; k3 < NK; ++k3) pt[E * p + 2 + 2 * k3] = pt[E * p + 3 + 2 * k3] = (unsigned short)(8 * SH::NF);
    }
    if (!SH::PACKED) {
        // real input: |Z[k]| goes to bin k, or to its mirror N - k, or nowhere (the parking double at index NF)
        constexpr int NF = SH::NF;
        auto where = [&](int q1, int k) { return 8 * ((k < NF) ? k : (q1 > 0 && N - k < NF) ? N - k : NF); };
        unsigned short *pt = reinterpret_cast<unsigned short *>(b + L.off_p3);
        // split prime passes (Shape::H1 / H2): per part and output slot the true index, conjugate flag, and whether this part is the
        // one that delivers it (the first part that produces an index)
        int *sp = reinterpret_cast<int *>(b + L.off_split);
        if constexpr (SH::H1 > 1) {
            typedef SplitSel<R1, SH::H1> S1;
            std::vector<char> seen(R1, 0);
            for (int h = 0; h < SH::H1; ++h) {
                sp[8 * h] = 0 | (h == 0 ? 0x200 : 0);
                for (int i = 0; i < S1::NQ; ++i) {
                    const int gq = (S1::g(h) * S1::q(i)) % R1;
                    const int t = gq <= R1 / 2 ? gq : R1 - gq;
                    sp[8 * h + 1 + i] = t | (gq > R1 / 2 ? 0x100 : 0) | (seen[t] ? 0 : 0x200);
                    seen[t] = 1;
                }
            }
        }
        if constexpr (SH::H2 > 1) {
            typedef SplitSel<R2, SH::H2> S2;
            int *sp2 = sp + 8 * SH::H1;
            std::vector<char> seen(R2, 0);
            for (int h = 0; h < SH::H2; ++h) {
                sp2[16 * h] = 0 | (h == 0 ? 0x200 : 0);
                for (int i = 0; i < S2::NQ; ++i) {
                    const int tp = (S2::g(h) * S2::q(i)) % R2, tm = R2 - tp;
                    sp2[16 * h + 1 + 2 * i] = tp | (seen[tp] ? 0 : 0x200);
                    sp2[16 * h + 2 + 2 * i] = tm | (seen[tm] ? 0 : 0x200);
                    seen[tp] = seen[tm] = 1;
                }
            }
        }
        if (R3 == 1) {
            if constexpr (SH::H2 > 1) {
                // where slot s of lane (part, q1) puts its magnitude: the bin of Z[q1 + R1 t], its mirror, or the parking double
                const int *sp2 = sp + 8 * SH::H1;
                constexpr int NS2 = 1 + 2 * SplitNQ<R2, SH::H2>::value;
                for (int sl = 0; sl < NS2; ++sl)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int part = lane / SH::J2, q1 = lane % SH::J2;
                        const bool on = part < SH::H2 && (sp2[16 * (part < SH::H2 ? part : 0) + sl] & 0x200);
                        const int t = sp2[16 * (part < SH::H2 ? part : 0) + sl] & 0xff;
                        pt[sl * 64 + lane] = (unsigned short)(on ? where(q1, q1 + R1 * t) : 8 * NF);
                    }
            } else {
            for (int q2 = 0; q2 < R2; ++q2)
                for (int lane = 0; lane < 64; ++lane)
                    pt[q2 * 64 + lane] = (unsigned short)(lane < NQ1 ? where(lane, lane + R1 * q2) : 8 * NF);
            }
        } else {
            for (int m3 = 0; m3 < 64 * NR3; ++m3) {
                const int q1 = m3 / R2, q2 = m3 % R2;
                const bool act = m3 < SH::NJOB3;
                pt[8 * m3] = (unsigned short)(act ? 8 * (q1 * SH::P + q2 * SH::R3P) : 0);
                for (int k3 = 0; k3 < 7; ++k3)
                    pt[8 * m3 + 1 + k3] = (unsigned short)((act && k3 < R3) ? where(q1, q1 + R1 * (q2 + R2 * k3)) : 8 * NF);
            }
        }
    }
    if (mel && !mel->w.empty()) {
        memcpy(b + L.off_mello, mel->lo.data(), 40 * 4);
        memcpy(b + L.off_melcnt, mel->cnt.data(), 40 * 4);
        memcpy(b + L.off_meloff, mel->off.data(), 40 * 4);
        memcpy(b + L.off_melw, mel->w.data(), n_melw * 8);
        double dct[kNumMfcc * kNumMel];
        build_dct(dct);
        double *d = reinterpret_cast<double *>(b + L.off_dct);
        for (int q = 0; q < 13; ++q)
            for (int n = 0; n < 40; ++n) d[q * 41 + n] = dct[q * 40 + n];
    }
    if (chroma && !chroma->src.empty()) {
        memcpy(b + L.off_chstart, chroma->class_start, 13 * 4);
        memcpy(b + L.off_chsrc, chroma->src.data(), n_ch * 4);
        memcpy(b + L.off_chw, chroma->w.data(), n_ch * 8);
        int first[12], cnt[12];
        for (int c = 0; c < 12; ++c) { first[c] = chroma->class_start[c]; cnt[c] = chroma->class_start[c + 1] - chroma->class_start[c]; }
        lane_jobs(12, first, first, cnt, reinterpret_cast<LaneJob *>(b + L.off_g_chjob));
    }
    if (mel && !mel->w.empty())
        lane_jobs(40, mel->lo.data(), mel->off.data(), mel->cnt.data(), reinterpret_cast<LaneJob *>(b + L.off_g_meljob));
    // as many waves as the shape allows and the LDS holds beside this (fs, window)'s table blob (longer mel lists at low rates)
    tl.waves = (mode == 0) ? SH::NW : SH::NWR;
    const size_t wave_bytes = (size_t)(mode == 0 ? SH::WAVE_DOUBLES : SH::WAVE_DOUBLES_ROWS) * 8;
    while (tl.waves > 4 && (size_t)L.table_bytes + (size_t)tl.waves * wave_bytes > 160 * 1024) --tl.waves;
    L.waves = tl.waves;
    tl.lds = (size_t)L.table_bytes + (size_t)tl.waves * wave_bytes;
}

// returns 1 when a three-pass instance exists for this window (fills tl and the table blob), 0 otherwise
inline int tri_select(int window, int mode, double fs, const MelTable *mel, const ChromaTable *chroma, TriLaunch &tl,
                      std::vector<unsigned char> &blob) {
    const int sh = tri_shape_of(window);
    if (sh < 0) return 0;
    tl.shape = sh;
    static const char *names[3][12] = {
        {"st_tri_20x20x3", "st_tri_r21x21x5", "st_tri_21x21x2", "st_tri_20x16x3", "st_tri_20x20x2", "st_tri_20x10x3", "st_tri_r29x19",
         "st_tri_r19x29x2", "st_tri_8x8x8", "st_tri_16x16x4", "st_tri_4x8x8", "st_tri_4x4x8"},
        {"spectrogram_tri_20x20x3", "spectrogram_tri_r21x21x5", "spectrogram_tri_21x21x2", "spectrogram_tri_20x16x3",
         "spectrogram_tri_20x20x2", "spectrogram_tri_20x10x3", "spectrogram_tri_r29x19", "spectrogram_tri_r19x29x2",
         "spectrogram_tri_8x8x8", "spectrogram_tri_16x16x4", "spectrogram_tri_4x8x8", "spectrogram_tri_4x4x8"},
        {"chromagram_tri_20x20x3", "chromagram_tri_r21x21x5", "chromagram_tri_21x21x2", "chromagram_tri_20x16x3",
         "chromagram_tri_20x20x2", "chromagram_tri_20x10x3", "chromagram_tri_r29x19", "chromagram_tri_r19x29x2",
         "chromagram_tri_8x8x8", "chromagram_tri_16x16x4", "chromagram_tri_4x8x8", "chromagram_tri_4x4x8"}};
    tl.name = names[mode][sh];
    switch (sh) {
#define PAA_TRI_FILL(ID, SH) case ID: tri_fill<SH>(fs, mode, mel, chroma, tl, blob); break;
        PAA_TRI_SHAPES(PAA_TRI_FILL)
#undef PAA_TRI_FILL
        default: return 0;
    }
    if (tl.lds > 160 * 1024) return 0;
    return 1;
}

#if !defined(PAA_NO_HOST_LAUNCHERS) || defined(PAA_LAUNCH_TRI)      // (kernels are instantiated only in family_tri*.hip)
template <typename SH, typename T, int MODE, int DELTAS>
static inline int tri_launch_one(const TriLaunch &tl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                          const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                          hipStream_t stream) {
    static LdsAttrCache attr;
    if (!attr.covers(tl.lds)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&st_tri_kernel<SH, T, MODE, DELTAS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl.lds) != hipSuccess) return -1;
        attr.set(tl.lds);
    }
    const int nwm = tl.waves;           // (<= the shape's maximum, which is what __launch_bounds__ promises)
    const unsigned grid = (unsigned)((n_tiles + nwm - 1) / nwm);
    hipLaunchKernelGGL((st_tri_kernel<SH, T, MODE, DELTAS>), dim3(grid), dim3(64 * nwm), tl.lds, stream, P, tl.layout, blob,
                       (const T *)d_packed, clips, norms, tiles, (int)n_tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <typename SH, typename T>
static inline int tri_launch_mode(const TriLaunch &tl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                           const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                           hipStream_t stream) {
    if (P.mode == 1) return tri_launch_one<SH, T, 1, 0>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (P.mode == 2) return tri_launch_one<SH, T, 2, 0>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (P.deltas) return tri_launch_one<SH, T, 0, 1>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return tri_launch_one<SH, T, 0, 0>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
// (internal linkage: the two units that instantiate these launchers name different shapes in PAA_TRI_SHAPES_HERE)
template <typename T>
static inline int tri_launch_shape(const TriLaunch &tl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                            const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                            hipStream_t stream) {
    // (the shapes are spread over two translation units: PAA_TRI_SHAPES_HERE names the ones this unit instantiates)
#ifndef PAA_TRI_SHAPES_HERE
#define PAA_TRI_SHAPES_HERE(X) PAA_TRI_SHAPES(X)
#endif
    switch (tl.shape) {
#define PAA_TRI_GO(ID, SH) case ID: return tri_launch_mode<SH, T>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        PAA_TRI_SHAPES_HERE(PAA_TRI_GO)
#undef PAA_TRI_GO
        default: return -1;
    }
}
// sample_kind 0: int16, 1: float64, 2: interleaved stereo int16 (summed in the loads)
static inline int tri_launch(const TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                      const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                      hipStream_t stream) {
    if (sample_kind == 0) return tri_launch_shape<int16_t>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (sample_kind == 2) return tri_launch_shape<stereo16>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return tri_launch_shape<double>(tl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}

#endif  // PAA_NO_HOST_LAUNCHERS

}  // namespace tri
}  // namespace paa

// nfm_reduce_median.hip -- median of every row of a contiguous (rows, red) array by radix
// selection: `median` of the reference (`reduce.py:384-428`, which defers to torch.median after
// moving the reduced dims last -- the facade does the same move).
//
// Semantics (torch.median / torch.nanmedian): the LOWER median, i.e. the element of rank
// (count - 1) / 2 in ascending order; omitnan = 0: any NaN in a row makes its median NaN (index:
// the first NaN); omitnan = 1: the median of the non-NaN elements (all NaN: NaN, index 0).
// Index: the first position that holds the median value (bit for bit: -0.0 sorts before +0.0).
//
// A float is mapped to an unsigned key that sorts like the float (sign bit flipped for positives,
// all bits flipped for negatives; NaN -> the largest key, so NaNs sort last and are never selected
// when omitted), and the key of rank k is found digit by digit from the top (8 bits at a time where
// the row sits in registers, 11 where every digit is a pass over HBM):
// count the elements per digit value among those that match the digits chosen so far, keep the
// digit whose cumulative count passes k.  No data movement.
//
//   red <= 128:  (float64: 64) one row per LANE, sorted in registers by a compile-time merge-exchange
//                network; with fewer than 4096 rows, or beyond that length:
//   red <= 32:   8 / 16 / 32 lanes per row, one element per lane, ranks counted directly;
//   red <= 1024: 16 / 32 / 64 lanes per row (4 / 2 / 1 rows per WAVEFRONT), the row in registers (<= 16 keys
//                per lane), a 256-bin LDS histogram per row, ds_add for the counts, a scan inside the
//                group to pick the digit;
//   longer rows: every pass streams the row once with 16-byte loads -- a grid of (chunks, rows)
//                workgroups histograms into LDS and adds its counts to the row's global histogram;
//                a one-wavefront kernel per row picks the digit between passes.  Here a pass costs
//                a trip over HBM, so the digits are 11 bits wide (2048 bins, 8 KiB of LDS): 3 passes
//                for float32 (11 + 11 + 10 bits), 6 for float64 -- a full reduction of 2^33
//                elements is 3 passes at the streaming rate.
#include "nfm_reduce_median.hpp"

namespace nfm {
namespace med {

__device__ __forceinline__ unsigned wave_excl_scan(unsigned v, unsigned &total)
{
    const int lane = threadIdx.x & 63;
    unsigned s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(s, off, 64);
        if (lane >= off) s += o;
    }
    total = __shfl(s, 63, 64);
    return s - v;
}

// pick the digit whose cumulative count passes k: `cnt` = this lane's 4 consecutive bins
// (4 * lane .. 4 * lane + 3); returns the digit and reduces k to the rank inside that digit
__device__ __forceinline__ unsigned pick_digit(const unsigned (&cnt)[4], unsigned long long &k)
{
    const int lane = threadIdx.x & 63;
    unsigned tot;
    const unsigned mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    const unsigned before = wave_excl_scan(mine, tot);
    const bool here = (unsigned long long)before <= k && k < (unsigned long long)before + mine;
    const unsigned long long m = __ballot(here);
    const int src = m ? __builtin_ctzll(m) : 63; // k < total always: exactly one lane
    unsigned digit = 0;
    unsigned long long kk = k;
    if (lane == src) {
        unsigned long long r = k - before;
        unsigned d = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (d == (unsigned)b && r >= cnt[b]) {
                r -= cnt[b];
                d = b + 1;
            }
        digit = 4 * lane + d;
        kk = r;
    }
    digit = __shfl(digit, src, 64);
    const unsigned lo = __shfl((unsigned)kk, src, 64), hi = __shfl((unsigned)(kk >> 32), src, 64);
    k = ((unsigned long long)hi << 32) | lo;
    return digit;
}

// ---------------------------------------------------------------------------------------------
// short rows: one wavefront per 1 / 2 / 4 rows
// the part of a key above digit d (0 for the top digit: nothing chosen yet)
template <typename U>
__device__ __forceinline__ U above(U key, int d, int digits)
{
    return d + 1 < digits ? (U)(key >> (8 * (d + 1))) : U(0);
}

// G lanes per row (16 / 32 / 64: 4 / 2 / 1 rows per wavefront), E keys per lane: rows of up to G * E
// elements.  The per-pass frame (clear the histogram, scan it, pick the digit) is the same instruction
// stream whatever the row length, so short rows share a wavefront: 4 rows of <= 256 elements cost what
// one did.  Every row has its own 256-bin histogram; the scan and the broadcasts run inside the group.
template <typename T, int E, int G>
__global__ __launch_bounds__(256) void median_rows_kernel(const T *__restrict__ x, int64_t rows, int red, int omitnan,
                                                          T *__restrict__ val, int64_t *__restrict__ idx)
{
    using K = Key<T>;
    using U = typename K::U;
    constexpr int RPW = 64 / G;  // rows per wavefront
    constexpr int NB = 256 / G;  // histogram bins owned by a lane
    __shared__ unsigned hist_all[4][RPW][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane / G, j = lane % G;
    unsigned *hist = hist_all[w][g];
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * RPW;
    if (row0 >= rows) return; // whole wavefronts leave: no barrier below is workgroup-wide
    const int64_t row = row0 + g;
    const bool live_row = row < rows;
    const T *p = x + (live_row ? row : row0) * red; // groups past the end redo the first row (never stored)
    U key[E];
    unsigned nan = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int jj = j + G * e;
        const T v = jj < red ? NFM_LDG(p + jj) : T(0);
        key[e] = K::of(v);
        nan += (jj < red && v != v) ? 1u : 0u;
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) nan += __shfl_xor(nan, off, G);
    const unsigned count = omitnan ? (unsigned)red - nan : (unsigned)red;
    const bool want_nan = (!omitnan && nan > 0) || count == 0;
    unsigned k = count ? (count - 1) / 2 : 0;
    U prefix = 0;
#pragma unroll
    for (int d = K::digits - 1; d >= 0; --d) {
        const int shift = 8 * d;
#pragma unroll
        for (int b = 0; b < NB; ++b) hist[NB * j + b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int jj = j + G * e;
            // elements that agree with the digits chosen so far (all of them in the first pass)
            const bool in = jj < red && above(key[e], d, K::digits) == prefix;
            if (in) atomicAdd(&hist[(unsigned)(key[e] >> shift) & 255u], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // pick the digit whose cumulative count passes k, inside the group (rows with a NaN result
        // have no such digit: they go through the motions and are overridden below)
        unsigned cnt[NB], mine = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            cnt[b] = hist[NB * j + b];
            mine += cnt[b];
        }
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, G);
            if (j >= off) incl += o;
        }
        const unsigned before = incl - mine;
        const bool here = before <= k && k < before + mine;
        const unsigned long long gm = (__ballot(here) >> (g * G)) & (G == 64 ? ~0ull : ((1ull << (G % 64)) - 1ull));
        const int src = gm ? __builtin_ctzll(gm) : G - 1;
        unsigned digit = 0, kk = k;
        if (j == src) {
            unsigned r = k - before, dd = 0;
#pragma unroll
            for (int b = 0; b < NB - 1; ++b)
                if (dd == (unsigned)b && r >= cnt[b]) {
                    r -= cnt[b];
                    dd = b + 1;
                }
            digit = NB * j + dd;
            kk = r;
        }
        digit = __shfl(digit, src, G);
        k = __shfl(kk, src, G);
        prefix = (prefix << 8) | (U)digit;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const U chosen = want_nan ? ~U(0) : prefix;
    // first position holding the chosen key (a NaN result: the first NaN; all-NaN with omitnan: 0)
    int first = 0x7fffffff;
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        const int jj = j + G * e;
        if (jj < red && key[e] == chosen) first = jj;
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const int o = __shfl_xor(first, off, G);
        first = o < first ? o : first;
    }
    if (j == 0 && live_row) {
        val[row] = want_nan ? (T)__builtin_nanf("") : K::back(chosen);
        if (idx) idx[row] = first == 0x7fffffff ? 0 : first;
    }
}

// ---------------------------------------------------------------------------------------------
// tiny rows (red <= 32): G = 8 / 16 / 32 lanes per row, 64 / G rows per wavefront, one element per
// lane; the rank of an element is counted directly -- #smaller + #equal at a lower position --
// with one group broadcast per position, and the element of rank k is the median.
template <typename T, int G>
__global__ __launch_bounds__(256) void median_tiny_kernel(const T *__restrict__ x, int64_t rows, int red, int omitnan,
                                                          T *__restrict__ val, int64_t *__restrict__ idx)
{
    using K = Key<T>;
    using U = typename K::U;
    constexpr int RPW = 64 / G; // rows per wavefront
    const int lane = threadIdx.x & 63, j = lane % G, gbase = lane - j;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
    const bool live = row < rows && j < red;
    const T v = live ? NFM_LDG(x + row * red + j) : T(0);
    const U key = live ? K::of(v) : ~U(0);
    unsigned nan = (live && v != v) ? 1u : 0u;
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) nan += __shfl_xor(nan, off, 64);
    const unsigned count = omitnan ? (unsigned)red - nan : (unsigned)red;
    const bool want_nan = (!omitnan && nan > 0) || count == 0;
    const unsigned k = count ? (count - 1) / 2 : 0;
    unsigned rank = 0;
    for (int t = 0; t < red; ++t) { // red is uniform; positions beyond it hold nothing
        U kt;
        if constexpr (sizeof(U) == 4) kt = (U)__shfl((unsigned)key, gbase + t, 64);
        else {
            const unsigned lo = __shfl((unsigned)key, gbase + t, 64), hi = __shfl((unsigned)(key >> 32), gbase + t, 64);
            kt = ((U)hi << 32) | lo;
        }
        rank += (kt < key || (kt == key && t < j)) ? 1u : 0u;
    }
    // exactly one live lane of the row has rank k (NaN keys are the largest: never rank k when omitted)
    const unsigned long long hit = __ballot(live && !want_nan && rank == k);
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gbase;
    const int src = (hit & gmask) ? __builtin_ctzll(hit & gmask) : gbase;
    U chosen;
    if constexpr (sizeof(U) == 4) chosen = (U)__shfl((unsigned)key, src, 64);
    else {
        const unsigned lo = __shfl((unsigned)key, src, 64), hi = __shfl((unsigned)(key >> 32), src, 64);
        chosen = ((U)hi << 32) | lo;
    }
    if (want_nan) chosen = ~U(0);
    const unsigned long long same = __ballot(live && key == chosen) & gmask;
    if (j == 0 && row < rows) {
        val[row] = want_nan ? (T)__builtin_nanf("") : K::back(chosen);
        if (idx) idx[row] = same ? (int64_t)(__builtin_ctzll(same) - gbase) : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// long rows: 11-bit digits.  Pass p (0 = most significant) covers key bits [shift, shift + width).
constexpr int kLongBits = 11, kLongBins = 1 << kLongBits;
template <typename T>
struct LongDigits {
    static constexpr int bits = 8 * (int)sizeof(T);
    static constexpr int passes = (bits + kLongBits - 1) / kLongBits;
    static __host__ __device__ constexpr int width(int p) { return p + 1 < passes ? kLongBits : bits - kLongBits * (passes - 1); }
    static __host__ __device__ constexpr int shift(int p) { return p + 1 < passes ? bits - kLongBits * (p + 1) : 0; }
};

__device__ __forceinline__ unsigned long long wave_excl_scan64(unsigned long long v)
{
    const int lane = threadIdx.x & 63;
    unsigned long long s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned lo = __shfl_up((unsigned)s, off, 64), hi = __shfl_up((unsigned)(s >> 32), off, 64);
        if (lane >= off) s += ((unsigned long long)hi << 32) | lo;
    }
    return s - v;
}

// pick_digit for NB consecutive bins per lane (64 * NB bins) of a LONG row: counts in 64 bits (a single row
// can hold 2^33 elements -- the dim=None median of a 32 GiB tensor -- and constant data puts them all in one bin)
template <int NB>
__device__ __forceinline__ unsigned pick_digit_n(const unsigned long long (&cnt)[NB], unsigned long long &k)
{
    const int lane = threadIdx.x & 63;
    unsigned long long mine = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) mine += cnt[b];
    const unsigned long long before = wave_excl_scan64(mine);
    const bool here = before <= k && k < before + mine;
    const unsigned long long m = __ballot(here);
    const int src = m ? __builtin_ctzll(m) : 63; // k < total always: exactly one lane
    unsigned digit = 0;
    unsigned long long kk = k;
    if (lane == src) {
        unsigned long long r = k - before;
        unsigned d = 0;
#pragma unroll
        for (int b = 0; b < NB - 1; ++b)
            if (d == (unsigned)b && r >= cnt[b]) {
                r -= cnt[b];
                d = b + 1;
            }
        digit = NB * lane + d;
        kk = r;
    }
    digit = __shfl(digit, src, 64);
    const unsigned lo = __shfl((unsigned)kk, src, 64), hi = __shfl((unsigned)(kk >> 32), src, 64);
    k = ((unsigned long long)hi << 32) | lo;
    return digit;
}

// per-row state in the workspace
struct RowState {
    unsigned long long prefix; // digits chosen so far
    unsigned long long k;      // rank still to resolve inside the prefix
    unsigned long long nan;    // NaNs in the row (counted by the first pass)
    unsigned long long first;  // first position of the result
    int want_nan;              // the result is NaN
    int pad;
};
static_assert(sizeof(RowState) == 40, "workspace layout");

template <typename T>
__global__ __launch_bounds__(256) void median_hist_kernel(const T *__restrict__ x, int64_t red, int64_t chunk, int pass,
                                                          const RowState *__restrict__ st,
                                                          unsigned long long *__restrict__ ghist,
                                                          unsigned long long *__restrict__ gnan)
{
    using K = Key<T>;
    using U = typename K::U;
    using D = LongDigits<T>;
    using V = typename VecOf<T>::gtype;
    constexpr int NV = VecOf<T>::N;
    __shared__ unsigned hist[kLongBins];
    __shared__ unsigned nan_s;
    const int64_t row = blockIdx.y;
    const bool first_pass = pass == 0;
    if (!first_pass && st[row].want_nan) return;
#pragma unroll
    for (int b = 0; b < kLongBins / 256; ++b) hist[threadIdx.x + 256 * b] = 0;
    if (threadIdx.x == 0) nan_s = 0;
    __syncthreads();
    const U prefix = first_pass ? U(0) : (U)st[row].prefix;
    const int shift = D::shift(pass), width = D::width(pass);
    const int up = first_pass ? 0 : shift + width; // bits above this digit: chosen already (never the full width)
    const unsigned mask = (1u << width) - 1u;
    const T *p = x + row * red;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < red ? lo + chunk : red;
    unsigned nan = 0;
    auto take = [&](T v) {
        const U key = K::of(v);
        if (first_pass) nan += (v != v) ? 1u : 0u;
        if (first_pass || (U)(key >> up) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & mask], 1u);
    };
    // 16-byte loads over the aligned middle of the chunk, elements at its ragged ends
    int64_t j = lo + threadIdx.x;
    const int64_t mis = (int64_t)((reinterpret_cast<uintptr_t>(p + lo) / sizeof(T)) % NV);
    const int64_t head = mis ? (NV - mis < hi - lo ? NV - mis : hi - lo) : 0;
    if (j < lo + head) take(NFM_LDG(p + j));
    const int64_t body0 = lo + head, nvec = (hi - body0) / NV;
    for (int64_t q = threadIdx.x; q < nvec; q += 256) {
        const V v = NFM_LDG(reinterpret_cast<const V *>(p + body0) + q);
#pragma unroll
        for (int c = 0; c < NV; ++c) take(v[c]);
    }
    j = body0 + nvec * NV + threadIdx.x;
    if (j < hi) take(NFM_LDG(p + j));
    if (first_pass) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nan += __shfl_xor(nan, off, 64);
        if ((threadIdx.x & 63) == 0 && nan) atomicAdd(&nan_s, nan);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kLongBins / 256; ++b) {
        const unsigned c = hist[threadIdx.x + 256 * b];
        if (c) atomicAdd(&ghist[row * kLongBins + threadIdx.x + 256 * b], (unsigned long long)c); // (a block's chunk < 2^32)
    }
    if (first_pass && threadIdx.x == 0 && nan_s) atomicAdd(&gnan[row], (unsigned long long)nan_s);
}

// one wavefront per row: choose the digit of this pass, clear the histogram for the next one
template <typename T>
__global__ __launch_bounds__(64) void median_pick_kernel(int64_t red, int pass, int omitnan, RowState *__restrict__ st,
                                                         unsigned long long *__restrict__ ghist,
                                                         unsigned long long *__restrict__ gnan, T *__restrict__ val)
{
    using K = Key<T>;
    using U = typename K::U;
    using D = LongDigits<T>;
    constexpr int NB = kLongBins / 64;
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x;
    RowState s = st[row];
    if (pass == 0) {
        s.prefix = 0;
        s.nan = gnan[row];
        const unsigned long long count = omitnan ? (unsigned long long)red - s.nan : (unsigned long long)red;
        s.want_nan = ((!omitnan && s.nan > 0) || count == 0) ? 1 : 0;
        s.k = count ? (count - 1) / 2 : 0;
        s.first = red; // "not found yet" for the index pass
    }
    unsigned long long *gh = ghist + row * kLongBins + NB * lane;
    if (!s.want_nan) {
        unsigned long long cnt[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) cnt[b] = gh[b];
        unsigned long long k = s.k;
        const unsigned digit = pick_digit_n<NB>(cnt, k);
        s.k = k;
        s.prefix = (s.prefix << D::width(pass)) | digit;
    } else {
        s.prefix = ~0ull;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) gh[b] = 0;
    if (lane == 0) {
        st[row] = s;
        if (pass == D::passes - 1) val[row] = s.want_nan ? (T)__builtin_nanf("") : K::back((U)s.prefix);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void median_index_kernel(const T *__restrict__ x, int64_t red, int64_t chunk,
                                                           RowState *__restrict__ st)
{
    using K = Key<T>;
    using U = typename K::U;
    const int64_t row = blockIdx.y;
    const U chosen = st[row].want_nan ? ~U(0) : (U)st[row].prefix;
    const T *p = x + row * red;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < red ? lo + chunk : red;
    unsigned long long first = ~0ull;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256)
        if (K::of(NFM_LDG(p + j)) == chosen) {
            first = (unsigned long long)j;
            break; // positions grow with j: the first hit of this thread is its smallest
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo32 = __shfl_xor((unsigned)first, off, 64), hi32 = __shfl_xor((unsigned)(first >> 32), off, 64);
        const unsigned long long o = ((unsigned long long)hi32 << 32) | lo32;
        first = o < first ? o : first;
    }
    if ((threadIdx.x & 63) == 0 && first != ~0ull) atomicMin(&st[row].first, first);
}

__global__ void median_store_index_kernel(int64_t rows, int64_t red, const RowState *__restrict__ st,
                                          int64_t *__restrict__ idx)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row < rows) idx[row] = st[row].first >= (unsigned long long)red ? 0 : (int64_t)st[row].first;
}

constexpr int kShortMax = 1024;
constexpr int64_t kLaneMinRows = 4096; // below this the rows do not fill the lanes of the chip: the group kernel
constexpr int64_t kLanePadMinRows = 32768; // padded rows: one wavefront per SIMD (256+ registers) -- 64 Ki lanes fill the chip once

static int64_t chunk_of(int64_t rows, int64_t red)
{
    // ~2048 workgroups in flight, chunks of at least 16 Ki elements, a multiple of 1024
    int64_t per_row = (2048 + rows - 1) / rows;
    if (per_row < 1) per_row = 1;
    int64_t chunk = (red + per_row - 1) / per_row;
    if (chunk < 16384) chunk = 16384;
    return ((chunk + 1023) / 1024) * 1024;
}

// one row per lane (nfm_reduce_median_lane.hip, one object per residue of the length mod 8)
static int lane_any(int dt, int red, int omitnan, int64_t rows, int64_t inner, const void *x, void *val, void *idx,
                    void *stream)
{
    switch (red % kLaneParts) {
    case 0: return lane_part0(dt, red, omitnan, rows, inner, x, val, idx, stream);
    case 1: return lane_part1(dt, red, omitnan, rows, inner, x, val, idx, stream);
    case 2: return lane_part2(dt, red, omitnan, rows, inner, x, val, idx, stream);
    case 3: return lane_part3(dt, red, omitnan, rows, inner, x, val, idx, stream);
    case 4: return lane_part4(dt, red, omitnan, rows, inner, x, val, idx, stream);
    case 5: return lane_part5(dt, red, omitnan, rows, inner, x, val, idx, stream);
    case 6: return lane_part6(dt, red, omitnan, rows, inner, x, val, idx, stream);
    default: return lane_part7(dt, red, omitnan, rows, inner, x, val, idx, stream);
    }
}

// one PADDED row per lane: lengths LaneMax+1 .. 2 LaneMax, bucket = part (nfm_reduce_median_lane.hip)
template <typename T>
static int lane_pad_any(int red, int omitnan, int64_t rows, const void *x, void *val, void *idx, void *stream)
{
    constexpr int step = LaneMax<T>::value / kLaneParts; // 16 / 8
    const int dt = sizeof(T) == 4 ? NFM_F32 : NFM_F64;
    if (red > LanePadMax<T>::value) // four lanes per row: part 5 (float32) / part 1 (float64)
        return sizeof(T) == 4 ? lane_part5(dt, -red, omitnan, rows, 1, x, val, idx, stream)
                              : lane_part1(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    switch ((red - LaneMax<T>::value - 1) / step + LanePadBuckets<T>::first_part) {
    case 0: return lane_part0(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    case 1: return lane_part1(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    case 2: return lane_part2(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    case 3: return lane_part3(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    case 4: return lane_part4(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    case 5: return lane_part5(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    case 6: return lane_part6(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    default: return lane_part7(dt, -red, omitnan, rows, 1, x, val, idx, stream);
    }
}

template <typename T>
static int run(int omitnan, int64_t rows, int64_t red, const void *x, void *ws, size_t ws_bytes, void *val, void *idx,
               void *stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (red >= 2 && red <= LaneMax<T>::value && rows >= kLaneMinRows && reinterpret_cast<uintptr_t>(x) % sizeof(T) == 0)
        return lane_any(sizeof(T) == 4 ? NFM_F32 : NFM_F64, (int)red, omitnan, rows, 1, x, val, idx, stream);
    if (red > LaneMax<T>::value && red <= 2 * LaneMax<T>::value && rows >= kLanePadMinRows &&
        reinterpret_cast<uintptr_t>(x) % sizeof(T) == 0)
        return lane_pad_any<T>((int)red, omitnan, rows, x, val, idx, stream);
    if (red <= 32) {
        const int G = red <= 8 ? 8 : (red <= 16 ? 16 : 32);
        const int64_t nblk = (rows + 4 * (64 / G) - 1) / (4 * (64 / G));
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
#define NFM_MED_TINY(Gv)                                                                                              \
    hipLaunchKernelGGL((median_tiny_kernel<T, Gv>), dim3((unsigned)nblk), dim3(256), 0, s, static_cast<const T *>(x), \
                       rows, (int)red, omitnan, static_cast<T *>(val), static_cast<int64_t *>(idx))
        if (G == 8) NFM_MED_TINY(8);
        else if (G == 16) NFM_MED_TINY(16);
        else NFM_MED_TINY(32);
#undef NFM_MED_TINY
        return launch_status();
    }
    if (red <= kShortMax) {
#define NFM_MED_ROWS(Ev, Gv)                                                                                          \
    {                                                                                                                 \
        const int64_t nblk = (rows + 4 * (64 / Gv) - 1) / (4 * (64 / Gv));                                            \
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;                                                                    \
        hipLaunchKernelGGL((median_rows_kernel<T, Ev, Gv>), dim3((unsigned)nblk), dim3(256), 0, s,                    \
                           static_cast<const T *>(x), rows, (int)red, omitnan, static_cast<T *>(val),                 \
                           static_cast<int64_t *>(idx));                                                              \
    }
        if (red <= 64) NFM_MED_ROWS(4, 16)
        else if (red <= 128) NFM_MED_ROWS(8, 16)
        else if (red <= 256) NFM_MED_ROWS(16, 16)
        else if (red <= 512) NFM_MED_ROWS(16, 32)
        else NFM_MED_ROWS(16, 64)
#undef NFM_MED_ROWS
        return launch_status();
    }
    if (rows > 65535) return NFM_ESIZE; // grid.y; the facade splits (long rows are few)
    const size_t need = (size_t)rows * (sizeof(RowState) + kLongBins * sizeof(unsigned long long) + sizeof(unsigned long long));
    if (ws == nullptr || ws_bytes < need) return NFM_EINVAL;
    RowState *st = static_cast<RowState *>(ws);
    unsigned long long *gnan = reinterpret_cast<unsigned long long *>(st + rows);
    unsigned long long *ghist = gnan + rows;
    hipError_t e = hipMemsetAsync(ws, 0, need, s);
    if (e != hipSuccess) return (int)e;
    const int64_t chunk = chunk_of(rows, red);
    const int64_t nch = (red + chunk - 1) / chunk;
    if (nch > 0x7fffffffLL) return NFM_ESIZE;
    for (int pass = 0; pass < LongDigits<T>::passes; ++pass) {
        hipLaunchKernelGGL((median_hist_kernel<T>), dim3((unsigned)nch, (unsigned)rows), dim3(256), 0, s,
                           static_cast<const T *>(x), red, chunk, pass, st, ghist, gnan);
        hipLaunchKernelGGL((median_pick_kernel<T>), dim3((unsigned)rows), dim3(64), 0, s, red, pass, omitnan, st, ghist,
                           gnan, static_cast<T *>(val));
    }
    if (idx) {
        hipLaunchKernelGGL((median_index_kernel<T>), dim3((unsigned)nch, (unsigned)rows), dim3(256), 0, s,
                           static_cast<const T *>(x), red, chunk, st);
        hipLaunchKernelGGL(median_store_index_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, rows, red, st,
                           static_cast<int64_t *>(idx));
    }
    return launch_status();
}

} // namespace med
} // namespace nfm

using namespace nfm;

extern "C" {

size_t nfm_reduce_median_workspace_bytes(int64_t rows, int64_t red)
{
    if (rows <= 0 || red <= med::kShortMax) return 0;
    return (size_t)rows * (sizeof(med::RowState) + med::kLongBins * sizeof(unsigned long long) + sizeof(unsigned long long));
}

int nfm_reduce_median_lane_max(int dtype)
{
    return dtype == NFM_F32 ? med::LaneMax<float>::value : (dtype == NFM_F64 ? med::LaneMax<double>::value : 0);
}

int nfm_reduce_median_mid(int dtype, int omitnan, int64_t outer, int64_t red, int64_t inner, const void *x, void *val,
                          void *idx, void *stream)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (outer < 0 || red < 0 || inner < 0) return NFM_EINVAL;
    if (outer == 0 || inner == 0) return NFM_OK;
    if (red < 2 || red > nfm_reduce_median_lane_max(dtype)) return NFM_ESIZE;
    if (x == nullptr || val == nullptr) return NFM_EINVAL;
    if (reinterpret_cast<uintptr_t>(x) % (dtype == NFM_F32 ? 4 : 8) != 0) return NFM_EALIGN;
    return med::lane_any(dtype, (int)red, omitnan ? 1 : 0, outer * inner, inner, x, val, idx, stream);
}

int nfm_reduce_median(int dtype, int omitnan, int64_t rows, int64_t red, const void *x, void *workspace,
                      size_t workspace_bytes, void *val, void *idx, void *stream)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (rows < 0 || red < 0) return NFM_EINVAL;
    if (rows == 0) return NFM_OK;
    if (red == 0 || x == nullptr || val == nullptr) return NFM_EINVAL;
    return dtype == NFM_F32 ? med::run<float>(omitnan ? 1 : 0, rows, red, x, workspace, workspace_bytes, val, idx, stream)
                            : med::run<double>(omitnan ? 1 : 0, rows, red, x, workspace, workspace_bytes, val, idx, stream);
}

} // extern "C"

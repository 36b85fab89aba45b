// nfm_reduce_median_lane.hip -- median of short rows, one row per lane (see nfm_reduce_median.hip for
// the other regimes and the semantics).  Every row length is its own instantiation (the sorting network
// is generated at compile time), so this file is compiled kLaneParts times, -DNFM_MED_LANE_PART=p holding
// the lengths with red % 8 == p.
#include <utility>
#include "nfm_reduce_median.hpp"

#ifndef NFM_MED_LANE_PART
#error "compile with -DNFM_MED_LANE_PART=0..7"
#endif

namespace nfm {
namespace med {

// ---------------------------------------------------------------------------------------------
// rows of 2..128 elements (float64: 2..64), ONE ROW PER LANE (longer ones, up to twice that: below): the rows of a contiguous (rows, RED) array are
// records like the small matrices of the other kernels -- the workgroup streams its TILE * RED elements with
// 16-byte loads through the LDS transpose (TileIO), every lane picks up its row, sorts the RED keys
// in registers with Batcher's odd-even merge network (compile-time indices: v_min_u32 / v_max_u32 per
// comparator, no cross-lane traffic, ~160 comparators for 27 keys) and reads the key of rank k off
// the sorted array.  64 rows per wavefront instead of 2-8: the kernel becomes a stream over the data.
template <typename U>
__device__ __forceinline__ void cmpxchg(U &a, U &b)
{
    const U lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

// The RED keys of a lane, held as two register arrays (the promotion of ONE large private array to VGPRs is
// subject to a size limit of the compiler; every index used below is a literal, so the split is free).
template <typename U, int RED>
struct Keys {
    static constexpr int H = RED > 64 ? (RED + 1) / 2 : RED;
    U a[H];
    U b[RED - H > 0 ? RED - H : 1];
    __device__ __forceinline__ U &at(int i) { return i < H ? a[i] : b[i - H]; }
};

// Batcher's merge exchange (Knuth 5.2.2, Algorithm M) for any n.  The comparator list is built by a
// constexpr function and applied through a fold over an index sequence, so every register index is a
// literal whatever the unroller's thresholds are (left to `#pragma unroll`, the 6000-iteration loop
// nest of n = 125 was only partly unrolled: dynamic indices, the keys in scratch memory, 50x slower).
template <int RED>
struct MergeExchange {
    static constexpr int t = RED <= 1 ? 0 : (32 - __builtin_clz((unsigned)(RED - 1))); // ceil(log2(RED))
    template <class F>
    static constexpr void each(F &&f) // f(i, j) for every comparator, in order
    {
        for (int pi = t - 1; pi >= 0; --pi) {
            const int p = 1 << pi;
            int q = 1 << (t - 1), r = 0, d = p;
            for (;;) {
                for (int i = 0; i + d < RED; ++i)
                    if ((i & p) == r) f(i, i + d);
                if (q == p) break;
                d = q - p;
                q >>= 1;
                r = p;
            }
        }
    }
    static constexpr int count()
    {
        int n = 0;
        each([&](int, int) { ++n; });
        return n;
    }
    struct List {
        unsigned char a[count() > 0 ? count() : 1], b[count() > 0 ? count() : 1];
    };
    static constexpr List list()
    {
        List l{};
        int n = 0;
        each([&](int i, int j) {
            l.a[n] = (unsigned char)i;
            l.b[n] = (unsigned char)j;
            ++n;
        });
        return l;
    }
};

// (applied in chunks of 1024 comparators: a fold expression nests once per operand, and the front end stops at 2048)
template <typename U, int RED, size_t OFF, size_t... I>
__device__ __forceinline__ void apply_network(Keys<U, RED> &s, std::index_sequence<I...>)
{
    constexpr auto net = MergeExchange<RED>::list();
    (cmpxchg(s.at(net.a[OFF + I]), s.at(net.b[OFF + I])), ...);
}

template <typename U, int RED, size_t OFF = 0>
__device__ __forceinline__ void sort_network(Keys<U, RED> &s)
{
    constexpr size_t total = (size_t)MergeExchange<RED>::count();
    if constexpr (OFF < total) {
        constexpr size_t n = total - OFF < 1024 ? total - OFF : 1024;
        apply_network<U, RED, OFF>(s, std::make_index_sequence<n>{});
        sort_network<U, RED, OFF + n>(s);
    }
}

// The conversion of the hot path is Key::raw (three instructions, no NaN rule); whether a NaN went by is collected
// two elements per compare (v_cmp_u of a PAIR) into a wavefront mask.  Only a wavefront that saw one runs `fix_nans`:
// NaN keys -> the largest key, counted.  (Key::of per element -- compare, select, count -- was 6 instructions a key.)
template <typename T>
__device__ __forceinline__ bool either_nan(T a, T b)
{
    return __builtin_isunordered(a, b);
}
template <typename T, typename U, int RED>
__device__ __forceinline__ unsigned fix_nans(Keys<U, RED> &s, int real)
{
    unsigned nan = 0;
#pragma unroll
    for (int i = 0; i < RED; ++i) {
        const bool n = i < real && Key<T>::raw_is_nan(s.at(i));
        s.at(i) = n ? ~U(0) : s.at(i);
        nan += n ? 1u : 0u;
    }
    return nan;
}

constexpr int64_t kMidGrid2D = 2048; // middle-dim layout: planes at least this wide take the 2-D grid

template <int RED, typename T>
struct LaneTile {
    static constexpr int value = RED * (int)sizeof(T) * 256 <= 36 * 1024 ? 256 : (RED * (int)sizeof(T) * 128 <= 36 * 1024 ? 128 : 64);
};

// `inner` = 1: the rows are contiguous, (rows, RED).  `inner` > 1: the array is (outer, RED, inner) and the
// MIDDLE dim is reduced (a channel dim of a channel-first field): row (o, i) has its elements `inner` apart,
// consecutive lanes hold consecutive i -- every load is already coalesced, no LDS transpose, and the facade
// does not have to move the reduced dim last (a transposing copy of the whole tensor) first.
template <typename T, int RED>
__global__ __launch_bounds__((LaneTile<RED, T>::value)) void median_lane_kernel(const T *__restrict__ x, int64_t rows,
                                                                                int64_t inner, int omitnan,
                                                                                T *__restrict__ val,
                                                                                int64_t *__restrict__ idx)
{
    using K = Key<T>;
    using U = typename K::U;
    constexpr int TILE = LaneTile<RED, T>::value;
    using IO = TileIO<T, RED, TILE>;
    __shared__ __align__(16) unsigned char smem[IO::kLdsBytes];
    const bool mid = inner != 1; // uniform
    // row of this lane, and for the middle-dim layout its (o, i).  No 64-bit division (a 64-iteration
    // software loop per lane: it cost 3x the rest of the kernel): wide planes take a 2-D grid, blockIdx.y
    // + 65535 blockIdx.z = o; narrow ones (inner < kMidGrid2D) a 32-bit division (rows < 2^31 there).
    const bool grid2d = mid && inner >= kMidGrid2D; // the launcher's rule (a single outer index is a 2-D grid too:
                                                    // the 32-bit division below is for rows < 2^31 only)
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    int64_t row = tile0 + threadIdx.x, mo = 0, mi = 0;
    bool live = row < rows;
    if (mid) {
        if (grid2d) {
            mo = (int64_t)blockIdx.y + 65535LL * blockIdx.z;
            mi = tile0 + threadIdx.x;
            live = mi < inner && mo * inner + mi < rows;
            row = mo * inner + (mi < inner ? mi : inner - 1);
        } else {
            const unsigned rr = (unsigned)(live ? row : rows - 1);
            const unsigned q = rr / (unsigned)inner;
            mo = q;
            mi = rr - q * (unsigned)inner;
        }
    }
    Keys<U, RED> s;
    bool seen = false; // a NaN in this lane's row
    // this lane's row: in the LDS image, or in global memory with its elements `inner` apart (two
    // pointers, so that neither becomes a flat pointer)
    const T *lown = reinterpret_cast<const T *>(smem + threadIdx.x * IO::kRowStride);
    const T *gown = x;
    if (!mid) {
        typename IO::Stage st;
        IO::issue(x + tile0 * RED, (rows - tile0) * RED, st);
        IO::commit(smem, st);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RED; i += 2) {
            const T v0 = lown[i], v1 = lown[i + 1 < RED ? i + 1 : i];
            s.at(i) = K::raw(v0);
            if (i + 1 < RED) s.at(i + 1) = K::raw(v1);
            seen |= either_nan(v0, v1);
            // keep the scheduler from hoisting all RED reads above the conversions (2 x RED live registers)
            if (i % 32 == 30) __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // (lanes past the end redo a valid row and store nothing)
        if (grid2d && mo * inner >= rows) mo = 0;
        gown = x + (mo * RED) * inner + (mi < inner ? mi : inner - 1);
#pragma unroll
        for (int i = 0; i < RED; i += 2) {
            const T v0 = NFM_LDG(gown + i * inner), v1 = NFM_LDG(gown + (i + 1 < RED ? i + 1 : i) * inner);
            s.at(i) = K::raw(v0);
            if (i + 1 < RED) s.at(i + 1) = K::raw(v1);
            seen |= either_nan(v0, v1);
            if (i % 32 == 30) __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned nan = 0;
    if (__builtin_expect(__any(seen), 0)) nan = fix_nans<T>(s, RED);
    sort_network<U, RED>(s);
    const unsigned count = omitnan ? (unsigned)RED - nan : (unsigned)RED;
    const bool want_nan = (!omitnan && nan > 0) || count == 0;
    const unsigned k = count ? (count - 1) / 2 : 0;
    U chosen = s.at((RED - 1) / 2); // no NaN, or NaNs kept: the middle of the row
    if (omitnan) {               // NaN keys sort last: rank k of the others
#pragma unroll
        for (int i = 0; i < RED; ++i) chosen = (k == (unsigned)i) ? s.at(i) : chosen;
    }
    if (want_nan) chosen = ~U(0);
    if (live) {
        val[row] = want_nan ? (T)__builtin_nanf("") : K::back(chosen);
        if (idx != nullptr) { // uniform
            // first position holding the chosen key (a NaN result: the first NaN): the row is read again
            // (LDS image / global memory), so the unsorted keys need not stay in registers during the sort
            // A value that is not a NaN is looked for by its bit pattern (compare + select per element); a NaN
            // result by its key, any payload (a wavefront with such a row: rare, a loop that is not unrolled).
            int first = 0;
            if (__builtin_expect(__any(want_nan), 0)) {
#pragma unroll 1
                for (int i = RED - 1; i >= 0; --i) first = (K::of(mid ? gown[i * inner] : lown[i]) == chosen) ? i : first;
            } else if (!mid) {
                const U target = K::bits(K::back(chosen));
#pragma unroll
                for (int i = RED - 1; i >= 0; --i) {
                    first = (K::bits(lown[i]) == target) ? i : first;
                    if (i % 32 == 0) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const U target = K::bits(K::back(chosen));
#pragma unroll
                for (int i = RED - 1; i >= 0; --i) {
                    first = (K::bits(gown[i * inner]) == target) ? i : first;
                    if (i % 32 == 0) __builtin_amdgcn_sched_barrier(0);
                }
            }
            idx[row] = first;
        }
    }
}

template <typename T, int RED>
static int run_lane(int omitnan, int64_t rows, int64_t inner, const void *x, void *val, void *idx, hipStream_t s)
{
    constexpr int TILE = LaneTile<RED, T>::value;
    dim3 grid;
    if (inner >= kMidGrid2D) { // one grid row per outer index: no division in the kernel
        const int64_t outer = rows / inner, nbx = (inner + TILE - 1) / TILE;
        const int64_t gy = outer < 65535 ? outer : 65535, gz = (outer + 65534) / 65535;
        if (nbx > 0x7fffffffLL || gz > 65535) return NFM_ESIZE;
        grid = dim3((unsigned)nbx, (unsigned)gy, (unsigned)gz);
    } else {
        const int64_t nblk = (rows + TILE - 1) / TILE;
        if (nblk > 0x7fffffffLL || (inner != 1 && rows > 0x7fffffffLL)) return NFM_ESIZE;
        grid = dim3((unsigned)nblk, 1, 1);
    }
    hipLaunchKernelGGL((median_lane_kernel<T, RED>), grid, dim3(TILE), 0, s, static_cast<const T *>(x), rows, inner,
                       omitnan, static_cast<T *>(val), static_cast<int64_t *>(idx));
    return launch_status();
}

// ---------------------------------------------------------------------------------------------
// rows of LaneMax+1 .. LanePadMax elements (float32: 129..192, float64: 65..96), still ONE ROW PER LANE: the
// row is padded to the next BUCKET length (a multiple of 16 / 8 elements: eight buckets per dtype instead of
// 128 / 64 more instantiations) with keys that sort after everything, the rank is taken among the `red` real
// ones.  A 2 Ki-element LDS histogram kernel served these lengths before, at 0.5-1.4 TB/s (LDS atomics); the
// network costs ~34 v_min/v_max per key and no cross-lane traffic.  The lane fetches its row itself with
// element-aligned 16-byte loads (no LDS image: 64 rows of up to 1 KiB would be the whole LDS of a
// workgroup; consecutive loads of a lane walk the same cache lines), the index search reads it again.
template <typename T>
struct PadStep {
    static constexpr int value = sizeof(T) == 4 ? 16 : 8;
};
template <typename T, int RED> // RED: bucket length; RED - PadStep < red <= RED
__global__ __launch_bounds__(64) void median_lane_pad_kernel(const T *__restrict__ x, int64_t rows, int red, int omitnan,
                                                             T *__restrict__ val, int64_t *__restrict__ idx)
{
    using K = Key<T>;
    using U = typename K::U;
    using VG = typename VecOf<T>::gtype;
    constexpr int V = VecOf<T>::N;
    constexpr int SURE = RED - PadStep<T>::value; // elements every row of this bucket has
    static_assert(SURE % V == 0 && SURE > 0, "buckets are multiples of the pad step");
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = row < rows;
    const T *own = x + (live ? row : rows - 1) * (int64_t)red;
    Keys<U, RED> s;
    bool seen = false; // a NaN in this lane's row
    // every load of the row is issued before the first conversion (one round trip to memory, not one per group: a
    // kernel with one or two wavefronts per SIMD has nothing else to hide them behind); the conversions then reuse
    // the registers of the loaded words.  (Plain loads, not the nontemporal ones of the streaming kernels: the
    // eight 16-byte loads a lane makes to one 128-byte line come from eight instructions, and the line has to stay
    // in the cache between them.)
    VG words[SURE / V];
    T tail[RED - SURE];
#pragma unroll
    for (int i = 0; i < SURE; i += V) words[i / V] = *reinterpret_cast<const VG *>(own + i);
#pragma unroll
    for (int i = SURE; i < RED; ++i) tail[i - SURE] = own[i < red ? i : red - 1]; // no branch: a slot past the row re-reads the last element
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < SURE; i += V) {
#pragma unroll
        for (int q = 0; q < V; q += 2) {
            s.at(i + q) = K::raw(words[i / V][q]);
            s.at(i + q + 1) = K::raw(words[i / V][q + 1]);
            seen |= either_nan(words[i / V][q], words[i / V][q + 1]);
        }
    }
#pragma unroll
    for (int i = SURE; i < RED; ++i) {
        const T v = tail[i - SURE];
        const U kv = K::raw(v);
        s.at(i) = i < red ? kv : ~U(0); // padding: the NaN key, sorts last (not counted in `nan`)
        seen |= v != v;
    }
    unsigned nan = 0;
    if (__builtin_expect(__any(seen), 0)) nan = fix_nans<T>(s, red);
    sort_network<U, RED>(s);
    const unsigned count = omitnan ? (unsigned)red - nan : (unsigned)red;
    const bool want_nan = (!omitnan && nan > 0) || count == 0;
    const unsigned k = count ? (count - 1) / 2 : 0;
    // rank k of the sorted keys; k < red <= RED and k >= (RED - PadStep - 1) / 2 - ... : only ranks a row of this
    // bucket can ask for are looked at (omitnan lowers the rank by up to half the NaN count)
    U chosen = s.at(0);
#pragma unroll
    for (int i = 1; i <= (RED - 1) / 2; ++i) chosen = (k == (unsigned)i) ? s.at(i) : chosen;
    if (want_nan) chosen = ~U(0);
    if (live) {
        val[row] = want_nan ? (T)__builtin_nanf("") : K::back(chosen);
        if (idx != nullptr) { // uniform: first position holding the chosen value, by its bit pattern (a NaN result:
                              // the first NaN of any payload, by its key); the row is read again
            int first = 0;
            if (__builtin_expect(__any(want_nan), 0)) {
#pragma unroll 1
                for (int i = red - 1; i >= 0; --i) first = (K::of(own[i]) == chosen) ? i : first;
            } else {
                const U target = K::bits(K::back(chosen));
#pragma unroll
                for (int i = RED - 1; i >= SURE; --i) {
                    const U b = K::bits(own[i < red ? i : red - 1]);
                    first = (i < red && b == target) ? i : first;
                }
#pragma unroll
                for (int i = SURE - V; i >= 0; i -= V) {
                    const VG v = *reinterpret_cast<const VG *>(own + i);
#pragma unroll
                    for (int q = V - 1; q >= 0; --q) first = (K::bits(v[q]) == target) ? i + q : first;
                    if (i % 32 == 0) __builtin_amdgcn_sched_barrier(0);
                }
            }
            idx[row] = first;
        }
    }
}

template <typename T, int RED>
static int run_lane_pad(int red, int omitnan, int64_t rows, const void *x, void *val, void *idx, hipStream_t s)
{
    const int64_t nblk = (rows + 63) / 64;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    hipLaunchKernelGGL((median_lane_pad_kernel<T, RED>), dim3((unsigned)nblk), dim3(64), 0, s, static_cast<const T *>(x),
                       rows, red, omitnan, static_cast<T *>(val), static_cast<int64_t *>(idx));
    return launch_status();
}

// bucket b = 0..7 of dtype T: lengths LaneMax + b * step + 1 .. LaneMax + (b + 1) * step; part p holds bucket p
template <typename T>
static int lane_pad_bucket(int red, int omitnan, int64_t rows, const void *x, void *val, void *idx, hipStream_t s)
{
    constexpr int b = NFM_MED_LANE_PART - LanePadBuckets<T>::first_part; // this part's bucket of dtype T, if any
    if constexpr (b < 0 || b >= LanePadBuckets<T>::value) {
        return NFM_EINVAL;
    } else {
        constexpr int RED = LaneMax<T>::value + (b + 1) * PadStep<T>::value;
        if (red <= RED - PadStep<T>::value || red > RED) return NFM_EINVAL;
        return run_lane_pad<T, RED>(red, omitnan, rows, x, val, idx, s);
    }
}

// ---------------------------------------------------------------------------------------------
// rows of LanePadMax+1 .. 2 LaneMax elements (float32: 193..256, float64: 97..128): FOUR LANES PER ROW (a quad).
// Each lane sorts a quarter of the row (H = LaneMax / 2 keys) with the merge-exchange network; the quarters then
// meet in two bitonic steps, partner registers coming through DPP:
//   level 1, lanes (0,1) and (2,3): for two sorted runs A, B the H smallest keys of their union are
//     { min(A[i], B[H-1-i]) } and the H largest { max(A[i], B[H-1-i]) } -- the even lane keeps the first set, the
//     odd lane the second; each set is a bitonic sequence, which log2(H) half-cleaner stages sort inside the lane;
//   level 2, lanes (0,3) and (1,2): with AB and CD now sorted across two lanes each, the key of rank 2H-1 of the
//     4 H slots is max_i min(AB[i], CD[2H-1-i]): H v_min per lane against the mirrored partner, a running v_max,
//     and a maximum over the quad.
// The rank wanted, k = (count-1)/2 of `count` real keys, is MADE to be 2H-1: the 4 H - count slots that hold no real
// key (padding, omitted NaNs) are filled with 2H-1-k smallest keys (0: never a real key) and largest keys (~0) for
// the rest.  With no omitted NaN in the wavefront the split is the same for every row and decided per slot index;
// rows with omitted NaNs turn that many more of their NaN keys into smallest keys (a vote, a sequential pass).
// ~34 instructions per key on ~100 registers: several wavefronts per SIMD hide the row fetch, which a lane does
// itself with element-aligned 16-byte loads.  (Two lanes per row, 128 keys each, was built first: 29 instructions
// per key but 260 registers, one wavefront per SIMD, 39 % of its time waiting for its loads: 1.9-2.05 TB/s.)
template <int CTRL>
__device__ __forceinline__ unsigned quad_dpp32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL, typename U>
__device__ __forceinline__ U quad_dpp(U v)
{
    if constexpr (sizeof(U) == 4) {
        return (U)quad_dpp32<CTRL>((unsigned)v);
    } else {
        const unsigned lo = quad_dpp32<CTRL>((unsigned)v), hi = quad_dpp32<CTRL>((unsigned)((unsigned long long)v >> 32));
        return (U)(((unsigned long long)hi << 32) | lo);
    }
}
constexpr int kQuadSwap1 = 0xB1;   // quad_perm [1,0,3,2]: lane ^ 1
constexpr int kQuadSwap2 = 0x4E;   // quad_perm [2,3,0,1]: lane ^ 2
constexpr int kQuadMirror = 0x1B;  // quad_perm [3,2,1,0]: lane ^ 3

template <typename T>
__global__ __launch_bounds__(64) void median_lane_quad_kernel(const T *__restrict__ x, int64_t rows, int red, int omitnan,
                                                              T *__restrict__ val, int64_t *__restrict__ idx)
{
    using K = Key<T>;
    using U = typename K::U;
    using VG = typename VecOf<T>::gtype;
    constexpr int V = VecOf<T>::N;
    constexpr int H = LaneMax<T>::value / 2; // key slots per lane
    constexpr int SURE = (3 * H) / 4;        // real elements every lane of this kernel has (red > 3 H)
    static_assert(SURE % V == 0 && (H & (H - 1)) == 0, "vector loads cover the sure part; H is a power of two");
    const int q = threadIdx.x & 3;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 2);
    const bool live = row < rows;
    // the row is cut in four: lane q takes mine(q) = (red + 3 - q) / 4 elements from off(q)
    const int m0 = (red + 3) / 4, m1 = (red + 2) / 4, m2 = (red + 1) / 4, m3 = red / 4;
    const int mine = q == 0 ? m0 : q == 1 ? m1 : q == 2 ? m2 : m3;
    const int off = q == 0 ? 0 : q == 1 ? m0 : q == 2 ? m0 + m1 : m0 + m1 + m2;
    const T *own = x + (live ? row : rows - 1) * (int64_t)red + off;
    // padding: 4 H - red slots, H - mine(q) in lane q; the first lo0 = 2H-1 - (red-1)/2 of them (lane 0's first, then
    // lane 1's, ...) hold the smallest key, the others the largest: the split of a row without omitted NaNs
    const int lo0 = (2 * H - 1) - (red - 1) / 2;
    const int before = q == 0 ? 0 : q == 1 ? H - m0 : q == 2 ? 2 * H - m0 - m1 : 3 * H - m0 - m1 - m2; // padding slots of the lanes before mine
    const int zmine = lo0 - before; // smallest-key slots among this lane's padding (<= 0: none; >= its count: all)
    Keys<U, H> s;
    bool seen = false; // a NaN in this lane's part of the row
#pragma unroll
    for (int i = 0; i < SURE; i += V) {
        const VG v = *reinterpret_cast<const VG *>(own + i); // plain loads: a lane walks its cache lines with consecutive loads
#pragma unroll
        for (int c = 0; c < V; c += 2) {
            s.at(i + c) = K::raw(v[c]);
            s.at(i + c + 1) = K::raw(v[c + 1]);
            seen |= either_nan(v[c], v[c + 1]);
        }
    }
#pragma unroll
    for (int e = SURE; e < H; ++e) { // single, address-clamped loads and a select: no branch, all in flight together
        const bool real = e < mine;
        const T v = own[real ? e : mine - 1]; // (a padding slot re-reads the last element: its NaN is seen anyway)
        const U pad = (e - mine < zmine) ? U(0) : ~U(0); // padding slot number e - mine of this lane
        const U kv = K::raw(v); // (both arms evaluated ahead of the select: a call in an arm becomes a branch, and a
                                // branch per slot is a load, a wait, a load, a wait ...)
        s.at(e) = real ? kv : pad;
        seen |= v != v;
    }
    unsigned nan = 0;
    if (__builtin_expect(__any(seen), 0)) nan = fix_nans<T>(s, mine);
    const unsigned n1 = nan + quad_dpp32<kQuadSwap1>(nan);
    const unsigned nan_row = n1 + quad_dpp32<kQuadSwap2>(n1);
    const unsigned count = omitnan ? (unsigned)red - nan_row : (unsigned)red;
    const bool want_nan = (!omitnan && nan_row > 0) || count == 0;
    if (__builtin_expect(__any(omitnan && nan_row > 0 && count > 0), 0)) {
        // omitted NaNs lower the rank: k = (count-1)/2, so (red-1)/2 - k more smallest keys are needed; they are
        // made out of this row's NaN keys, lane 0 of the quad first
        const int more = (omitnan && count > 0) ? (int)((unsigned)(red - 1) / 2 - (count - 1) / 2) : 0;
        const int n_0 = (int)quad_dpp32<0x00>(nan), n_1 = (int)quad_dpp32<0x55>(nan), n_2 = (int)quad_dpp32<0xAA>(nan); // lanes 0, 1, 2 of the quad
        const int prior = q == 0 ? 0 : q == 1 ? n_0 : q == 2 ? n_0 + n_1 : n_0 + n_1 + n_2;
        int budget = more - prior;
        budget = budget < 0 ? 0 : budget;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const bool real_nan = i < mine && s.at(i) == ~U(0);
            const bool turn = real_nan && budget > 0;
            s.at(i) = turn ? U(0) : s.at(i);
            budget -= turn ? 1 : 0;
        }
    }
    sort_network<U, H>(s);
    // level 1: the even lane of a pair keeps the H smallest of the pair's keys, the odd lane the H largest
    {
        const bool odd = (q & 1) != 0;
        U t[H];
#pragma unroll
        for (int i = 0; i < H; ++i) t[i] = quad_dpp<kQuadSwap1>(s.at(H - 1 - i));
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const U a = s.at(i), lo = a < t[i] ? a : t[i], hi = a < t[i] ? t[i] : a;
            s.at(i) = odd ? hi : lo;
        }
        // a bitonic sequence: log2(H) half-cleaner stages sort it
#pragma unroll
        for (int d = H / 2; d >= 1; d /= 2)
#pragma unroll
            for (int i = 0; i < H; ++i)
                if ((i & d) == 0) cmpxchg(s.at(i), s.at(i + d));
    }
    // level 2: rank 2H-1 of the quad's 4 H slots
    U chosen = U(0);
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const U other = quad_dpp<kQuadMirror>(s.at(H - 1 - i));
        const U m = s.at(i) < other ? s.at(i) : other;
        chosen = m > chosen ? m : chosen;
    }
    {
        const U o1 = quad_dpp<kQuadSwap1>(chosen);
        chosen = o1 > chosen ? o1 : chosen;
        const U o2 = quad_dpp<kQuadSwap2>(chosen);
        chosen = o2 > chosen ? o2 : chosen;
    }
    if (want_nan) chosen = ~U(0);
    if (live) {
        int first = 0x7fffffff;
        if (idx != nullptr) { // uniform: first position holding the chosen value, by its bit pattern (a NaN result:
                              // the first NaN of any payload, by its key); the row is read again
            if (__builtin_expect(__any(want_nan), 0)) {
#pragma unroll 1
                for (int i = mine - 1; i >= 0; --i) first = (K::of(own[i]) == chosen) ? i : first;
            } else {
                const U target = K::bits(K::back(chosen));
#pragma unroll
                for (int e = H - 1; e >= SURE; --e) {
                    const bool real = e < mine;
                    const U b = K::bits(own[real ? e : mine - 1]);
                    first = (real && b == target) ? e : first;
                }
#pragma unroll
                for (int i = SURE - V; i >= 0; i -= V) {
                    const VG v = *reinterpret_cast<const VG *>(own + i);
#pragma unroll
                    for (int c = V - 1; c >= 0; --c) first = (K::bits(v[c]) == target) ? i + c : first;
                }
            }
            first = first == 0x7fffffff ? first : first + off;
            const int f1 = (int)quad_dpp32<kQuadSwap1>((unsigned)first);
            first = f1 < first ? f1 : first;
            const int f2 = (int)quad_dpp32<kQuadSwap2>((unsigned)first);
            first = f2 < first ? f2 : first;
        }
        if (q == 0) {
            val[row] = want_nan ? (T)__builtin_nanf("") : K::back(chosen);
            if (idx != nullptr) idx[row] = first;
        }
    }
}

template <typename T>
static int run_lane_quad(int red, int omitnan, int64_t rows, const void *x, void *val, void *idx, hipStream_t s)
{
    if (red <= LanePadMax<T>::value || red > 2 * LaneMax<T>::value) return NFM_EINVAL;
    const int64_t nblk = (rows + 15) / 16;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    hipLaunchKernelGGL((median_lane_quad_kernel<T>), dim3((unsigned)nblk), dim3(64), 0, s, static_cast<const T *>(x), rows,
                       red, omitnan, static_cast<T *>(val), static_cast<int64_t *>(idx));
    return launch_status();
}

// the lengths of this part, RED = first, first + 8, ... <= LaneMax
template <typename T, int RED>
static int lane_chain(int red, int omitnan, int64_t rows, int64_t inner, const void *x, void *val, void *idx,
                      hipStream_t s)
{
    if constexpr (RED > LaneMax<T>::value) {
        return NFM_EINVAL;
    } else {
        if (red == RED) return run_lane<T, RED>(omitnan, rows, inner, x, val, idx, s);
        return lane_chain<T, RED + kLaneParts>(red, omitnan, rows, inner, x, val, idx, s);
    }
}

#define NFM_MED_CAT2(a, b) a##b
#define NFM_MED_CAT(a, b) NFM_MED_CAT2(a, b)
int NFM_MED_CAT(lane_part, NFM_MED_LANE_PART)(int dtype, int red, int omitnan, int64_t rows, int64_t inner,
                                              const void *x, void *val, void *idx, void *stream)
{
    constexpr int first = NFM_MED_LANE_PART >= 2 ? NFM_MED_LANE_PART : NFM_MED_LANE_PART + kLaneParts; // lengths start at 2
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (red < 0) { // padded rows (contiguous only): -red in this part's bucket (lane_pad_any picks the part)
#if NFM_MED_LANE_PART == 5 || NFM_MED_LANE_PART == 1 // the four-lanes-per-row kernels live in parts 5 (float32) and 1 (float64)
        if (-red > (dtype == NFM_F32 ? LanePadMax<float>::value : LanePadMax<double>::value)) {
            if (dtype == (NFM_MED_LANE_PART == 5 ? NFM_F32 : NFM_F64)) {
#if NFM_MED_LANE_PART == 5
                return run_lane_quad<float>(-red, omitnan, rows, x, val, idx, s);
#else
                return run_lane_quad<double>(-red, omitnan, rows, x, val, idx, s);
#endif
            }
            return NFM_EINVAL;
        }
#endif
        return dtype == NFM_F32 ? lane_pad_bucket<float>(-red, omitnan, rows, x, val, idx, s)
                                : lane_pad_bucket<double>(-red, omitnan, rows, x, val, idx, s);
    }
    return dtype == NFM_F32 ? lane_chain<float, first>(red, omitnan, rows, inner, x, val, idx, s)
                            : lane_chain<double, first>(red, omitnan, rows, inner, x, val, idx, s);
}

} // namespace med
} // namespace nfm

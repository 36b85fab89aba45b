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
    unsigned nan = 0;
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
        for (int i = 0; i < RED; ++i) {
            const T v = lown[i];
            s.at(i) = K::of(v);
            nan += (v != v) ? 1u : 0u;
            // keep the scheduler from hoisting all RED reads above the conversions (2 x RED live registers)
            if (i % 32 == 31) __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // (lanes past the end redo a valid row and store nothing)
        if (grid2d && mo * inner >= rows) mo = 0;
        gown = x + (mo * RED) * inner + (mi < inner ? mi : inner - 1);
#pragma unroll
        for (int i = 0; i < RED; ++i) {
            const T v = NFM_LDG(gown + i * inner);
            s.at(i) = K::of(v);
            nan += (v != v) ? 1u : 0u;
            if (i % 32 == 31) __builtin_amdgcn_sched_barrier(0);
        }
    }
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
            int first = 0;
            if (!mid) {
#pragma unroll
                for (int i = RED - 1; i >= 0; --i) {
                    first = (K::of(lown[i]) == chosen) ? i : first;
                    if (i % 32 == 0) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = RED - 1; i >= 0; --i) {
                    first = (K::of(gown[i * inner]) == chosen) ? i : first;
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
    unsigned nan = 0;
#pragma unroll
    for (int i = 0; i < SURE; i += V) {
        // (plain loads, not the nontemporal ones of the streaming kernels: the eight 16-byte loads a lane makes
        // to one 128-byte line come from eight instructions, and the line has to stay in the cache between them)
        const VG v = *reinterpret_cast<const VG *>(own + i);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            s.at(i + q) = K::of(v[q]);
            nan += (v[q] != v[q]) ? 1u : 0u;
        }
        if (i % 32 == 32 - V) __builtin_amdgcn_sched_barrier(0); // keep the loads from piling up ahead of the conversions
    }
#pragma unroll
    for (int i = SURE; i < RED; ++i) {
        if (i < red) { // uniform
            const T v = own[i];
            s.at(i) = K::of(v);
            nan += (v != v) ? 1u : 0u;
        } else {
            s.at(i) = ~U(0); // padding: the NaN key, sorts last (not counted in `nan`)
        }
    }
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
        if (idx != nullptr) { // uniform: first position holding the chosen key (a NaN result: the first NaN)
            int first = 0;
            for (int i = red - 1; i >= 0; --i) first = (K::of(own[i]) == chosen) ? i : first;
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
// rows of LanePadMax+1 .. 2 LaneMax elements (float32: 193..256, float64: 97..128): TWO LANES PER ROW.  Each lane
// of a pair sorts one half of the row (H = LaneMax keys, the network the exact-length kernels use), then the two
// sorted halves A, B are combined without a merge: the H smallest keys of their union are
// { min(A[i], B[H-1-i]) : i < H } (the first half of a bitonic split), so the key of rank H-1 of the 2 H slots is
// max_i min(A[i], B[H-1-i]) -- H v_min with the partner's register through DPP (quad_perm [1,0,3,2]) and H-1 v_max.
// The rank wanted, k = (count-1)/2 of `count` real keys, is MADE to be H-1: the 2 H - count slots that hold no real
// key (padding, omitted NaNs) are filled with H-1-k smallest keys (0: never a real key) and largest keys (~0) for
// the rest; k + (H-1-k) = H-1.  With no NaN in the wavefront the split is the same for every row and is decided
// per slot index; rows with omitted NaNs turn that many more of their NaN keys into smallest keys (a vote, a
// sequential pass).  ~27 compare-exchange instructions per key against ~30 for one lane sorting 2 H keys, on half
// the registers (three wavefronts per SIMD instead of one) and at the compile time of the H-key network.
__device__ __forceinline__ unsigned pair_swap32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
}
template <typename U>
__device__ __forceinline__ U pair_swap(U v)
{
    if constexpr (sizeof(U) == 4) {
        return (U)pair_swap32((unsigned)v);
    } else {
        const unsigned lo = pair_swap32((unsigned)v), hi = pair_swap32((unsigned)((unsigned long long)v >> 32));
        return (U)(((unsigned long long)hi << 32) | lo);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void median_lane_pair_kernel(const T *__restrict__ x, int64_t rows, int red, int omitnan,
                                                              T *__restrict__ val, int64_t *__restrict__ idx)
{
    using K = Key<T>;
    using U = typename K::U;
    using VG = typename VecOf<T>::gtype;
    constexpr int V = VecOf<T>::N;
    constexpr int H = LaneMax<T>::value;  // key slots per lane
    const int h = threadIdx.x & 1;
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 1);
    const bool live = row < rows;
    // the row is split in the middle: lane 0 of the pair takes the first mine0 = ceil(red / 2) elements, lane 1 the
    // other floor(red / 2) (3 H / 4 < mine <= H each), so both fetch with the same 16-byte loads up to `vend` and
    // at most V + 1 single elements after it; their remaining slots are padding
    const int mine0 = (red + 1) / 2, mine1 = red / 2;
    const int mine = h ? mine1 : mine0;
    const int vend = (mine1 / V) * V;  // uniform
    const T *own = x + (live ? row : rows - 1) * (int64_t)red + h * mine0;
    // padding: 2 H - red slots, pads0 of them in lane 0; the first lo0 = H-1 - (red-1)/2 of them (lane 0's first) hold
    // the smallest key, the others the largest: the split of a row without omitted NaNs (the same for every row)
    const int lo0 = (H - 1) - (red - 1) / 2;
    const int pads0 = H - mine0;
    const int z0 = lo0 < pads0 ? lo0 : pads0;
    const int zmine = h ? lo0 - z0 : z0; // smallest-key slots among this lane's padding (its first zmine padding slots)
    Keys<U, H> s;
    unsigned nan = 0;
#pragma unroll
    for (int i = 0; i < H; i += V) {
        if (i < vend) { // uniform
            const VG v = *reinterpret_cast<const VG *>(own + i); // plain loads: a lane walks its cache lines with consecutive loads
#pragma unroll
            for (int q = 0; q < V; ++q) {
                s.at(i + q) = K::of(v[q]);
                nan += (v[q] != v[q]) ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const int e = i + q;
                U key = (e - mine < zmine) ? U(0) : ~U(0); // padding slot number e - mine of this lane
                if (e < mine0) { // uniform: at most V + 1 elements past vend are real in either lane
                    const bool real = e < mine;
                    const T v = own[real ? e : mine - 1];
                    key = real ? K::of(v) : key;
                    nan += (real && v != v) ? 1u : 0u;
                }
                s.at(e) = key;
            }
        }
        if (i % 32 == 32 - V) __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned nan_row = nan + pair_swap(nan);
    const unsigned count = omitnan ? (unsigned)red - nan_row : (unsigned)red;
    const bool want_nan = (!omitnan && nan_row > 0) || count == 0;
    if (__builtin_expect(__any(omitnan && nan_row > 0 && count > 0), 0)) {
        // omitted NaNs lower the rank: k = (count-1)/2, so (red-1)/2 - k more smallest keys are needed; they are
        // made out of this row's NaN keys, the first lane of the pair first
        const int more = (omitnan && count > 0) ? (int)((unsigned)(red - 1) / 2 - (count - 1) / 2) : 0;
        const int nan0 = (int)(h ? nan_row - nan : nan);
        const int take0 = more < nan0 ? more : nan0; // lane 0's share
        int budget = h ? more - take0 : take0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const bool real_nan = i < mine && s.at(i) == ~U(0);
            const bool turn = real_nan && budget > 0;
            s.at(i) = turn ? U(0) : s.at(i);
            budget -= turn ? 1 : 0;
        }
    }
    sort_network<U, H>(s);
    // rank H-1 of the pair's 2 H slots
    U chosen = U(0);
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const U other = pair_swap(s.at(H - 1 - i));
        const U m = s.at(i) < other ? s.at(i) : other;
        chosen = m > chosen ? m : chosen;
    }
    if (want_nan) chosen = ~U(0);
    if (live) {
        int first = 0x7fffffff;
        if (idx != nullptr) { // uniform: first position holding the chosen key (a NaN result: the first NaN)
            for (int i = mine - 1; i >= 0; --i) first = (K::of(own[i]) == chosen) ? h * mine0 + i : first;
            const int of = (int)pair_swap((unsigned)first);
            first = of < first ? of : first;
        }
        if (h == 0) {
            val[row] = want_nan ? (T)__builtin_nanf("") : K::back(chosen);
            if (idx != nullptr) idx[row] = first;
        }
    }
}

template <typename T>
static int run_lane_pair(int red, int omitnan, int64_t rows, const void *x, void *val, void *idx, hipStream_t s)
{
    if (red <= LanePadMax<T>::value || red > 2 * LaneMax<T>::value) return NFM_EINVAL;
    const int64_t nblk = (rows + 31) / 32;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    hipLaunchKernelGGL((median_lane_pair_kernel<T>), dim3((unsigned)nblk), dim3(64), 0, s, static_cast<const T *>(x), rows,
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
#if NFM_MED_LANE_PART == 5 || NFM_MED_LANE_PART == 1 // the two-lanes-per-row kernels live in parts 5 (float32) and 1 (float64)
        if (-red > (dtype == NFM_F32 ? LanePadMax<float>::value : LanePadMax<double>::value)) {
            if (dtype == (NFM_MED_LANE_PART == 5 ? NFM_F32 : NFM_F64)) {
#if NFM_MED_LANE_PART == 5
                return run_lane_pair<float>(-red, omitnan, rows, x, val, idx, s);
#else
                return run_lane_pair<double>(-red, omitnan, rows, x, val, idx, s);
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

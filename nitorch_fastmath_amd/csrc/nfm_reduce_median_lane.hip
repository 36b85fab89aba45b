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
// rows of 2..128 elements (float64: 2..64), ONE ROW PER LANE: the rows of a contiguous (rows, RED) array are records
// like the small matrices of the other kernels -- the workgroup streams its TILE * RED elements with
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

// The RED keys of a lane, held as two register arrays: the compiler keeps a private array in VGPRs only
// up to a size limit (an array of more than ~100 dwords went to scratch memory, 50x slower), and every
// index below is a compile-time constant after unrolling, so the split costs nothing.
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

template <typename U, int RED, size_t... I>
__device__ __forceinline__ void apply_network(Keys<U, RED> &s, std::index_sequence<I...>)
{
    constexpr auto net = MergeExchange<RED>::list();
    (cmpxchg(s.at(net.a[I]), s.at(net.b[I])), ...);
}

template <typename U, int RED>
__device__ __forceinline__ void sort_network(Keys<U, RED> &s)
{
    apply_network<U, RED>(s, std::make_index_sequence<(size_t)MergeExchange<RED>::count()>{});
}

template <int RED, typename T>
struct LaneTile {
    static constexpr int value = RED * (int)sizeof(T) * 256 <= 36 * 1024 ? 256 : (RED * (int)sizeof(T) * 128 <= 36 * 1024 ? 128 : 64);
};

template <typename T, int RED>
__global__ __launch_bounds__((LaneTile<RED, T>::value)) void median_lane_kernel(const T *__restrict__ x, int64_t rows,
                                                                                int omitnan, T *__restrict__ val,
                                                                                int64_t *__restrict__ idx)
{
    using K = Key<T>;
    using U = typename K::U;
    constexpr int TILE = LaneTile<RED, T>::value;
    using IO = TileIO<T, RED, TILE>;
    __shared__ __align__(16) unsigned char smem[IO::kLdsBytes];
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int64_t row = tile0 + threadIdx.x;
    typename IO::Stage st;
    IO::issue(x + tile0 * RED, (rows - tile0) * RED, st);
    IO::commit(smem, st);
    __syncthreads();
    // this lane's row in the LDS image (read element by element: no second register array)
    const T *own = reinterpret_cast<const T *>(smem + threadIdx.x * IO::kRowStride);
    Keys<U, RED> s;
    unsigned nan = 0;
#pragma unroll
    for (int i = 0; i < RED; ++i) {
        const T v = own[i];
        s.at(i) = K::of(v);
        nan += (v != v) ? 1u : 0u;
        // keep the scheduler from hoisting all RED reads above the conversions (2 x RED live registers)
        if (i % 32 == 31) __builtin_amdgcn_sched_barrier(0);
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
    if (row < rows) {
        val[row] = want_nan ? (T)__builtin_nanf("") : K::back(chosen);
        if (idx != nullptr) { // uniform
            // first position holding the chosen key (a NaN result: the first NaN): the row is still in
            // the LDS image, so the unsorted keys need not stay in registers during the sort
            int first = 0;
#pragma unroll
            for (int i = RED - 1; i >= 0; --i) {
                first = (K::of(own[i]) == chosen) ? i : first;
                if (i % 32 == 0) __builtin_amdgcn_sched_barrier(0);
            }
            idx[row] = first;
        }
    }
}

template <typename T, int RED>
static int run_lane(int omitnan, int64_t rows, const void *x, void *val, void *idx, hipStream_t s)
{
    constexpr int TILE = LaneTile<RED, T>::value;
    const int64_t nblk = (rows + TILE - 1) / TILE;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    hipLaunchKernelGGL((median_lane_kernel<T, RED>), dim3((unsigned)nblk), dim3(TILE), 0, s, static_cast<const T *>(x),
                       rows, omitnan, static_cast<T *>(val), static_cast<int64_t *>(idx));
    return launch_status();
}

// the lengths of this part, RED = first, first + 8, ... <= LaneMax
template <typename T, int RED>
static int lane_chain(int red, int omitnan, int64_t rows, const void *x, void *val, void *idx, hipStream_t s)
{
    if constexpr (RED > LaneMax<T>::value) {
        return NFM_EINVAL;
    } else {
        if (red == RED) return run_lane<T, RED>(omitnan, rows, x, val, idx, s);
        return lane_chain<T, RED + kLaneParts>(red, omitnan, rows, x, val, idx, s);
    }
}

#define NFM_MED_CAT2(a, b) a##b
#define NFM_MED_CAT(a, b) NFM_MED_CAT2(a, b)
int NFM_MED_CAT(lane_part, NFM_MED_LANE_PART)(int dtype, int red, int omitnan, int64_t rows, const void *x, void *val,
                                              void *idx, void *stream)
{
    constexpr int first = NFM_MED_LANE_PART >= 2 ? NFM_MED_LANE_PART : NFM_MED_LANE_PART + kLaneParts; // lengths start at 2
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == NFM_F32 ? lane_chain<float, first>(red, omitnan, rows, x, val, idx, s)
                            : lane_chain<double, first>(red, omitnan, rows, x, val, idx, s);
}

} // namespace med
} // namespace nfm

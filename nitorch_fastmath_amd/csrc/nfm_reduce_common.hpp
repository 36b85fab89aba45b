// nfm_reduce_common.hpp -- pieces shared by the full (nfm_reduce.hip) and the dim-wise
// (nfm_reduce_dim.hip) NaN-aware reductions of the reference's `reduce.py`.
#pragma once
#include "nfm_common.hpp"

namespace nfm {

template <int OP>
struct RedOp {
    static constexpr bool is_sum = OP == NFM_RED_NANSUM || OP == NFM_RED_SUM || OP == NFM_RED_NANCOUNT ||
                                   OP == NFM_RED_NANSUMSQ;
    static constexpr bool is_max = OP == NFM_RED_NANMAX || OP == NFM_RED_MAX;
    __device__ static __forceinline__ double identity()
    {
        return is_sum ? 0.0 : (is_max ? -__builtin_inf() : __builtin_inf());
    }
    // fold one element into an accumulator
    template <typename T>
    __device__ static __forceinline__ void fold(double &acc, T v)
    {
        const double d = (double)v;
        if constexpr (OP == NFM_RED_NANSUM) acc += (v == v) ? d : 0.0;
        else if constexpr (OP == NFM_RED_SUM) acc += d;
        else if constexpr (OP == NFM_RED_NANCOUNT) acc += (v == v) ? 1.0 : 0.0;
        else if constexpr (OP == NFM_RED_NANSUMSQ) acc += (v == v) ? d * d : 0.0;
        else if constexpr (OP == NFM_RED_NANMAX) acc = d > acc ? d : acc;
        else if constexpr (OP == NFM_RED_NANMIN) acc = d < acc ? d : acc;
        else if constexpr (OP == NFM_RED_MAX) acc = (d > acc || d != d) ? d : acc;
        else acc = (d < acc || d != d) ? d : acc;
    }
    // combine two accumulators
    __device__ static __forceinline__ double merge(double a, double b)
    {
        if constexpr (is_sum) return a + b;
        else if constexpr (OP == NFM_RED_NANMAX) return b > a ? b : a;
        else if constexpr (OP == NFM_RED_NANMIN) return b < a ? b : a;
        else if constexpr (OP == NFM_RED_MAX) return (a != a) ? a : ((b > a || b != b) ? b : a);
        else return (a != a) ? a : ((b < a || b != b) ? b : a);
    }
};

template <int OP>
__device__ __forceinline__ double wave_reduce(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = RedOp<OP>::merge(v, __shfl_xor(v, off, kWave));
    return v;
}

// ---- max/min that also track the position of the selected element: the FIRST occurrence
// of the extremum after NaN replacement (nan ops) or the first NaN (propagating ops), which
// is what torch.max/min(dim) return on the reference's path (reduce.py:129-140).
template <int OP>
struct Pick {
    static constexpr bool is_max = RedOp<OP>::is_max;
    static constexpr bool omit = OP == NFM_RED_NANMAX || OP == NFM_RED_NANMIN;
    // value as the reduction sees it
    template <typename U>
    __device__ static __forceinline__ U see(U v)
    {
        return (omit && v != v) ? (U)RedOp<OP>::identity() : v;
    }
    // is candidate w strictly better than the current value?
    template <typename U>
    __device__ static __forceinline__ bool better(U w, U cur)
    {
        return cur == cur && (w != w || (is_max ? w > cur : w < cur));
    }
    template <typename U>
    __device__ static __forceinline__ bool same(U w, U cur)
    {
        return w == cur || (w != w && cur != cur);
    }
};

// ---- one-pass moments: [count, sum(x - K), sum((x - K)^2), K] over the non-NaN elements,
// K = the first finite element (a shift that removes the cancellation of the raw-moment
// variance formula).  Feeds nanmean / nanvar / nanstd (reduce.py:553-763) in ONE pass over
// memory instead of three.
struct Mom {
    double n, s, q;
};
__device__ __forceinline__ Mom mom_merge(Mom a, Mom b) { return {a.n + b.n, a.s + b.s, a.q + b.q}; }
template <typename T>
__device__ __forceinline__ void mom_fold(Mom &m, T v, double shift)
{
    const bool ok = v == v;
    const double d = ok ? (double)v - shift : 0.0;
    m.n += ok ? 1.0 : 0.0;
    m.s += d;
    m.q += d * d;
}
__device__ __forceinline__ Mom mom_wave(Mom m)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m.n += __shfl_xor(m.n, off, kWave);
        m.s += __shfl_xor(m.s, off, kWave);
        m.q += __shfl_xor(m.q, off, kWave);
    }
    return m;
}
template <typename T>
__device__ __forceinline__ double pick_shift(const T *x, int64_t n, int64_t stride)
{
    // first finite value among the first few elements (uniform across the block)
    double k = 0.0;
    for (int64_t j = 0; j < n && j < 8; ++j) {
        const double v = (double)x[j * stride];
        if (v == v && v - v == 0.0) { k = v; break; }
    }
    return k;
}

#define NFM_SWITCH_OP(op, CALL)                         \
    switch (op) {                                       \
    case NFM_RED_NANSUM: { constexpr int OP = NFM_RED_NANSUM; CALL; } break;     \
    case NFM_RED_NANMAX: { constexpr int OP = NFM_RED_NANMAX; CALL; } break;     \
    case NFM_RED_NANMIN: { constexpr int OP = NFM_RED_NANMIN; CALL; } break;     \
    case NFM_RED_SUM: { constexpr int OP = NFM_RED_SUM; CALL; } break;           \
    case NFM_RED_MAX: { constexpr int OP = NFM_RED_MAX; CALL; } break;           \
    case NFM_RED_MIN: { constexpr int OP = NFM_RED_MIN; CALL; } break;           \
    case NFM_RED_NANCOUNT: { constexpr int OP = NFM_RED_NANCOUNT; CALL; } break; \
    case NFM_RED_NANSUMSQ: { constexpr int OP = NFM_RED_NANSUMSQ; CALL; } break; \
    default: return NFM_EINVAL;                         \
    }


constexpr int kRedBlocks = 2048; // full reductions: 8 workgroups per CU on 256 CUs
constexpr int kRedThreads = 256;
int moments_all_launch(int dtype, const void *x, int64_t n, void *workspace, double *out, hipStream_t s);

} // namespace nfm

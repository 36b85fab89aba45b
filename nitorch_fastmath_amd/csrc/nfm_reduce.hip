// nfm_reduce.hip -- NaN-omitting reductions (reference `reduce.py`).
//
// Full reduction (`dim=None`): the reference makes up to four passes over memory
// (clone, isnan, masked_fill, sum: reduce.py:502-510); here it is ONE streaming pass.
//   kernel 1: 2048 workgroups x 256 lanes grid-stride over 16-byte vectors, four
//             independent loads in flight per lane, NaN -> identity by select,
//             per-lane accumulators (double for sums), 64-lane wavefront
//             shuffle-reduce, one LDS hop across the 4 waves, one partial per workgroup;
//   kernel 2: one workgroup folds the 2048 partials in the same fixed order.
// The launch geometry never depends on n, so results are bitwise reproducible.
#include "nfm_common.hpp"

namespace nfm {

constexpr int kRedBlocks = 2048; // 8 workgroups per CU on 256 CUs
constexpr int kRedThreads = 256;

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

template <int OP>
__device__ __forceinline__ double block_reduce(double v, double *lds)
{
    v = wave_reduce<OP>(v);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    constexpr int nw = kRedThreads / kWave;
    double r = RedOp<OP>::identity();
    if (threadIdx.x == 0) {
        r = lds[0];
#pragma unroll
        for (int w = 1; w < nw; ++w) r = RedOp<OP>::merge(r, lds[w]);
    }
    return r; // valid in thread 0
}

template <typename T, int OP>
__global__ __launch_bounds__(kRedThreads) void reduce_all_k1(const T *__restrict__ x, int64_t n,
                                                              double *__restrict__ partial)
{
    using V = typename VecOf<T>::type;
    constexpr int VEC = VecOf<T>::N;
    __shared__ double lds[kRedThreads / kWave];

    // split [0, n) into an unaligned head, 16-byte vectors, and a tail
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
    int64_t head = ((16 - (addr & 15)) & 15) / (int64_t)sizeof(T);
    if (head > n) head = n;
    const int64_t nvec = (n - head) / VEC;
    const int64_t tail0 = head + nvec * VEC;
    const V *xv = reinterpret_cast<const V *>(x + head);

    double a0 = RedOp<OP>::identity(), a1 = a0, a2 = a0, a3 = a0;
    const int64_t stride = (int64_t)gridDim.x * kRedThreads;
    int64_t q = (int64_t)blockIdx.x * kRedThreads + threadIdx.x;
    for (; q + 3 * stride < nvec; q += 4 * stride) {
        const V v0 = __builtin_nontemporal_load(xv + q);
        const V v1 = __builtin_nontemporal_load(xv + q + stride);
        const V v2 = __builtin_nontemporal_load(xv + q + 2 * stride);
        const V v3 = __builtin_nontemporal_load(xv + q + 3 * stride);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            RedOp<OP>::fold(a0, v0[k]);
            RedOp<OP>::fold(a1, v1[k]);
            RedOp<OP>::fold(a2, v2[k]);
            RedOp<OP>::fold(a3, v3[k]);
        }
    }
    for (; q < nvec; q += stride) {
        const V v0 = xv[q];
#pragma unroll
        for (int k = 0; k < VEC; ++k) RedOp<OP>::fold(a0, v0[k]);
    }
    // head and tail elements (fewer than 2 * VEC): one lane each, first workgroup
    const int64_t gid = (int64_t)blockIdx.x * kRedThreads + threadIdx.x;
    if (gid < head) RedOp<OP>::fold(a1, x[gid]);
    if (gid < n - tail0) RedOp<OP>::fold(a2, x[tail0 + gid]);

    double acc = RedOp<OP>::merge(RedOp<OP>::merge(a0, a1), RedOp<OP>::merge(a2, a3));
    acc = block_reduce<OP>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

template <int OP>
__global__ __launch_bounds__(kRedThreads) void reduce_all_k2(const double *__restrict__ partial, int nparts,
                                                              void *out, int out_dtype)
{
    __shared__ double lds[kRedThreads / kWave];
    double acc = RedOp<OP>::identity();
    for (int p = threadIdx.x; p < nparts; p += kRedThreads) acc = RedOp<OP>::merge(acc, partial[p]);
    acc = block_reduce<OP>(acc, lds);
    if (threadIdx.x == 0) {
        if (out_dtype == NFM_F32) *static_cast<float *>(out) = (float)acc;
        else *static_cast<double *>(out) = acc;
    }
}

// ---- (outer, red, inner) reductions ---------------------------------------------------
// inner == 1 and a long reduced axis: one wavefront per row (coalesced along the row);
// otherwise one lane per output element striding over `red` (coalesced along `inner`).
//
// max/min also track the position of the selected element: the FIRST occurrence of the
// extremum after NaN replacement (nan ops) or the first NaN (propagating ops), which is
// what torch.max/min(dim) return on the reference's path (reduce.py:129-140).
template <int OP>
struct Pick {
    static constexpr bool is_max = RedOp<OP>::is_max;
    static constexpr bool omit = OP == NFM_RED_NANMAX || OP == NFM_RED_NANMIN;
    // value as the reduction sees it
    __device__ static __forceinline__ double see(double v)
    {
        return (omit && v != v) ? RedOp<OP>::identity() : v;
    }
    // is candidate w strictly better than the current value?
    __device__ static __forceinline__ bool better(double w, double cur)
    {
        return cur == cur && (w != w || (is_max ? w > cur : w < cur));
    }
    __device__ static __forceinline__ bool same(double w, double cur)
    {
        return w == cur || (w != w && cur != cur);
    }
};

template <typename T, int OP, bool WAVE_PER_ROW>
__global__ __launch_bounds__(256) void reduce_dim_k(const T *__restrict__ x, int64_t outer, int64_t red,
                                                    int64_t inner, void *out, int out_dtype,
                                                    int64_t *__restrict__ idx)
{
    constexpr bool pick = !RedOp<OP>::is_sum;
    if constexpr (WAVE_PER_ROW) {
        const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kWave;
        const int lane = threadIdx.x % kWave;
        if (row >= outer) return; // whole wave exits together
        const T *p = x + row * red;
        double acc = RedOp<OP>::identity();
        int64_t best = -1;
        for (int64_t r = lane; r < red; r += kWave) {
            if constexpr (pick) {
                const double w = Pick<OP>::see((double)p[r]);
                if (best < 0 || Pick<OP>::better(w, acc)) { acc = w; best = r; }
            } else {
                RedOp<OP>::fold(acc, p[r]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(acc, off, kWave);
            if constexpr (pick) {
                const int64_t oi = __shfl_xor(best, off, kWave);
                const bool take = oi >= 0 && (best < 0 || Pick<OP>::better(ov, acc) ||
                                              (Pick<OP>::same(ov, acc) && oi < best));
                if (take) { acc = ov; best = oi; }
            } else {
                acc = RedOp<OP>::merge(acc, ov);
            }
        }
        if (lane == 0) {
            if (out_dtype == NFM_F32) static_cast<float *>(out)[row] = (float)acc;
            else static_cast<double *>(out)[row] = acc;
            if (pick && idx) idx[row] = best < 0 ? 0 : best;
        }
    } else {
        const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (e >= outer * inner) return;
        const int64_t o = e / inner, i = e - o * inner;
        const T *p = x + o * red * inner + i;
        double acc = RedOp<OP>::identity();
        int64_t best = -1;
        for (int64_t r = 0; r < red; ++r) {
            const T v = p[r * inner];
            if constexpr (pick) {
                const double w = Pick<OP>::see((double)v);
                if (best < 0 || Pick<OP>::better(w, acc)) { acc = w; best = r; }
            } else {
                RedOp<OP>::fold(acc, v);
            }
        }
        if (out_dtype == NFM_F32) static_cast<float *>(out)[e] = (float)acc;
        else static_cast<double *>(out)[e] = acc;
        if (pick && idx) idx[e] = best < 0 ? 0 : best;
    }
}

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

template <typename T>
__global__ __launch_bounds__(kRedThreads) void moments_all_k1(const T *__restrict__ x, int64_t n,
                                                               double *__restrict__ partial)
{
    using V = typename VecOf<T>::type;
    constexpr int VEC = VecOf<T>::N;
    __shared__ Mom lds[kRedThreads / kWave];
    const double shift = pick_shift(x, n, 1);
    // same streaming structure as reduce_all_k1: unaligned head, 16-byte vectors, tail
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
    int64_t head = ((16 - (addr & 15)) & 15) / (int64_t)sizeof(T);
    if (head > n) head = n;
    const int64_t nvec = (n - head) / VEC;
    const int64_t tail0 = head + nvec * VEC;
    const V *xv = reinterpret_cast<const V *>(x + head);
    Mom m = {0.0, 0.0, 0.0}, m1 = m, m2 = m, m3 = m;
    const int64_t stride = (int64_t)gridDim.x * kRedThreads;
    int64_t q = (int64_t)blockIdx.x * kRedThreads + threadIdx.x;
    for (; q + 3 * stride < nvec; q += 4 * stride) {
        const V v0 = __builtin_nontemporal_load(xv + q);
        const V v1 = __builtin_nontemporal_load(xv + q + stride);
        const V v2 = __builtin_nontemporal_load(xv + q + 2 * stride);
        const V v3 = __builtin_nontemporal_load(xv + q + 3 * stride);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            mom_fold(m, v0[k], shift);
            mom_fold(m1, v1[k], shift);
            mom_fold(m2, v2[k], shift);
            mom_fold(m3, v3[k], shift);
        }
    }
    for (; q < nvec; q += stride) {
        const V v0 = xv[q];
#pragma unroll
        for (int k = 0; k < VEC; ++k) mom_fold(m, v0[k], shift);
    }
    const int64_t gid = (int64_t)blockIdx.x * kRedThreads + threadIdx.x;
    if (gid < head) mom_fold(m1, x[gid], shift);
    if (gid < n - tail0) mom_fold(m2, x[tail0 + gid], shift);
    m = mom_merge(mom_merge(m, m1), mom_merge(m2, m3));
    m = mom_wave(m);
    if (threadIdx.x % kWave == 0) lds[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        Mom r = lds[0];
        for (int w = 1; w < kRedThreads / kWave; ++w) r = mom_merge(r, lds[w]);
        partial[3 * blockIdx.x + 0] = r.n;
        partial[3 * blockIdx.x + 1] = r.s;
        partial[3 * blockIdx.x + 2] = r.q;
    }
}

template <typename T>
__global__ __launch_bounds__(kRedThreads) void moments_all_k2(const double *__restrict__ partial, int nparts,
                                                               const T *__restrict__ x, int64_t n, double *out)
{
    __shared__ Mom lds[kRedThreads / kWave];
    Mom m = {0.0, 0.0, 0.0};
    for (int p = threadIdx.x; p < nparts; p += kRedThreads)
        m = mom_merge(m, Mom{partial[3 * p], partial[3 * p + 1], partial[3 * p + 2]});
    m = mom_wave(m);
    if (threadIdx.x % kWave == 0) lds[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        Mom r = lds[0];
        for (int w = 1; w < kRedThreads / kWave; ++w) r = mom_merge(r, lds[w]);
        out[0] = r.n;
        out[1] = r.s;
        out[2] = r.q;
        out[3] = pick_shift(x, n, 1);
    }
}

// (outer, red, inner) moments; out is (outer, inner, 4) doubles
template <typename T, bool WAVE_PER_ROW>
__global__ __launch_bounds__(256) void moments_dim_k(const T *__restrict__ x, int64_t outer, int64_t red,
                                                     int64_t inner, double *__restrict__ out)
{
    if constexpr (WAVE_PER_ROW) {
        const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kWave;
        const int lane = threadIdx.x % kWave;
        if (row >= outer) return;
        const T *p = x + row * red;
        const double shift = pick_shift(p, red, 1);
        Mom m = {0.0, 0.0, 0.0};
        for (int64_t r = lane; r < red; r += kWave) mom_fold(m, p[r], shift);
        m = mom_wave(m);
        if (lane == 0) {
            out[4 * row + 0] = m.n;
            out[4 * row + 1] = m.s;
            out[4 * row + 2] = m.q;
            out[4 * row + 3] = shift;
        }
    } else {
        const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (e >= outer * inner) return;
        const int64_t o = e / inner, i = e - o * inner;
        const T *p = x + o * red * inner + i;
        const double shift = pick_shift(p, red, inner);
        Mom m = {0.0, 0.0, 0.0};
        for (int64_t r = 0; r < red; ++r) mom_fold(m, p[r * inner], shift);
        out[4 * e + 0] = m.n;
        out[4 * e + 1] = m.s;
        out[4 * e + 2] = m.q;
        out[4 * e + 3] = shift;
    }
}

// ---- split reduction for "few outputs, long reduced axis" shapes: the reduced axis is cut
// into nchunk ranges (grid.y), one partial per (chunk, output) in the workspace, then a
// second tiny kernel folds the chunks in a fixed order (deterministic).
template <typename T, int OP>
__global__ __launch_bounds__(256) void reduce_dim_split_k1(const T *__restrict__ x, int64_t outer, int64_t red,
                                                           int64_t inner, int64_t chunk_len,
                                                           double *__restrict__ partial)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t c = blockIdx.y;
    const int64_t total = outer * inner;
    if (e >= total) return;
    const int64_t o = e / inner, i = e - o * inner;
    const T *p = x + o * red * inner + i;
    const int64_t r0 = c * chunk_len;
    int64_t r1 = r0 + chunk_len;
    if (r1 > red) r1 = red;
    double a0 = RedOp<OP>::identity(), a1 = a0;
    int64_t r = r0;
    for (; r + 1 < r1; r += 2) {
        RedOp<OP>::fold(a0, p[r * inner]);
        RedOp<OP>::fold(a1, p[(r + 1) * inner]);
    }
    if (r < r1) RedOp<OP>::fold(a0, p[r * inner]);
    partial[c * total + e] = RedOp<OP>::merge(a0, a1);
}

// inner == 1: the lanes of a wave walk one chunk of one row together (coalesced)
template <typename T, int OP>
__global__ __launch_bounds__(256) void reduce_row_split_k1(const T *__restrict__ x, int64_t outer, int64_t red,
                                                           int64_t chunk_len, int nchunk,
                                                           double *__restrict__ partial)
{
    const int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kWave; // (row, chunk) pair
    const int lane = threadIdx.x % kWave;
    if (w >= outer * nchunk) return;
    const int64_t row = w / nchunk, c = w - row * nchunk;
    const T *p = x + row * red;
    const int64_t r0 = c * chunk_len;
    int64_t r1 = r0 + chunk_len;
    if (r1 > red) r1 = red;
    double acc = RedOp<OP>::identity();
    for (int64_t r = r0 + lane; r < r1; r += kWave) RedOp<OP>::fold(acc, p[r]);
    acc = wave_reduce<OP>(acc);
    if (lane == 0) partial[c * outer + row] = acc;
}

template <int OP>
__global__ __launch_bounds__(256) void reduce_dim_split_k2(const double *__restrict__ partial, int64_t total,
                                                           int nchunk, void *out, int out_dtype)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    double acc = partial[e];
    for (int c = 1; c < nchunk; ++c) acc = RedOp<OP>::merge(acc, partial[(int64_t)c * total + e]);
    if (out_dtype == NFM_F32) static_cast<float *>(out)[e] = (float)acc;
    else static_cast<double *>(out)[e] = acc;
}

template <typename T, int OP>
static int reduce_dim_split_t(int out_dtype, int64_t outer, int64_t red, int64_t inner, int nchunk, const void *x,
                              double *ws, void *out, hipStream_t s)
{
    const T *xp = static_cast<const T *>(x);
    const int64_t total = outer * inner;
    const int64_t chunk_len = (red + nchunk - 1) / nchunk;
    if (inner == 1) {
        const int64_t nblk = (total * nchunk * kWave + 255) / 256;
        hipLaunchKernelGGL((reduce_row_split_k1<T, OP>), dim3((unsigned)nblk), dim3(256), 0, s, xp, outer, red,
                           chunk_len, nchunk, ws);
    } else {
        dim3 grid((unsigned)((total + 255) / 256), (unsigned)nchunk, 1);
        hipLaunchKernelGGL((reduce_dim_split_k1<T, OP>), grid, dim3(256), 0, s, xp, outer, red, inner, chunk_len, ws);
    }
    hipLaunchKernelGGL((reduce_dim_split_k2<OP>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, total,
                       nchunk, out, out_dtype);
    return launch_status();
}

template <typename T, int OP>
static int reduce_all_t(int out_dtype, int64_t n, const void *x, void *ws, void *out, hipStream_t s)
{
    double *partial = static_cast<double *>(ws);
    hipLaunchKernelGGL((reduce_all_k1<T, OP>), dim3(kRedBlocks), dim3(kRedThreads), 0, s,
                       static_cast<const T *>(x), n, partial);
    hipLaunchKernelGGL((reduce_all_k2<OP>), dim3(1), dim3(kRedThreads), 0, s, partial, kRedBlocks, out, out_dtype);
    return launch_status();
}

template <typename T, int OP>
static int reduce_dim_t(int out_dtype, int64_t outer, int64_t red, int64_t inner, const void *x, void *out,
                        int64_t *idx, hipStream_t s)
{
    const T *xp = static_cast<const T *>(x);
    if (inner == 1 && red >= 32) {
        const int64_t nblk = (outer * kWave + 255) / 256;
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
        hipLaunchKernelGGL((reduce_dim_k<T, OP, true>), dim3((unsigned)nblk), dim3(256), 0, s, xp, outer, red,
                           inner, out, out_dtype, idx);
    } else {
        const int64_t nblk = (outer * inner + 255) / 256;
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
        hipLaunchKernelGGL((reduce_dim_k<T, OP, false>), dim3((unsigned)nblk), dim3(256), 0, s, xp, outer, red,
                           inner, out, out_dtype, idx);
    }
    return launch_status();
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

} // namespace nfm

using namespace nfm;

extern "C" {

size_t nfm_reduce_workspace_bytes(void) { return (size_t)kRedBlocks * 3 * sizeof(double); }

int nfm_reduce_all(int dtype, int op, int out_dtype, int64_t n, const void *x, void *workspace,
                   size_t workspace_bytes, void *out, void *stream)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (out_dtype != NFM_F32 && out_dtype != NFM_F64) return NFM_EDTYPE;
    if (n < 0 || out == nullptr || workspace == nullptr || (n > 0 && x == nullptr)) return NFM_EINVAL;
    if (workspace_bytes < nfm_reduce_workspace_bytes()) return NFM_EWORKSPACE;
    if (reinterpret_cast<uintptr_t>(x) % (dtype == NFM_F32 ? 4 : 8) != 0) return NFM_EALIGN;
    if (reinterpret_cast<uintptr_t>(workspace) % 8 != 0) return NFM_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == NFM_F32) {
        NFM_SWITCH_OP(op, return (reduce_all_t<float, OP>(out_dtype, n, x, workspace, out, s)))
    } else {
        NFM_SWITCH_OP(op, return (reduce_all_t<double, OP>(out_dtype, n, x, workspace, out, s)))
    }
    return NFM_EINVAL;
}

int nfm_reduce_dim(int dtype, int op, int out_dtype, int64_t outer, int64_t red, int64_t inner, const void *x,
                   void *out, int64_t *idx, void *stream)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (out_dtype != NFM_F32 && out_dtype != NFM_F64) return NFM_EDTYPE;
    if (outer < 0 || red < 0 || inner < 0) return NFM_EINVAL;
    if (outer == 0 || inner == 0) return NFM_OK;
    if (out == nullptr || (red > 0 && x == nullptr)) return NFM_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == NFM_F32) {
        NFM_SWITCH_OP(op, return (reduce_dim_t<float, OP>(out_dtype, outer, red, inner, x, out, idx, s)))
    } else {
        NFM_SWITCH_OP(op, return (reduce_dim_t<double, OP>(out_dtype, outer, red, inner, x, out, idx, s)))
    }
    return NFM_EINVAL;
}

int nfm_reduce_dim_split(int dtype, int op, int out_dtype, int64_t outer, int64_t red, int64_t inner, int nchunk,
                         const void *x, void *workspace, size_t workspace_bytes, void *out, void *stream)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (out_dtype != NFM_F32 && out_dtype != NFM_F64) return NFM_EDTYPE;
    if (outer < 0 || red < 0 || inner < 0 || nchunk < 1 || nchunk > 65535) return NFM_EINVAL;
    if (outer == 0 || inner == 0) return NFM_OK;
    if (out == nullptr || workspace == nullptr || (red > 0 && x == nullptr)) return NFM_EINVAL;
    if (workspace_bytes < (size_t)nchunk * (size_t)(outer * inner) * sizeof(double)) return NFM_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double *ws = static_cast<double *>(workspace);
    if (dtype == NFM_F32) {
        NFM_SWITCH_OP(op, return (reduce_dim_split_t<float, OP>(out_dtype, outer, red, inner, nchunk, x, ws, out, s)))
    } else {
        NFM_SWITCH_OP(op, return (reduce_dim_split_t<double, OP>(out_dtype, outer, red, inner, nchunk, x, ws, out, s)))
    }
    return NFM_EINVAL;
}

int nfm_reduce_moments(int dtype, int64_t outer, int64_t red, int64_t inner, const void *x, void *workspace,
                       size_t workspace_bytes, double *out, void *stream)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (outer < 0 || red < 0 || inner < 0) return NFM_EINVAL;
    if (outer == 0 || inner == 0) return NFM_OK;
    if (out == nullptr || (red > 0 && x == nullptr)) return NFM_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (outer == 1 && inner == 1) { // full reduction: two-kernel streaming path
        if (workspace == nullptr) return NFM_EINVAL;
        if (workspace_bytes < nfm_reduce_workspace_bytes()) return NFM_EWORKSPACE;
        double *partial = static_cast<double *>(workspace);
        if (dtype == NFM_F32) {
            hipLaunchKernelGGL((moments_all_k1<float>), dim3(kRedBlocks), dim3(kRedThreads), 0, s,
                               static_cast<const float *>(x), red, partial);
            hipLaunchKernelGGL((moments_all_k2<float>), dim3(1), dim3(kRedThreads), 0, s, partial, kRedBlocks,
                               static_cast<const float *>(x), red, out);
        } else {
            hipLaunchKernelGGL((moments_all_k1<double>), dim3(kRedBlocks), dim3(kRedThreads), 0, s,
                               static_cast<const double *>(x), red, partial);
            hipLaunchKernelGGL((moments_all_k2<double>), dim3(1), dim3(kRedThreads), 0, s, partial, kRedBlocks,
                               static_cast<const double *>(x), red, out);
        }
        return launch_status();
    }
    const bool wpr = inner == 1 && red >= 32;
    const int64_t nblk = wpr ? (outer * kWave + 255) / 256 : (outer * inner + 255) / 256;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    if (dtype == NFM_F32) {
        if (wpr) hipLaunchKernelGGL((moments_dim_k<float, true>), dim3((unsigned)nblk), dim3(256), 0, s,
                                    static_cast<const float *>(x), outer, red, inner, out);
        else hipLaunchKernelGGL((moments_dim_k<float, false>), dim3((unsigned)nblk), dim3(256), 0, s,
                                static_cast<const float *>(x), outer, red, inner, out);
    } else {
        if (wpr) hipLaunchKernelGGL((moments_dim_k<double, true>), dim3((unsigned)nblk), dim3(256), 0, s,
                                    static_cast<const double *>(x), outer, red, inner, out);
        else hipLaunchKernelGGL((moments_dim_k<double, false>), dim3((unsigned)nblk), dim3(256), 0, s,
                                static_cast<const double *>(x), outer, red, inner, out);
    }
    return launch_status();
}

const char *nfm_strerror(int code)
{
    switch (code) {
    case NFM_OK: return "success";
    case NFM_EINVAL: return "invalid argument (null pointer, negative size or bad flag)";
    case NFM_EDTYPE: return "unsupported dtype code";
    case NFM_ESIZE: return "matrix order outside 1..16 or batch too large";
    case NFM_EALIGN: return "pointer not aligned to the element size";
    case NFM_EWORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown error";
    }
}

int nfm_version(void) { return NFM_VERSION; }

} // extern "C"

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
#include "nfm_reduce_common.hpp"

namespace nfm {


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

template <typename T, int OP>
static int reduce_all_t(int out_dtype, int64_t n, const void *x, void *ws, void *out, hipStream_t s)
{
    double *partial = static_cast<double *>(ws);
    hipLaunchKernelGGL((reduce_all_k1<T, OP>), dim3(kRedBlocks), dim3(kRedThreads), 0, s,
                       static_cast<const T *>(x), n, partial);
    hipLaunchKernelGGL((reduce_all_k2<OP>), dim3(1), dim3(kRedThreads), 0, s, partial, kRedBlocks, out, out_dtype);
    return launch_status();
}

// full-reduction moments (called by nfm_reduce_moments / nfm_reduce_stat in nfm_reduce_dim.hip)
int moments_all_launch(int dtype, const void *x, int64_t n, void *workspace, double *out, hipStream_t s)
{
    double *partial = static_cast<double *>(workspace);
    if (dtype == NFM_F32) {
        hipLaunchKernelGGL((moments_all_k1<float>), dim3(kRedBlocks), dim3(kRedThreads), 0, s,
                           static_cast<const float *>(x), n, partial);
        hipLaunchKernelGGL((moments_all_k2<float>), dim3(1), dim3(kRedThreads), 0, s, partial, kRedBlocks,
                           static_cast<const float *>(x), n, out);
    } else {
        hipLaunchKernelGGL((moments_all_k1<double>), dim3(kRedBlocks), dim3(kRedThreads), 0, s,
                           static_cast<const double *>(x), n, partial);
        hipLaunchKernelGGL((moments_all_k2<double>), dim3(1), dim3(kRedThreads), 0, s, partial, kRedBlocks,
                           static_cast<const double *>(x), n, out);
    }
    return launch_status();
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

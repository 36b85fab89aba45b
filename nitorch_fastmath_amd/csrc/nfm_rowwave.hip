// nfm_rowwave.hip -- orders 9..16 with ONE MATRIX PER 16 LANES (one row per lane).
//
// A 16x16 float64 matrix is 512 dwords: the whole register file of a lane.  The lane-per-matrix
// kernels of nfm_large.hip therefore spill at float64 orders 14..16 (0.4-1.5 TB/s) and run at one
// wave per SIMD long before that.  Here a matrix is spread over the 16 lanes of a DPP row: lane r
// holds row r (N values = 2N dwords), a wavefront works on 4 matrices, a 256-lane workgroup on 16.
//
//   * pivot search: max over the unused rows of an integer key that orders |a_rk| = 4 DPP row
//     rotations (row_ror 8/4/2/1) with a v_max_u32 each; the pivot lane is the lowest set bit of
//     the group's 16 bits of a ballot;
//   * elimination: Gauss-Jordan WITHOUT row exchanges -- the pivot row stays in its lane and is
//     broadcast value by value with ds_bpermute (the LDS crossbar, no LDS memory), every other
//     lane updates its row with one fma per value.  All 16 lanes work in every step, so the
//     Gauss-Jordan form (no back substitution) is free;
//   * inverse: the in-place Gauss-Jordan on [A | I], rows divided by their pivots at the end; lane
//     p_k ends up with row k of A^-1 whose l-th entry belongs to column p_l (p = the pivot lane
//     sequence), undone while writing the LDS image;
//   * HBM traffic: the tile's records are contiguous, so they are streamed with 16-byte accesses
//     through an LDS image (rows padded to an odd number of 16-byte slots: conflict-free
//     ds_read_b128 / ds_write_b128 by 16 lanes holding 16 rows), exactly the algorithmic bytes.
//
// 1 / pivot is v_rcp + Newton steps (2 for float64, 1 for float32: <= 1.5 ulp) instead of the
// IEEE division sequence; pivots are the same as the CPU restatement's (partial pivoting, first
// maximum), the rounding differs -- parity is asserted through the error model of the tests
// (tests/test_gpu_large_orders.py), like every LU-based result beyond the closed forms.
//
// Reference paths replaced: `torch.linalg.solve` of the densified matrix (`_impl/sym.py:392-396`),
// `a.inverse()` / `a.det()` (`_impl/batched.py:119-120`, `:53-54`).
#include <stdlib.h>
#include "nfm_common.hpp"
#include "nfm_smallmat.hpp"
#include "nfm_rowwave.hpp"

namespace nfm {
namespace roww {

constexpr int G = 16;    // lanes per matrix = one DPP row
constexpr int MPB = 16;  // matrices per 256-lane workgroup

enum { RW_SOLVE_SYM = 0, RW_INV_SYM, RW_INVDIAG_SYM, RW_DET_SYM, RW_INV_GEN, RW_DET_GEN };

struct RowParams {
    int has_eps;
    double eps[NFM_MAX_DIM];
};

// row stride of the N x N LDS image in elements: a whole, odd number of 16-byte slots
template <typename T, int N>
struct RowStride {
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int slots = (N + V - 1) / V;
    static constexpr int value = ((slots & 1) ? slots : slots + 1) * V;
};

__device__ __forceinline__ int bperm(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float bcast(float v, int src_lane) { return __int_as_float(bperm(src_lane, __float_as_int(v))); }
__device__ __forceinline__ double bcast(double v, int src_lane)
{
    const int lo = bperm(src_lane, __double2loint(v)), hi = bperm(src_lane, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int ROR>
__device__ __forceinline__ int dpp_ror(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + ROR, 0xf, 0xf, false); // row_ror:ROR
}
// Pivot search key: an unsigned integer that orders |x| -- float32: the bits of |x|; float64: the
// high dword of |x| (sign cleared: 11 exponent + 20 mantissa bits, the whole exponent range at a
// resolution of 2^-20; rows whose |a_rk| agree to 1e-6 tie and the first one pivots, which is as
// good a partial pivot).  +1 so that 0 is left for rows that may not pivot.  A NaN has the
// largest key: it pivots, and the result is NaN as it would be anyway.
__device__ __forceinline__ unsigned pivot_key(float x) { return ((unsigned)__float_as_int(x) & 0x7fffffffu) + 1u; }
__device__ __forceinline__ unsigned pivot_key(double x) { return ((unsigned)__double2hiint(x) & 0x7fffffffu) + 1u; }
template <int ROR>
__device__ __forceinline__ unsigned rowmax_step(unsigned v)
{
    const unsigned o = (unsigned)dpp_ror<ROR>((int)v);
    return v > o ? v : o;
}
// max over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ unsigned rowmax(unsigned v)
{
    v = rowmax_step<8>(v);
    v = rowmax_step<4>(v);
    v = rowmax_step<2>(v);
    return rowmax_step<1>(v);
}

__device__ __forceinline__ float recip(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double recip(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
// 1 / pivot; a zero / non-finite pivot goes through the IEEE division so that singular input
// yields inf / NaN exactly like a division would
template <typename T>
__device__ __forceinline__ T pivot_recip(T pv)
{
    const T a = fabs_(pv);
    if (__builtin_expect(!(a > T(1e-30) && a < T(1e30)), 0)) return T(1) / pv;
    return recip(pv);
}
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// compact-sym index of (i, j), `sym.py:7-14`
__device__ __forceinline__ int cidx(int N, int i, int j)
{
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return i == j ? i : N + lo * N - (lo * (lo + 1)) / 2 + (hi - lo - 1);
}

template <typename T, int N, int OP>
__global__ __launch_bounds__(256) void roww_kernel(const T *__restrict__ A, const T *__restrict__ B,
                                                   T *__restrict__ O, int64_t n, RowParams p)
{
    constexpr bool SYM = OP == RW_SOLVE_SYM || OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_DET_SYM;
    constexpr bool INV = OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_INV_GEN;
    constexpr bool DET = OP == RW_DET_SYM || OP == RW_DET_GEN;
    constexpr int K = N * (N + 1) / 2;
    constexpr int RIN = SYM ? K : N * N;                                    // input record
    constexpr int ROUT = OP == RW_SOLVE_SYM ? N : OP == RW_INV_SYM ? K : OP == RW_INVDIAG_SYM ? N : DET ? 1 : N * N;
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int RS = RowStride<T, N>::value;
    using Vec = T __attribute__((ext_vector_type(V)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *img = reinterpret_cast<T *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = tid / G, r = tid % G;
    const int gbase = lane & 48; // first lane of this matrix's DPP row within the wavefront
    const int64_t m0 = (int64_t)blockIdx.x * MPB;
    const int nm = (int)((n - m0) < MPB ? (n - m0) : MPB);

    // ---- stream the tile's contiguous input records into LDS
    {
        const T *src = A + m0 * RIN;
        const int total = nm * RIN;
        // a tile starts a whole number of 16-matrix blocks into the operand: 16-byte aligned exactly
        // when the operand's base is (row slices x[i:] of float64 tensors may not be)
        const bool vec_ok = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        for (int e = tid * V; e < total; e += 256 * V) {
            if (vec_ok && e + V <= total) {
                const Vec v = NFM_LDG(reinterpret_cast<const Vec *>(src + e));
                if constexpr (SYM) { // flat copy of the compact records
                    *reinterpret_cast<Vec *>(img + e) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) {
                        const int ee = e + q, m = ee / (N * N), rem = ee - m * (N * N), i = rem / N, j = rem - i * N;
                        img[(m * N + i) * RS + j] = v[q];
                    }
                }
            } else {
                for (int ee = e; ee < total && ee < e + V; ++ee) {
                    const T x = NFM_LDG(src + ee);
                    if constexpr (SYM) img[ee] = x;
                    else {
                        const int m = ee / (N * N), rem = ee - m * (N * N), i = rem / N, j = rem - i * N;
                        img[(m * N + i) * RS + j] = x;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- my row
    const bool live = r < N && g < nm;
    T row[N];
    if constexpr (SYM) {
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] = live ? img[g * K + cidx(N, r, j)] : T(0);
        if (OP == RW_SOLVE_SYM && p.has_eps) {
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (r == j) row[j] += (T)p.eps[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] = live ? img[(g * N + r) * RS + j] : T(0);
    }
    T rhs = T(0);
    if constexpr (OP == RW_SOLVE_SYM) rhs = live ? NFM_LDG(B + (m0 + g) * N + r) : T(0);
    // rows beyond the order (and matrices beyond the batch) never pivot
    bool used = !(r < N);
    if (!(g < nm)) { // idle matrices of a ragged last tile: the identity, so that nothing divides by zero
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] = (r == j) ? T(1) : T(0);
    }
    int ppos = -1;   // the step at which my row was the pivot row = the logical row it holds
    int src_of = 0;  // lane (within the DPP row) that pivoted at step r: where output row r lives
    int pl_of[INV ? N : 1]; // inverse: pivot lane of every step (the column permutation)
    T det = T(1);
    int inversions = 0;

    T mypv = T(1); // inverse: the pivot of my row (rows are scaled once, at the end)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // -- pivot: first unused row with the largest |a_rk| (pivot_key above)
        const unsigned key = used ? 0u : pivot_key(row[k]);
        const unsigned mx = rowmax(key);
        const unsigned grp = (unsigned)(__ballot(key == mx) >> gbase) & 0xffffu; // never 0: some row is unused
        const int pl = __builtin_ctz(grp | 0x10000u);
        const bool isp = r == pl;
        if (r == k) src_of = pl;
        if constexpr (INV) pl_of[k] = pl;
        if constexpr (DET) { // parity of the row permutation: unused rows above the pivot row
            const unsigned un = (unsigned)(__ballot(!used) >> gbase) & 0xffffu;
            inversions += __builtin_popcount(un & ((1u << pl) - 1u));
        }
        const int psrc = gbase + pl;
        const T pv = bcast(row[k], psrc);
        // multiplier of my row; 0 in the pivot row itself, so that the same fma leaves it unchanged
        // (a determinant with a zero pivot is 0 whatever follows: no elimination then)
        T f = row[k] * pivot_recip(pv);
        f = isp ? T(0) : f;
        if constexpr (DET) {
            det *= pv;
            f = (pv == T(0)) ? T(0) : f;
        }
        if constexpr (INV) {
            // in-place Gauss-Jordan on [A | I] WITHOUT scaling the pivot row (rows are divided by their
            // pivots at the end): column k of A is spent, its slot takes the column of the right half
            // that becomes non-trivial in this step -- 1 in the pivot row, -f elsewhere
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j == k) continue;
                row[j] = fma_(-f, bcast(row[j], psrc), row[j]);
            }
            row[k] = isp ? T(1) : -f;
            mypv = isp ? pv : mypv;
        } else {
            // Gauss-Jordan on [A | b] (solve) or plain elimination (det): columns k+1.. only
#pragma unroll
            for (int j = k + 1; j < N; ++j) row[j] = fma_(-f, bcast(row[j], psrc), row[j]);
            if constexpr (OP == RW_SOLVE_SYM) rhs = fma_(-f, bcast(rhs, psrc), rhs);
        }
        if (isp) {
            used = true;
            ppos = k;
        }
    }
    if constexpr (INV) {
        const T rp = T(1) / mypv;
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] *= rp;
    }

    // ---- results
    if constexpr (OP == RW_SOLVE_SYM) {
        // my row holds x_ppos = rhs / pivot (its pivot is still at column ppos); output row r lives in lane src_of
        T piv = T(1);
#pragma unroll
        for (int j = 0; j < N; ++j) piv = (ppos == j) ? row[j] : piv;
        const T x = rhs / piv;
        const T xr = bcast(x, gbase + src_of);
        if (live) NFM_STG(xr, O + (m0 + g) * N + r);
    } else if constexpr (DET) {
        const T d = (inversions & 1) ? -det : det;
        if (r == 0 && g < nm) NFM_STG(d, O + (m0 + g));
    } else {
        // lane p_k holds row k = ppos of A^-1; its l-th value is column pl_of[l]
        __syncthreads(); // everyone is done reading the input image
        if (r < N && g < nm && ppos >= 0) {
            if constexpr (OP == RW_INV_GEN) {
#pragma unroll
                for (int l = 0; l < N; ++l) img[(g * N + ppos) * RS + pl_of[l]] = row[l];
            } else if constexpr (OP == RW_INV_SYM) {
#pragma unroll
                for (int l = 0; l < N; ++l)
                    if (pl_of[l] >= ppos) img[g * K + cidx(N, ppos, pl_of[l])] = row[l];
            } else { // diagonal only
#pragma unroll
                for (int l = 0; l < N; ++l)
                    if (pl_of[l] == ppos) img[g * N + ppos] = row[l];
            }
        }
        __syncthreads();
        T *dst = O + m0 * ROUT;
        const int total = nm * ROUT;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
        for (int e = tid * V; e < total; e += 256 * V) {
            T tmp[V];
            if constexpr (OP == RW_INV_GEN) {
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    const int ee = e + q, m = ee / (N * N), rem = ee - m * (N * N), i = rem / N, j = rem - i * N;
                    tmp[q] = (ee < total) ? img[(m * N + i) * RS + j] : T(0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < V; ++q) tmp[q] = (e + q < total) ? img[e + q] : T(0);
            }
            if (vec_ok && e + V <= total) {
                Vec v;
#pragma unroll
                for (int q = 0; q < V; ++q) v[q] = tmp[q];
                NFM_STG(v, reinterpret_cast<Vec *>(dst + e));
            } else {
#pragma unroll
                for (int q = 0; q < V; ++q)
                    if (e + q < total) NFM_STG(tmp[q], dst + e + q);
            }
        }
    }
}

// contiguous batch-major records (what the facade allocates; the base need not be 16-byte aligned)
static bool rec_contig(const nfm_operand *o, int64_t rec, int rows, int cols, size_t elem)
{
    if (o == nullptr || o->ptr == nullptr) return false;
    if (reinterpret_cast<uintptr_t>(o->ptr) % elem != 0) return false;
    if (o->stride_inner != rec) return false;
    if (cols > 1 && o->stride_col != 1) return false;
    if (rows > 1 && o->stride_row != cols) return false;
    return true;
}

template <typename T, int N, int OP>
static int launch(const void *a, const void *b, void *o, int64_t n, const RowParams &p, void *stream)
{
    constexpr bool SYM = OP == RW_SOLVE_SYM || OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_DET_SYM;
    constexpr int K = N * (N + 1) / 2;
    constexpr int RS = RowStride<T, N>::value;
    constexpr size_t lds = (SYM ? (size_t)MPB * K + 16 : (size_t)MPB * N * RS) * sizeof(T);
    static_assert(lds <= 64 * 1024, "row-wave tile must fit the default dynamic LDS limit");
    if (n == 0) return NFM_OK;
    const int64_t nblk = (n + MPB - 1) / MPB;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    hipLaunchKernelGGL((roww_kernel<T, N, OP>), dim3((unsigned)nblk), dim3(256), lds, static_cast<hipStream_t>(stream),
                       static_cast<const T *>(a), static_cast<const T *>(b), static_cast<T *>(o), n, p);
    return launch_status();
}

#define NFM_RW_SWITCH(Nexpr, ...)                                                                  \
    switch (Nexpr) {                                                                               \
    case 9: { constexpr int N = 9; __VA_ARGS__; } break;                                           \
    case 10: { constexpr int N = 10; __VA_ARGS__; } break;                                         \
    case 11: { constexpr int N = 11; __VA_ARGS__; } break;                                         \
    case 12: { constexpr int N = 12; __VA_ARGS__; } break;                                         \
    case 13: { constexpr int N = 13; __VA_ARGS__; } break;                                         \
    case 14: { constexpr int N = 14; __VA_ARGS__; } break;                                         \
    case 15: { constexpr int N = 15; __VA_ARGS__; } break;                                         \
    case 16: { constexpr int N = 16; __VA_ARGS__; } break;                                         \
    default: break;                                                                                \
    }

} // namespace roww

using namespace roww;

template <typename T>
int RowWave<T>::sym_solve(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                          const double *eps, void *stream)
{
    const int K = M * (M + 1) / 2;
    if (!rec_contig(mat, K, 1, K, sizeof(T)) || !rec_contig(vec, M, 1, M, sizeof(T)) ||
        !rec_contig(out, M, 1, M, sizeof(T)))
        return NFM_EFALLBACK_RW;
    RowParams p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? eps[i] : 0.0;
    NFM_RW_SWITCH(M, return (launch<T, N, RW_SOLVE_SYM>(mat->ptr, vec->ptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::sym_invert(int M, int diag_only, int64_t ni, const nfm_operand *mat, const nfm_operand *out,
                           void *stream)
{
    const int K = M * (M + 1) / 2;
    const int RO = diag_only ? M : K;
    if (!rec_contig(mat, K, 1, K, sizeof(T)) || !rec_contig(out, RO, 1, RO, sizeof(T))) return NFM_EFALLBACK_RW;
    RowParams p{};
    if (diag_only) {
        NFM_RW_SWITCH(M, return (launch<T, N, RW_INVDIAG_SYM>(mat->ptr, nullptr, out->ptr, ni, p, stream)))
    } else {
        NFM_RW_SWITCH(M, return (launch<T, N, RW_INV_SYM>(mat->ptr, nullptr, out->ptr, ni, p, stream)))
    }
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::sym_det(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    const int K = M * (M + 1) / 2;
    if (!rec_contig(mat, K, 1, K, sizeof(T)) || out == nullptr || out->ptr == nullptr || out->stride_inner != 1)
        return NFM_EFALLBACK_RW;
    RowParams p{};
    NFM_RW_SWITCH(M, return (launch<T, N, RW_DET_SYM>(mat->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::batch_inv(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (!rec_contig(a, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)) || !rec_contig(out, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)))
        return NFM_EFALLBACK_RW;
    RowParams p{};
    NFM_RW_SWITCH(Nn, return (launch<T, N, RW_INV_GEN>(a->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::batch_det(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (!rec_contig(a, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)) || out == nullptr || out->ptr == nullptr ||
        out->stride_inner != 1)
        return NFM_EFALLBACK_RW;
    RowParams p{};
    NFM_RW_SWITCH(Nn, return (launch<T, N, RW_DET_GEN>(a->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

#if NFM_ROWW_F64
template struct RowWave<double>;
#else
template struct RowWave<float>;

// defaults: measured cross-over orders (profiles/r02/rowwave_table.md)
int rowwave_min_order(int is_f64, int what)
{
    static const int env64 = [] { const char *e = getenv("NFM_ROWWAVE_MIN_F64"); return e ? atoi(e) : 0; }();
    static const int env32 = [] { const char *e = getenv("NFM_ROWWAVE_MIN_F32"); return e ? atoi(e) : 0; }();
    if (what == RWW_INVDIAG_SYM) return 9;
    if (is_f64 ? env64 : env32) return is_f64 ? env64 : env32;
    //                                   solve inv_sym diag det_sym inv_gen det_gen
    static const int min64[6] = {14, 14, 9, 15, 13, 13};
    static const int min32[6] = {17, 17, 9, 17, 15, 17};
    return is_f64 ? min64[what] : min32[what];
}
#endif

} // namespace nfm

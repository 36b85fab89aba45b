// nfm_big.hpp -- orders 9..16 ("the larger tiles"): one matrix per lane still, but the
// N x N working matrix lives in LDS instead of VGPRs (a 16x16 fp64 matrix is 512
// registers).  Layout is lane-interleaved, element e of lane l at word e*64 + l, so a
// wave access touches 64 consecutive words (bank = lane, conflict-free) and rows can be
// indexed at run time -- pivoting is a plain loop here, not a select network.
// One wave (64 lanes) per workgroup; LDS = (N*N + 2N) * 64 * sizeof(T) <= 147 KiB.
// Operands are read/written directly with their strides (correct for every layout;
// these orders are not on the headline path).
#pragma once
#include "nfm_common.hpp"
#include "nfm_smallmat.hpp"

namespace nfm {

template <typename T>
struct LaneMem {
    T *base;
    __device__ __forceinline__ T &operator[](int e) const { return base[e * kWave]; }
};

template <typename T>
__device__ inline const T *opnd_ptr(const Opnd &op, int64_t o, int64_t i)
{
    return reinterpret_cast<const T *>(op.ptr) + o * op.so + i * op.si;
}
template <typename T>
__device__ inline T *opnd_ptr_w(const Opnd &op, int64_t o, int64_t i)
{
    return reinterpret_cast<T *>(op.ptr) + o * op.so + i * op.si;
}

// load a matrix of the given kind as a full N x N into LDS
template <typename T>
__device__ inline void big_load_full(int N, int kind, const Opnd &mat, int64_t o, int64_t i, LaneMem<T> A)
{
    const T *p = opnd_ptr<T>(mat, o, i);
    if (kind == NFM_MAT_FULL) {
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < N; ++c) A[r * N + c] = p[r * mat.sr + c * mat.sc];
    } else { // compact symmetric
        for (int r = 0; r < N; ++r) A[r * N + r] = p[r * mat.sc];
        int k = N;
        for (int r = 0; r < N; ++r)
            for (int c = r + 1; c < N; ++c, ++k) {
                const T x = p[k * mat.sc];
                A[r * N + c] = x;
                A[c * N + r] = x;
            }
    }
}

// LU with partial pivoting in LDS; piv[k] = row exchanged with k; returns the sign
template <typename T>
__device__ inline T big_lu(int N, LaneMem<T> A, LaneMem<T> piv)
{
    T sign = T(1);
    for (int k = 0; k < N; ++k) {
        int p = k;
        T best = fabs_(A[k * N + k]);
        for (int r = k + 1; r < N; ++r) {
            const T x = fabs_(A[r * N + k]);
            if (x > best) { best = x; p = r; }
        }
        piv[k] = (T)p;
        if (p != k) {
            for (int c = 0; c < N; ++c) {
                const T t = A[k * N + c];
                A[k * N + c] = A[p * N + c];
                A[p * N + c] = t;
            }
            sign = -sign;
        }
        const T rp = T(1) / A[k * N + k];
        for (int r = k + 1; r < N; ++r) {
            const T l = A[r * N + k] * rp;
            A[r * N + k] = l;
            for (int c = k + 1; c < N; ++c) A[r * N + c] -= l * A[k * N + c];
        }
    }
    return sign;
}

template <typename T>
__device__ inline void big_lu_solve(int N, LaneMem<T> A, LaneMem<T> piv, LaneMem<T> b)
{
    for (int k = 0; k < N; ++k) {
        const int p = (int)piv[k];
        if (p != k) { const T t = b[k]; b[k] = b[p]; b[p] = t; }
    }
    for (int r = 1; r < N; ++r) {
        T s = b[r];
        for (int c = 0; c < r; ++c) s -= A[r * N + c] * b[c];
        b[r] = s;
    }
    for (int r = N - 1; r >= 0; --r) {
        T s = b[r];
        for (int c = r + 1; c < N; ++c) s -= A[r * N + c] * b[c];
        b[r] = s / A[r * N + r];
    }
}

struct BigArgs {
    Opnd a, b, c, out;
    int64_t n_inner;
    int N, N2, kind, mode;
    double eps[NFM_MAX_DIM];
    int has_eps;
};

// op codes for the single "big" kernel
enum { BIG_SOLVE = 0, BIG_INVERT = 1, BIG_DET = 2, BIG_GINV = 3, BIG_GDET = 4 };

template <typename T, int OP>
__global__ __launch_bounds__(kWave) void big_lu_kernel(BigArgs g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kWave + lane;
    const int64_t o = blockIdx.y;
    if (i >= g.n_inner) return; // no barriers in this kernel: lanes are independent
    const int N = g.N;
    T *base = reinterpret_cast<T *>(smem) + lane;
    LaneMem<T> A{base};
    LaneMem<T> piv{base + (size_t)N * N * kWave};
    LaneMem<T> b{base + (size_t)(N * N + N) * kWave};

    if (OP == BIG_SOLVE) {
        big_load_full<T>(N, g.kind, g.a, o, i, A);
        if (g.has_eps)
            for (int r = 0; r < N; ++r) A[r * N + r] += (T)g.eps[r];
        const T *v = opnd_ptr<T>(g.b, o, i);
        for (int r = 0; r < N; ++r) b[r] = v[r * g.b.sc];
        big_lu<T>(N, A, piv);
        big_lu_solve<T>(N, A, piv, b);
        T *x = opnd_ptr_w<T>(g.out, o, i);
        for (int r = 0; r < N; ++r) x[r * g.out.sc] = b[r];
    } else if (OP == BIG_INVERT || OP == BIG_GINV) {
        big_load_full<T>(N, OP == BIG_INVERT ? NFM_MAT_SYM : NFM_MAT_FULL, g.a, o, i, A);
        big_lu<T>(N, A, piv);
        T *x = opnd_ptr_w<T>(g.out, o, i);
        // column c of the inverse = solve against e_c (what the reference does, once per column)
        for (int c = 0; c < N; ++c) {
            for (int r = 0; r < N; ++r) b[r] = T(r == c ? 1 : 0);
            big_lu_solve<T>(N, A, piv, b);
            if (OP == BIG_GINV) {
                for (int r = 0; r < N; ++r) x[r * g.out.sr + c * g.out.sc] = b[r];
            } else {
                x[c * g.out.sc] = b[c];
                if (!g.mode) // mode != 0: diagonal only
                    for (int r = c + 1; r < N; ++r) x[sym_idx(N, c, r) * g.out.sc] = b[r];
            }
        }
    } else { // determinants
        big_load_full<T>(N, OP == BIG_DET ? NFM_MAT_SYM : NFM_MAT_FULL, g.a, o, i, A);
        T d = big_lu<T>(N, A, piv);
        for (int r = 0; r < N; ++r) d *= A[r * N + r];
        *opnd_ptr_w<T>(g.out, o, i) = d;
    }
}

// element-wise style ops that need no working matrix: direct strided global access
enum { BIGE_MATVEC = 0, BIGE_TOFULL = 1, BIGE_OUTER = 2, BIGE_DIVDIAG = 3, BIGE_GMATVEC = 4, BIGE_MATMUL = 5, BIGE_OUTER2 = 6 };

template <typename T, int OP>
__global__ __launch_bounds__(256) void big_elem_kernel(BigArgs g)
{
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t o = blockIdx.y;
    if (i >= g.n_inner) return;
    const int N = g.N;
    if (OP == BIGE_MATVEC) {
        const T *m = opnd_ptr<T>(g.a, o, i);
        const T *v = opnd_ptr<T>(g.b, o, i);
        const T *inp = g.mode ? opnd_ptr<T>(g.c, o, i) : nullptr;
        T *y = opnd_ptr_w<T>(g.out, o, i);
        for (int r = 0; r < N; ++r) {
            T s;
            if (g.kind == NFM_MAT_SYM) {
                // same per-row fma chain as _sym_matvecn (_impl/sym.py:122-131) produces
                s = m[r * g.a.sc] * v[r * g.b.sc];
                for (int c = 0; c < N; ++c)
                    if (c != r) s = fma_(m[sym_idx(N, r, c) * g.a.sc], v[c * g.b.sc], s);
            } else if (g.kind == NFM_MAT_DIAG) {
                s = m[r * g.a.sc] * v[r * g.b.sc];
            } else if (g.kind == NFM_MAT_SCAL) {
                s = m[0] * v[r * g.b.sc];
            } else {
                s = m[r * g.a.sr] * v[0];
                for (int c = 1; c < N; ++c) s = s + m[r * g.a.sr + c * g.a.sc] * v[c * g.b.sc];
            }
            const T in = g.mode ? inp[r * g.c.sc] : T(0);
            // NB: out may alias inp (in-place add/sub) but never vec
            y[r * g.out.sc] = g.mode > 0 ? in + s : (g.mode < 0 ? in - s : s);
        }
    } else if (OP == BIGE_GMATVEC) { // (N x N2) general matrix times vector
        const T *m = opnd_ptr<T>(g.a, o, i);
        const T *v = opnd_ptr<T>(g.b, o, i);
        T *y = opnd_ptr_w<T>(g.out, o, i);
        for (int r = 0; r < N; ++r) {
            T s = m[r * g.a.sr] * v[0];
            for (int c = 1; c < g.N2; ++c) s = s + m[r * g.a.sr + c * g.a.sc] * v[c * g.b.sc];
            y[r * g.out.sc] = s;
        }
    } else if (OP == BIGE_TOFULL) {
        const T *m = opnd_ptr<T>(g.a, o, i);
        T *f = opnd_ptr_w<T>(g.out, o, i);
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < N; ++c) f[r * g.out.sr + c * g.out.sc] = m[sym_idx(N, r, c) * g.a.sc];
    } else if (OP == BIGE_OUTER) {
        const T *x = opnd_ptr<T>(g.a, o, i);
        T *f = opnd_ptr_w<T>(g.out, o, i);
        for (int r = 0; r < N; ++r)
            for (int c = r; c < N; ++c) f[sym_idx(N, r, c) * g.out.sc] = x[r * g.a.sc] * x[c * g.a.sc];
    } else if (OP == BIGE_OUTER2) { // x y^T + y x^T pulled back onto compact storage
        const T *x = opnd_ptr<T>(g.a, o, i);
        const T *y = opnd_ptr<T>(g.b, o, i);
        T *f = opnd_ptr_w<T>(g.out, o, i);
        for (int r = 0; r < N; ++r)
            for (int c = r; c < N; ++c) {
                const T v = (r == c) ? x[r * g.a.sc] * y[r * g.b.sc]
                                     : x[r * g.a.sc] * y[c * g.b.sc] + x[c * g.a.sc] * y[r * g.b.sc];
                f[sym_idx(N, r, c) * g.out.sc] = g.mode ? -v : v;
            }
    } else if (OP == BIGE_DIVDIAG) { // solve with a diagonal / scaled-identity matrix
        const T *m = opnd_ptr<T>(g.a, o, i);
        const T *v = opnd_ptr<T>(g.b, o, i);
        T *x = opnd_ptr_w<T>(g.out, o, i);
        for (int r = 0; r < N; ++r) {
            T d = g.kind == NFM_MAT_DIAG ? m[r * g.a.sc] : m[0];
            if (g.has_eps) d += (T)g.eps[r];
            x[r * g.out.sc] = v[r * g.b.sc] / d;
        }
    } else if (OP == BIGE_MATMUL) { // jhjn, K = N rows, D = N2 columns
        const int K = N, D = g.N2;
        const T *J = opnd_ptr<T>(g.a, o, i);
        const T *H = opnd_ptr<T>(g.b, o, i);
        T *out = opnd_ptr_w<T>(g.out, o, i);
        const bool hs = g.kind == NFM_MAT_SYM;
#define J_(k, d) J[(k) * g.a.sr + (d) * g.a.sc]
        for (int d = 0; d < D; ++d) {
            T acc = T(0);
            for (int k = 0; k < K; ++k) {
                acc += H[k * g.b.sc] * (J_(k, d) * J_(k, d));
                if (hs)
                    for (int l = k + 1; l < K; ++l)
                        acc += ((T(2) * H[sym_idx(K, k, l) * g.b.sc]) * J_(k, d)) * J_(l, d);
            }
            out[d * g.out.sc] = acc;
            for (int e = d + 1; e < D; ++e) {
                T ac = T(0);
                for (int k = 0; k < K; ++k) {
                    ac += (H[k * g.b.sc] * J_(k, d)) * J_(k, e);
                    if (hs)
                        for (int l = k + 1; l < K; ++l)
                            ac += H[sym_idx(K, k, l) * g.b.sc] * (J_(k, d) * J_(l, e) + J_(l, d) * J_(k, e));
                }
                out[sym_idx(D, d, e) * g.out.sc] = ac;
            }
        }
#undef J_
    }
}

inline BigArgs big_args(const nfm_operand *a, const nfm_operand *b, const nfm_operand *c, const nfm_operand *out,
                        int64_t ni, int N, int N2, int kind, int mode, const double *eps)
{
    BigArgs g;
    nfm_operand none = {nullptr, 0, 0, 0, 0};
    g.a = make_opnd(a ? a : &none, 0);
    g.b = make_opnd(b ? b : &none, 0);
    g.c = make_opnd(c ? c : &none, 0);
    g.out = make_opnd(out ? out : &none, 0);
    g.n_inner = ni;
    g.N = N;
    g.N2 = N2;
    g.kind = kind;
    g.mode = mode;
    g.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) g.eps[i] = eps ? eps[i] : 0.0;
    return g;
}

template <typename T, int OP>
int big_lu_launch(const BigArgs &g, int64_t no, int64_t ni, void *stream)
{
    if (no == 0 || ni == 0) return NFM_OK;
    const size_t lds = (size_t)(g.N * g.N + 2 * g.N) * kWave * sizeof(T);
    if (lds > 64 * 1024) { // opt-in per kernel and per device (lds_opt_in, nfm_common.hpp)
        static std::atomic<uint64_t> have{0};
        const int rc = lds_opt_in(have, reinterpret_cast<const void *>(&big_lu_kernel<T, OP>), 160 * 1024);
        if (rc != NFM_OK) return rc;
    }
    dim3 grid((unsigned)((ni + kWave - 1) / kWave), (unsigned)no, 1);
    hipLaunchKernelGGL((big_lu_kernel<T, OP>), grid, dim3(kWave), lds, static_cast<hipStream_t>(stream), g);
    return launch_status();
}

template <typename T, int OP>
int big_elem_launch(const BigArgs &g, int64_t no, int64_t ni, void *stream)
{
    if (no == 0 || ni == 0) return NFM_OK;
    dim3 grid((unsigned)((ni + 255) / 256), (unsigned)no, 1);
    hipLaunchKernelGGL((big_elem_kernel<T, OP>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), g);
    return launch_status();
}

template <typename T>
int big_sym_solve(int M, int kind, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                  const nfm_operand *out, const double *eps, void *stream)
{
    BigArgs g = big_args(mat, vec, nullptr, out, ni, M, M, kind, 0, eps);
    if (kind == NFM_MAT_DIAG || kind == NFM_MAT_SCAL) return big_elem_launch<T, BIGE_DIVDIAG>(g, no, ni, stream);
    return big_lu_launch<T, BIG_SOLVE>(g, no, ni, stream);
}

template <typename T>
int big_sym_matvec(int M, int kind, int mode, int64_t no, int64_t ni, const nfm_operand *mat,
                   const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(mat, vec, inp, out, ni, M, M, kind, mode, nullptr);
    return big_elem_launch<T, BIGE_MATVEC>(g, no, ni, stream);
}

template <typename T>
int big_sym_invert(int M, int diag_only, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out,
                   void *stream)
{
    BigArgs g = big_args(mat, nullptr, nullptr, out, ni, M, M, NFM_MAT_SYM, diag_only, nullptr);
    return big_lu_launch<T, BIG_INVERT>(g, no, ni, stream);
}

template <typename T>
int big_sym_det(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(mat, nullptr, nullptr, out, ni, M, M, NFM_MAT_SYM, 0, nullptr);
    return big_lu_launch<T, BIG_DET>(g, no, ni, stream);
}

template <typename T>
int big_sym_to_full(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(mat, nullptr, nullptr, out, ni, M, M, NFM_MAT_SYM, 0, nullptr);
    return big_elem_launch<T, BIGE_TOFULL>(g, no, ni, stream);
}

template <typename T>
int big_sym_outer(int M, int64_t no, int64_t ni, const nfm_operand *x, const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(x, nullptr, nullptr, out, ni, M, M, NFM_MAT_SYM, 0, nullptr);
    return big_elem_launch<T, BIGE_OUTER>(g, no, ni, stream);
}

template <typename T>
int big_sym_outer2(int M, int neg, int64_t no, int64_t ni, const nfm_operand *x, const nfm_operand *y,
                   const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(x, y, nullptr, out, ni, M, M, NFM_MAT_SYM, neg, nullptr);
    return big_elem_launch<T, BIGE_OUTER2>(g, no, ni, stream);
}

template <typename T>
int big_sym_matmul(int K, int D, int hess_kind, int64_t no, int64_t ni, const nfm_operand *jac,
                   const nfm_operand *hess, const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(jac, hess, nullptr, out, ni, K, D, hess_kind, 0, nullptr);
    return big_elem_launch<T, BIGE_MATMUL>(g, no, ni, stream);
}

template <typename T>
int big_batch_inv(int N, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(a, nullptr, nullptr, out, ni, N, N, NFM_MAT_FULL, 0, nullptr);
    return big_lu_launch<T, BIG_GINV>(g, no, ni, stream);
}

template <typename T>
int big_batch_det(int N, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(a, nullptr, nullptr, out, ni, N, N, NFM_MAT_FULL, 0, nullptr);
    return big_lu_launch<T, BIG_GDET>(g, no, ni, stream);
}

template <typename T>
int big_batch_matvec(int rows, int cols, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *v,
                     const nfm_operand *out, void *stream)
{
    BigArgs g = big_args(a, v, nullptr, out, ni, rows, cols, NFM_MAT_FULL, 0, nullptr);
    return big_elem_launch<T, BIGE_GMATVEC>(g, no, ni, stream);
}

} // namespace nfm

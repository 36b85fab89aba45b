// nfm_smallmat.hpp -- per-lane small-matrix arithmetic (everything in registers,
// every loop fully unrolled so that no array is ever indexed at run time).
//
// Closed forms for order <= 4 evaluate the reference's formulas in the reference's
// operation order with contraction OFF, and use an explicit fma exactly where the
// reference's ATen kernel is fused (`addcmul_`), so the results are bit-identical to
// the reference's CPU TorchScript path.  File:line citations are relative to
// /root/reference/nitorch_fastmath/.
#pragma once
#include "nfm_common.hpp"

namespace nfm {

template <typename T>
__device__ __forceinline__ T fabs_(T x)
{
    return __builtin_elementwise_abs(x);
}
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fmax_(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ----------------------------------------------------------------------------
// Select policy of the pivoting code.
//   Sel<false>: plain `c ? a : b` (orders <= 8: LLVM keeps these as v_cndmask).
//   Sel<true> : an OPAQUE select -- one v_cndmask_b32 per 32-bit half under the wave mask of
//               the condition, written as inline asm.  For orders >= 9 LLVM turns part of the
//               fully unrolled select network into data-dependent branches; with hundreds of
//               live values the register allocator then places spill code inside those
//               divergent regions and hipcc 7.2 produces wrong, run-to-run varying results
//               (scripts/dbg_stream.hip).  Opaque selects leave no data-dependent control flow
//               (so every spill executes under a full EXEC mask) and, as a bonus, cut the
//               register pressure (16x16 f32 inverse: 512 regs + scratch -> 341 regs).
// ----------------------------------------------------------------------------
__device__ __forceinline__ int asel_(unsigned long long m, int a, int b)
{
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ float asel_(unsigned long long m, float a, float b)
{
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ double asel_(unsigned long long m, double a, double b)
{
    union {
        double d;
        int i[2];
    } ua, ub, ur;
    ua.d = a;
    ub.d = b;
    ur.i[0] = asel_(m, ua.i[0], ub.i[0]);
    ur.i[1] = asel_(m, ua.i[1], ub.i[1]);
    return ur.d;
}

template <bool OPAQUE>
struct Sel;
template <>
struct Sel<false> {
    using M = bool;
    static __device__ __forceinline__ M mask(bool c) { return c; }
    template <typename T>
    static __device__ __forceinline__ T pick(M m, T a, T b)
    {
        return m ? a : b;
    }
};
template <>
struct Sel<true> {
    using M = unsigned long long;
    static __device__ __forceinline__ M mask(bool c) { return __ballot(c); }
    template <typename T>
    static __device__ __forceinline__ T pick(M m, T a, T b)
    {
        return asel_(m, a, b);
    }
};

// ----------------------------------------------------------------------------
// compact symmetric, closed forms (M <= 4)
// ----------------------------------------------------------------------------

// _impl/sym.py:186-190  det = -(u0^2); det.addcmul_(d0, d1)
template <typename T>
__device__ __forceinline__ T sym_det2(const T *d, const T *u)
{
#pragma clang fp contract(off)
    T det = -(u[0] * u[0]);
    return fma_(d[0], d[1], det);
}

// _impl/sym.py:203-209
template <typename T>
__device__ __forceinline__ T sym_det3(const T *d, const T *u)
{
#pragma clang fp contract(off)
    T pd = (d[0] * d[1]) * d[2];
    T pu = (u[0] * u[1]) * u[2];
    T t = (d[0] * (u[2] * u[2]) + d[2] * (u[0] * u[0])) + d[1] * (u[1] * u[1]);
    return (pd + T(2) * pu) - t;
}

// _impl/sym.py:229-248
template <typename T>
__device__ __forceinline__ T sym_det4(const T *d, const T *u)
{
#pragma clang fp contract(off)
    T s0 = u[0] * u[0], s1 = u[1] * u[1], s2 = u[2] * u[2];
    T s3 = u[3] * u[3], s4 = u[4] * u[4], s5 = u[5] * u[5];
    T q05 = u[0] * u[5], q14 = u[1] * u[4], q23 = u[2] * u[3];
    T pd = ((d[0] * d[1]) * d[2]) * d[3];
    T a = (q05 * q05 + q14 * q14) + q23 * q23;
    T b = T(2) * ((((u[0] * u[1]) * u[4]) * u[5] + ((u[0] * u[2]) * u[3]) * u[5]) +
                  ((u[1] * u[2]) * u[3]) * u[4]);
    T c = T(2) * (((((d[0] * u[3]) * u[4]) * u[5] + ((d[1] * u[1]) * u[2]) * u[5]) +
                   ((d[2] * u[0]) * u[2]) * u[4]) +
                  ((d[3] * u[0]) * u[1]) * u[3]);
    T e = ((((((d[0] * d[1]) * s5 + (d[0] * d[2]) * s4) + (d[0] * d[3]) * s3) + (d[1] * d[2]) * s2) +
            (d[1] * d[3]) * s1) +
           (d[2] * d[3]) * s0);
    return (((pd + a) + (-b)) + c) - e;
}

// cofactors of a 4x4 compact matrix (the `inv??` terms and the diagonal brackets of
// _impl/sym.py:251-322).  co[0..3] diagonal cofactors, co[4..9] = inv01 02 03 12 13 23.
template <typename T>
__device__ __forceinline__ void sym_cof4(const T *d, const T *u, T (&co)[10])
{
#pragma clang fp contract(off)
    T s0 = u[0] * u[0], s1 = u[1] * u[1], s2 = u[2] * u[2];
    T s3 = u[3] * u[3], s4 = u[4] * u[4], s5 = u[5] * u[5];
    co[4] = ((((((-d[2]) * d[3]) * u[0] + (d[2] * u[2]) * u[4]) + (d[3] * u[1]) * u[3]) + u[0] * s5) -
             (u[1] * u[4]) * u[5]) -
            (u[2] * u[3]) * u[5];
    co[5] = ((((((-d[1]) * d[3]) * u[1] + (d[1] * u[2]) * u[5]) + (d[3] * u[0]) * u[3]) + u[1] * s4) -
             (u[0] * u[4]) * u[5]) -
            (u[2] * u[3]) * u[4];
    co[6] = ((((((-d[1]) * d[2]) * u[2] + (d[1] * u[1]) * u[5]) + (d[2] * u[0]) * u[4]) + u[2] * s3) -
             (u[0] * u[3]) * u[5]) -
            (u[1] * u[3]) * u[4];
    co[7] = ((((((-d[0]) * d[3]) * u[3] + (d[0] * u[4]) * u[5]) + (d[3] * u[0]) * u[1]) + u[3] * s2) -
             (u[0] * u[2]) * u[5]) -
            (u[1] * u[2]) * u[4];
    co[8] = ((((((-d[0]) * d[2]) * u[4] + (d[0] * u[3]) * u[5]) + (d[2] * u[0]) * u[2]) + u[4] * s1) -
             (u[0] * u[1]) * u[5]) -
            (u[1] * u[2]) * u[3];
    co[9] = ((((((-d[0]) * d[1]) * u[5] + (d[0] * u[4]) * u[3]) + (d[1] * u[1]) * u[2]) + u[5] * s0) -
             (u[0] * u[1]) * u[4]) -
            (u[0] * u[2]) * u[3];
    co[0] = ((((d[1] * d[2]) * d[3] - d[1] * s5) - d[2] * s4) - d[3] * s3) + ((T(2) * u[3]) * u[4]) * u[5];
    co[1] = ((((d[0] * d[2]) * d[3] - d[0] * s5) - d[2] * s2) - d[3] * s1) + ((T(2) * u[1]) * u[2]) * u[5];
    co[2] = ((((d[0] * d[1]) * d[3] - d[0] * s4) - d[1] * s2) - d[3] * s0) + ((T(2) * u[0]) * u[2]) * u[4];
    co[3] = ((((d[0] * d[1]) * d[2] - d[0] * s3) - d[1] * s1) - d[2] * s0) + ((T(2) * u[0]) * u[1]) * u[3];
}

// cofactors of a 3x3 compact matrix, _impl/sym.py:216-224; co[0..2] diag, co[3..5] = 01 02 12
template <typename T>
__device__ __forceinline__ void sym_cof3(const T *d, const T *u, T (&co)[6])
{
#pragma clang fp contract(off)
    co[0] = d[1] * d[2] - u[2] * u[2];
    co[1] = d[0] * d[2] - u[1] * u[1];
    co[2] = d[0] * d[1] - u[0] * u[0];
    co[3] = u[1] * u[2] - d[2] * u[0];
    co[4] = u[0] * u[2] - d[1] * u[1];
    co[5] = u[0] * u[1] - d[0] * u[2];
}

// x = A^-1 v, compact A of order M <= 4, _impl/sym.py:193-200, 212-226, 251-324, 384-391.
// Two halves, so that one matrix solved against many vectors derives its cofactors once
// (sym_solve_bcast_kernel, nfm_sym.hip): the part that depends on the matrix only ...
template <int M>
struct SymCofLen {
    static constexpr int value = M == 4 ? 10 : (M == 3 ? 6 : (M == 2 ? 3 : 1));
};
template <typename T, int M>
__device__ __forceinline__ void sym_solve_prepare(const T (&m)[sym_k(M)], T (&co)[SymCofLen<M>::value], T &det)
{
#pragma clang fp contract(off)
    static_assert(M >= 1 && M <= 4, "closed forms exist for M <= 4");
    if constexpr (M == 1) {
        co[0] = T(1);
        det = m[0];
    } else if constexpr (M == 2) {
        det = sym_det2(&m[0], &m[2]);
        co[0] = m[0];
        co[1] = m[1];
        co[2] = m[2];
    } else if constexpr (M == 3) {
        det = sym_det3(&m[0], &m[3]);
        sym_cof3(&m[0], &m[3], co);
    } else {
        det = sym_det4(&m[0], &m[4]);
        sym_cof4(&m[0], &m[4], co);
    }
}

// ... and the part per right-hand side
template <typename T, int M>
__device__ __forceinline__ void sym_solve_apply(const T (&co)[SymCofLen<M>::value], T det, const T (&v)[M], T (&r)[M])
{
#pragma clang fp contract(off)
    if constexpr (M == 1) {
        r[0] = v[0] / det;
    } else if constexpr (M == 2) {
        r[0] = (co[1] * v[0] - co[2] * v[1]) / det;
        r[1] = (co[0] * v[1] - co[2] * v[0]) / det;
    } else if constexpr (M == 3) {
        r[0] = ((co[0] * v[0] + co[3] * v[1]) + co[4] * v[2]) / det;
        r[1] = ((co[3] * v[0] + co[1] * v[1]) + co[5] * v[2]) / det;
        r[2] = ((co[4] * v[0] + co[5] * v[1]) + co[2] * v[2]) / det;
    } else {
        // res[i] = cof_ii * v[i]; res[i] += inv.. * v[..] in the reference's order
        r[0] = (((co[0] * v[0] + co[4] * v[1]) + co[5] * v[2]) + co[6] * v[3]) / det;
        r[1] = (((co[1] * v[1] + co[4] * v[0]) + co[7] * v[2]) + co[8] * v[3]) / det;
        r[2] = (((co[2] * v[2] + co[5] * v[0]) + co[7] * v[1]) + co[9] * v[3]) / det;
        r[3] = (((co[3] * v[3] + co[6] * v[0]) + co[8] * v[1]) + co[9] * v[2]) / det;
    }
}

template <typename T, int M>
__device__ __forceinline__ void sym_solve_closed(const T (&m)[sym_k(M)], const T (&v)[M], T (&r)[M])
{
    T co[SymCofLen<M>::value], det;
    sym_solve_prepare<T, M>(m, co, det);
    sym_solve_apply<T, M>(co, det, v, r);
}

// compact inverse, order M <= 4: what `sym_invert` (_impl/sym.py:455-493) obtains by
// solving against each basis vector = cofactor / det, in compact order.
template <typename T, int M>
__device__ __forceinline__ void sym_invert_closed(const T (&m)[sym_k(M)], T (&r)[sym_k(M)])
{
#pragma clang fp contract(off)
    if constexpr (M == 1) {
        r[0] = T(1) / m[0];
    } else if constexpr (M == 2) {
        T det = sym_det2(&m[0], &m[2]);
        r[0] = m[1] / det;
        r[1] = m[0] / det;
        r[2] = (-m[2]) / det;
    } else if constexpr (M == 3) {
        T det = sym_det3(&m[0], &m[3]);
        T co[6];
        sym_cof3(&m[0], &m[3], co);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = co[i] / det;
    } else {
        T det = sym_det4(&m[0], &m[4]);
        T co[10];
        sym_cof4(&m[0], &m[4], co);
#pragma unroll
        for (int i = 0; i < 10; ++i) r[i] = co[i] / det;
    }
}

// y = A v, compact A: _impl/sym.py:87-131.  mm = diag * v, then one fused
// `addcmul_` per off-diagonal entry, in the reference's order for each M.
template <typename T, int M>
__device__ __forceinline__ void sym_matvec_compact(const T (&m)[sym_k(M)], const T (&v)[M], T (&r)[M])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < M; ++i) r[i] = m[i] * v[i];
    if constexpr (M >= 2 && M <= 4) {
        // _sym_matvec2/3/4: row by row, each row's chain left to right
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j != i) r[i] = fma_(m[sym_idx(M, i, j)], v[j], r[i]);
    } else if constexpr (M > 4) {
        // _sym_matvecn: walk the upper triangle once, updating rows i and j
        int c = M;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i + 1; j < M; ++j) {
                r[i] = fma_(m[c], v[j], r[i]);
                r[j] = fma_(m[c], v[i], r[j]);
                ++c;
            }
    }
}

// ----------------------------------------------------------------------------
// general N x N in registers
// ----------------------------------------------------------------------------

template <typename T, int M>
__device__ __forceinline__ void sym_expand(const T (&m)[sym_k(M)], T (&a)[M][M])
{
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) a[i][j] = m[sym_idx(M, i, j)];
}

// NB: registers cannot be indexed dynamically, so a run-time row swap k <-> p is
// done with selects: every candidate row i > k exchanges with row k under (p == i).

// Gaussian elimination with partial pivoting + back substitution on [A | B],
// NR right-hand sides; B is overwritten by A^-1 B.  Mirrors LAPACK getrf/getrs
// (what torch.linalg.solve runs on the reference's M > 4 branch,
// _impl/sym.py:392-396): pivot = first largest |a_ik|, multipliers formed with
// the reciprocal pivot, back substitution divides by the diagonal.
template <typename T, int N, int NR>
__device__ __forceinline__ void ge_solve(T (&a)[N][N], T (&b)[N][NR])
{
    using S = Sel<(N > 8)>;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if constexpr (N > 1) {
            int p = k;
            T best = fabs_(a[k][k]);
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const T x = fabs_(a[i][k]);
                const typename S::M g = S::mask(x > best);
                best = S::pick(g, x, best);
                p = S::pick(g, i, p);
            }
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const typename S::M s = S::mask(p == i);
#pragma unroll
                for (int j = k; j < N; ++j) {
                    const T t = a[k][j];
                    a[k][j] = S::pick(s, a[i][j], t);
                    a[i][j] = S::pick(s, t, a[i][j]);
                }
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const T t = b[k][r];
                    b[k][r] = S::pick(s, b[i][r], t);
                    b[i][r] = S::pick(s, t, b[i][r]);
                }
            }
        }
        const T rp = T(1) / a[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const T l = a[i][k] * rp;
#pragma unroll
            for (int j = k + 1; j < N; ++j) a[i][j] -= l * a[k][j];
#pragma unroll
            for (int r = 0; r < NR; ++r) b[i][r] -= l * b[k][r];
        }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            T s = b[i][r];
#pragma unroll
            for (int j = i + 1; j < N; ++j) s -= a[i][j] * b[j][r];
            b[i][r] = s / a[i][i];
        }
    }
}

// ----------------------------------------------------------------------------
// symmetric positive definite matrices in COMPACT storage: A = U^T D U, U unit upper triangular
// ----------------------------------------------------------------------------
// The sym_* functions are fed Hessians: symmetric positive definite in practice.  For those the factorisation
// without pivoting is backward stable (no growth: |u_kj|^2 d_k <= a_jj), works on the N (N + 1) / 2 stored
// values in place and costs N^3 / 6 fma -- against N^2 registers, 2 N^3 / 3 fma and as many selects again for the
// run-time row exchanges of the pivoted LU the reference takes (`torch.linalg.solve`, `_impl/sym.py:392-396`).
// `ldl_factor` reports whether every pivot was positive (Sylvester: exactly the positive definite matrices); the
// callers vote over the wavefront and redo the whole wavefront with the pivoted elimination when one matrix was not.
// After the call: m[i] = 1 / d_i, m[sym_idx(i, j)] = u_ij (i < j).
template <typename T, int N>
__device__ __forceinline__ bool ldl_factor(T (&m)[sym_k(N)], T &det)
{
    // (loops over literal bounds with a predicate inside: a bound that depends on an outer induction variable is
    // not a constant until the outer loop is unrolled, the inner loop then misses the unroller's first pass, and
    // the array stays in scratch memory -- seen at order 10, and with ldl_inverse at every order)
    bool ok = true;
    det = T(1);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const T d = m[k];
        ok = ok && d > T(0); // (a NaN fails)
        det *= d;
        const T r = T(1) / d;
        m[k] = r;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i > k) {
                const T u = m[sym_idx(N, k, i)] * r; // u_ki; the row k itself stays unscaled until its last use
                m[i] = fma_(-u, m[sym_idx(N, k, i)], m[i]);
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j > i) m[sym_idx(N, i, j)] = fma_(-u, m[sym_idx(N, k, j)], m[sym_idx(N, i, j)]);
            }
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i > k) m[sym_idx(N, k, i)] *= r;
    }
    return ok;
}

// x = A^-1 v from the factors: U^T y = v, z = D^-1 y, U x = z
template <typename T, int N>
__device__ __forceinline__ void ldl_solve(const T (&m)[sym_k(N)], const T (&v)[N], T (&x)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = v[i];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j > k) x[j] = fma_(-m[sym_idx(N, k, j)], x[k], x[j]);
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] *= m[k];
#pragma unroll
    for (int k = N - 1; k >= 0; --k)
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j > k) x[k] = fma_(-m[sym_idx(N, k, j)], x[j], x[k]);
}

// U -> W = U^-1 in place (unit upper triangular, compact storage): W_ij = -(u_ij + sum_{i<k<j} W_ik u_kj); row i
// uses the rows below it as they were and its own earlier columns as they have become
template <typename T, int N>
__device__ __forceinline__ void ldl_unit_inverse(T (&m)[sym_k(N)])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j > i) {
                T s = m[sym_idx(N, i, j)];
#pragma unroll
                for (int k = 0; k < N; ++k)
                    if (k > i && k < j) s = fma_(m[sym_idx(N, i, k)], m[sym_idx(N, k, j)], s);
                m[sym_idx(N, i, j)] = -s;
            }
        // (rows are independent of each other's results: left alone, the scheduler interleaves them all and holds
        // U and W at the same time -- 295 registers instead of 175 at order 16; one row at a time)
        if (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
}

// the factors of ldl_factor -> A^-1 = W D^-1 W^T in compact storage, in place: N^3 / 3 fma more
//   (A^-1)_ij = z_j + sum_{k>j} z_k W_jk,  (A^-1)_ii = 1/d_i + sum_{k>i} z_k W_ik,  z_k = W_ik / d_k
template <typename T, int N>
__device__ __forceinline__ void ldl_inverse(T (&m)[sym_k(N)])
{
    ldl_unit_inverse<T, N>(m);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T z[N];
        T dsum = m[i];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            z[k] = T(0);
            if (k > i) {
                z[k] = m[sym_idx(N, i, k)] * m[k];
                dsum = fma_(z[k], m[sym_idx(N, i, k)], dsum);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j > i) {
                T s = z[j];
#pragma unroll
                for (int k = 0; k < N; ++k)
                    if (k > j) s = fma_(z[k], m[sym_idx(N, j, k)], s);
                m[sym_idx(N, i, j)] = s;
            }
        m[i] = dsum;
        if (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
}

// diagonal of A^-1 only
template <typename T, int N>
__device__ __forceinline__ void ldl_inverse_diag(T (&m)[sym_k(N)], T (&dg)[N])
{
    ldl_unit_inverse<T, N>(m);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T s = m[i];
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (k > i) s = fma_(m[sym_idx(N, i, k)] * m[k], m[sym_idx(N, i, k)], s);
        dg[i] = s;
    }
}

// ----------------------------------------------------------------------------
// general matrices WITHOUT row exchanges, when the diagonal allows it
// ----------------------------------------------------------------------------
// Partial pivoting exchanges rows to keep every multiplier <= 1; THRESHOLD pivoting accepts a pivot that is within
// a factor of the column maximum (multipliers <= 1 / u: the rule of the sparse direct solvers, u = 0.1 by default
// there, which prefer the diagonal for the same reason as here -- an exchange is expensive) and is as stable in
// practice.  The diagonal is accepted when |a_kk| >= max_{i>k} |a_ik| / 8 at every step; `ok` reports whether it
// was -- always, for the diagonally dominant and the regularised matrices this library is fed; N(0,1) + 6 I at
// order 16: all but 3 matrices in 10 000 (with 1 / 2 it was 29, and every wavefront that holds one of them pays the
// attempt AND the pivoted elimination) -- and the callers vote over the wavefront and redo it with the pivoted
// elimination when one matrix said no.  No exchange: no selects (they were half of the instructions of the pivoted
// kernels at order 16), and the inverse runs in place on N^2 registers.
template <typename T>
__device__ __forceinline__ bool diag_pivot_ok(T akk, T cmax)
{
    return fabs_(akk) >= T(0.125) * cmax && fabs_(akk) > T(0); // (a NaN fails)
}

// determinant by elimination without exchanges (the product of the pivots)
template <typename T, int N>
__device__ __forceinline__ T lu_det_nopivot(T (&a)[N][N], bool &ok)
{
    T det = T(1);
    ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T cmax = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i > k) cmax = fmax_(cmax, fabs_(a[i][k]));
        ok = ok && diag_pivot_ok(a[k][k], cmax);
        det *= a[k][k];
        const T r = T(1) / a[k][k];
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i > k) {
                const T l = a[i][k] * r;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j > k) a[i][j] = fma_(-l, a[k][j], a[i][j]);
            }
    }
    return det;
}

// in-place Gauss-Jordan inverse without exchanges: N^3 fma on N^2 registers
template <typename T, int N>
__device__ __forceinline__ void gj_inverse_nopivot(T (&a)[N][N], bool &ok)
{
    ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T cmax = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i > k) cmax = fmax_(cmax, fabs_(a[i][k]));
        ok = ok && diag_pivot_ok(a[k][k], cmax);
        const T p = T(1) / a[k][k];
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j != k) a[k][j] *= p;
        a[k][k] = p;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i != k) {
                const T f = a[i][k];
                a[i][k] = -f * p;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j != k) a[i][j] = fma_(-f, a[k][j], a[i][j]);
            }
        if (N > 8) __builtin_amdgcn_sched_barrier(0); // one step at a time (register pressure, see ldl_unit_inverse)
    }
}

// In-place Gauss-Jordan inverse with partial pivoting (N^2 registers, no second
// matrix): row swaps during elimination, the matching column swaps undone at the
// end.  Singular input -> inf/NaN, like the reference (no error).
template <typename T, int N>
__device__ __forceinline__ void gj_inverse(T (&a)[N][N])
{
    using S = Sel<(N > 8)>;
    int piv[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int p = k;
        if constexpr (N > 1) {
            T best = fabs_(a[k][k]);
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const T x = fabs_(a[i][k]);
                const typename S::M g = S::mask(x > best);
                best = S::pick(g, x, best);
                p = S::pick(g, i, p);
            }
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const typename S::M s = S::mask(p == i);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const T t = a[k][j];
                    a[k][j] = S::pick(s, a[i][j], t);
                    a[i][j] = S::pick(s, t, a[i][j]);
                }
            }
        }
        piv[k] = p;
        const T rp = T(1) / a[k][k];
        a[k][k] = T(1);
#pragma unroll
        for (int j = 0; j < N; ++j) a[k][j] *= rp;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i != k) {
                const T f = a[i][k];
                a[i][k] = T(0);
#pragma unroll
                for (int j = 0; j < N; ++j) a[i][j] -= f * a[k][j];
            }
        }
        if constexpr (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = N - 2; k >= 0; --k) {
#pragma unroll
        for (int c = k + 1; c < N; ++c) {
            const typename S::M s = S::mask(piv[k] == c);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const T t = a[i][k];
                a[i][k] = S::pick(s, a[i][c], t);
                a[i][c] = S::pick(s, t, a[i][c]);
            }
        }
        if constexpr (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
}

// LU factorisation in place with partial pivoting, keeping the row permutation as the
// original index of every row (rowid), and the solve of L U x = P e_c for one unit vector.
// Together they build an inverse column by column with only N^2 + 3N live values -- the
// in-place Gauss-Jordan needs noticeably more temporaries and spills beyond 13x13.
template <typename T, int N>
__device__ __forceinline__ void lu_factor_rowid(T (&a)[N][N], int (&rowid)[N])
{
    using S = Sel<(N > 8)>;
#pragma unroll
    for (int i = 0; i < N; ++i) rowid[i] = i;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if constexpr (N > 1) {
            int p = k;
            T best = fabs_(a[k][k]);
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const T x = fabs_(a[i][k]);
                const typename S::M g = S::mask(x > best);
                best = S::pick(g, x, best);
                p = S::pick(g, i, p);
            }
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const typename S::M s = S::mask(p == i);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const T t = a[k][j];
                    a[k][j] = S::pick(s, a[i][j], t);
                    a[i][j] = S::pick(s, t, a[i][j]);
                }
                const int ti = rowid[k];
                rowid[k] = S::pick(s, rowid[i], ti);
                rowid[i] = S::pick(s, ti, rowid[i]);
            }
        }
        const T rp = T(1) / a[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const T l = a[i][k] * rp;
            a[i][k] = l;
#pragma unroll
            for (int j = k + 1; j < N; ++j) a[i][j] -= l * a[k][j];
        }
        // keep the scheduler from interleaving elimination steps: it lengthens live ranges
        // until a 16x16 matrix no longer fits the register file
        if constexpr (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
}

// x = A^-1 e_c from the factors above (column c of the inverse)
template <typename T, int N>
__device__ __forceinline__ void lu_solve_unit(const T (&lu)[N][N], const int (&rowid)[N], int c, T (&x)[N])
{
    using S = Sel<(N > 8)>;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = S::pick(S::mask(rowid[i] == c), T(1), T(0));
#pragma unroll
    for (int i = 1; i < N; ++i) {
        T s = x[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= lu[i][j] * x[j];
        x[i] = s;
        if constexpr (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        T s = x[i];
#pragma unroll
        for (int j = i + 1; j < N; ++j) s -= lu[i][j] * x[j];
        x[i] = s / lu[i][i];
        if constexpr (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
}

// determinant by LU with partial pivoting (torch.det on the reference's fallbacks,
// _impl/batched.py:53-54, _impl/sym.py:447-450)
template <typename T, int N>
__device__ __forceinline__ T lu_det(T (&a)[N][N])
{
    using S = Sel<(N > 8)>;
    T det = T(1);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if constexpr (N > 1) {
            int p = k;
            T best = fabs_(a[k][k]);
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const T x = fabs_(a[i][k]);
                const typename S::M g = S::mask(x > best);
                best = S::pick(g, x, best);
                p = S::pick(g, i, p);
            }
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const typename S::M s = S::mask(p == i);
#pragma unroll
                for (int j = k; j < N; ++j) {
                    const T t = a[k][j];
                    a[k][j] = S::pick(s, a[i][j], t);
                    a[i][j] = S::pick(s, t, a[i][j]);
                }
            }
            det = S::pick(S::mask(p != k), -det, det);
        }
        const T pivot = a[k][k];
        // a zero pivot is the largest of a zero column: nothing to eliminate, and the determinant is 0 whatever
        // follows (LAPACK's getrf leaves the column alone too) -- not 0 * inf = NaN
        const T rp = pivot == T(0) ? T(0) : T(1) / pivot;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const T l = a[i][k] * rp;
#pragma unroll
            for (int j = k + 1; j < N; ++j) a[i][j] -= l * a[k][j];
        }
        if constexpr (N > 8) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) det *= a[k][k];
    return det;
}

// ----------------------------------------------------------------------------
// general small matrices, closed forms of _impl/batched.py (N <= 3)
// ----------------------------------------------------------------------------
template <typename T, int N>
__device__ __forceinline__ T det_closed(const T (&A)[N * N])
{
#pragma clang fp contract(off)
    if constexpr (N == 1) {
        return A[0];
    } else if constexpr (N == 2) { // det2 :21-24
        return A[0] * A[3] - A[1] * A[2];
    } else { // det3 :27-32
        return (A[0] * (A[4] * A[8] - A[5] * A[7]) + A[1] * (A[5] * A[6] - A[3] * A[8])) +
               A[2] * (A[3] * A[7] - A[4] * A[6]);
    }
}

// inv2 / inv3, _impl/batched.py:66-98; `perturb` adds (max|A| - min|A|) * 1e-12 to det
template <typename T, int N>
__device__ __forceinline__ void inv_closed(const T (&A)[N * N], T (&F)[N * N], bool perturb)
{
#pragma clang fp contract(off)
    if constexpr (N == 1) {
        F[0] = T(1) / A[0]; // a.reciprocal() :128
    } else {
        T dt = det_closed<T, N>(A);
        if (perturb) {
            T amax = fabs_(A[0]), amin = fabs_(A[0]);
#pragma unroll
            for (int i = 1; i < N * N; ++i) {
                const T x = fabs_(A[i]);
                amax = x > amax ? x : amax;
                amin = x < amin ? x : amin;
            }
            dt = dt + (amax - amin) * T(1E-12);
        }
        if constexpr (N == 2) {
            F[0] = A[3] / dt;
            F[1] = (-A[1]) / dt;
            F[2] = (-A[2]) / dt;
            F[3] = A[0] / dt;
        } else {
            F[0] = (A[4] * A[8] - A[5] * A[7]) / dt;
            F[1] = (A[2] * A[7] - A[1] * A[8]) / dt;
            F[2] = (A[1] * A[5] - A[2] * A[4]) / dt;
            F[3] = (A[5] * A[6] - A[3] * A[8]) / dt;
            F[4] = (A[0] * A[8] - A[2] * A[6]) / dt;
            F[5] = (A[3] * A[2] - A[5] * A[0]) / dt;
            F[6] = (A[7] * A[3] - A[6] * A[4]) / dt;
            F[7] = (A[6] * A[1] - A[7] * A[0]) / dt;
            F[8] = (A[0] * A[4] - A[1] * A[3]) / dt;
        }
    }
}

} // namespace nfm

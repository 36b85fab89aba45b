// nfm_qr_core.hpp -- per-lane Givens / Householder / QR-algorithm arithmetic
// (reference `_impl/qr.py`, real dtypes).
//
// Every routine is a template on <T, NT>: NT > 0 is a compile-time order -- all loops
// unroll, every array index is a constant and the matrices live in VGPRs; NT == 0 is a
// run-time order n <= 16 with the same source -- the arrays are then indexed at run time
// and the compiler places them in per-lane scratch memory (orders 9..16: correct, not tuned).
//
// Contraction is off and operations follow the CPU restatement's order (which follows the
// reference's), so results agree with it to the last bit wherever sqrt/div are correctly
// rounded; the parity tests still allow the north-star tolerance.
//
// Deliberate deviations from the reference: Q7 the symmetric mat-vec works
// for every n (upstream raises for batched n > 5); Q8 `rq_hessenberg` applies its column
// rotations to rows 0..k+1 (a true R Q) unless the tridiagonal `sym` shortcut is requested;
// Q9 convergence of the QR iterations is judged PER MATRIX with the reference's criterion
// (upstream sums it over the whole batch, three host syncs per iteration).
#pragma once
#include "nfm_common.hpp"
#include "nfm_smallmat.hpp"

namespace nfm {
namespace qr {

template <int NT>
struct Dim {
    static constexpr int MAX = NT > 0 ? NT : NFM_MAX_DIM;
};

__device__ __forceinline__ float sqrt_(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return __builtin_sqrt(x); }
template <typename T>
__device__ __forceinline__ bool finite_(T x)
{
    return fabs_(x) < __builtin_huge_val() && x == x;
}

// ---------------------------------------------------------------------------------------
// Arithmetic policy of the QR SWEEPS of eig_sym (not of the public givens / rq_hessenberg /
// householder entry points, which keep the reference's operation order and IEEE div / sqrt).
//
// IEEE division and square root are software sequences on gfx950 (v_div_scale x2, v_rcp,
// 4 fma, v_div_fmas, v_div_fixup; v_sqrt + fix-up), a third of the instructions of a sweep.
// The fast sweeps use the hardware approximations instead, refined where the value
// enters the result: a rotation is (c, s) = (x, -y) * rsqrt(x^2 + y^2) with one Newton step
// on v_rsq_f32 (relative error <= ~1 ulp, i.e. the same as the reference's sqrt followed by
// two divisions), products and sums contract to fma (each fma rounds once where the
// reference rounds twice).  The Wilkinson shift and the convergence ratio only steer the
// iteration -- ANY shift gives a similarity transform -- so they use v_sqrt_f32 / v_rcp_f32
// unrefined.  Arguments outside [2^-100, 2^100] (zeros, denormal squares, inf, NaN) take the
// IEEE path, so the special values behave exactly as before.  float64 does the same with two
// Newton steps on v_rsq_f64 (~26 bits): 11 instructions for a rotation against ~50 for the IEEE
// square root and two divisions.
//
// What the fast sweeps change is rounding, not accuracy (measured against numpy.linalg.eigvalsh
// in float64: within 2x the error of the reference-order arithmetic at every order,
// profiles/r02/accuracy_eig.md) -- but the ORDER in which the eigenvalues deflate and the SIGNS
// of the eigenvectors are decided by bits that the rounding moves (for 8x8 one matrix in eight
// deflates in another order), and two float32 runs that are each 7e-7 accurate can differ by
// 1.4e-6.  So the reference-order arithmetic stays available (FM = false: bit-identical to the
// CPU restatement, same order, same signs) and the caller picks: `eig_sym(..., arithmetic=)`.
// Error model and its tests: tests/test_gpu_qr.py.
template <typename T>
struct FastSweeps {
    static constexpr bool on = true; // both dtypes; FAST (a template argument of the callers) selects it
};

__device__ __forceinline__ float rsq_nr(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    const float h = (0.5f * x) * r;
    return __builtin_fmaf(r, __builtin_fmaf(-h, r, 0.5f), r); // r (1.5 - 0.5 x r^2)
}
__device__ __forceinline__ double rsq_nr(double x)
{
    double r = __builtin_amdgcn_rsq(x); // ~26 bits: two Newton steps
    const double hx = 0.5 * x;
    r = __builtin_fma(r, __builtin_fma(-(hx * r), r, 0.5), r);
    return __builtin_fma(r, __builtin_fma(-(hx * r), r, 0.5), r);
}
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float copysign_t(float a, float b) { return __builtin_copysignf(a, b); }
__device__ __forceinline__ double copysign_t(double a, double b) { return __builtin_copysign(a, b); }
__device__ __forceinline__ float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double hw_sqrt(double x) { return x * rsq_nr(x); }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double hw_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r); // one step: ~50 bits, the value only steers
}
// safe range of the squared magnitudes for the hardware approximations (no denormals, no overflow)
template <typename T>
struct FastRange;
template <>
struct FastRange<float> {
    static constexpr float lo = 0x1p-100f, hi = 0x1p100f;
};
template <>
struct FastRange<double> {
    static constexpr double lo = 0x1p-900, hi = 0x1p900;
};

// _givens_jit :326-334
template <typename T>
__device__ __forceinline__ void givens1(T x, T y, T &c, T &s);

// sweep form of the same rotation (policy above).  The fast values are computed unconditionally;
// only if SOME lane of the wavefront is outside the safe range (a uniform vote: no exec-mask
// juggling on the common path) the IEEE form runs as well and those lanes take its result.
template <typename T>
__device__ __forceinline__ void givens_fast1(T x, T y, T &c, T &s)
{
    const T r2 = fma_t(x, x, y * y);
    // lanes that need the IEEE form: squared norm outside the safe range, or an axis-aligned pair
    // (x y == 0) -- that one is an exact rotation in the reference (x / |x| = +-1) and must stay
    // exact, so that diagonal / already deflated input comes back bit for bit
    // |x y| <= r2 / 2: a product above the floor bounds r2 from below as well
    const bool ok = fabs_(x * y) > FastRange<T>::lo && r2 < FastRange<T>::hi;
    const T inv = rsq_nr(r2);
    c = x * inv;
    s = -(y * inv);
    if (__builtin_expect(__any(!ok), 0)) {
        T c2, s2;
        givens1<T>(x, y, c2, s2);
        c = ok ? c : c2;
        s = ok ? s : s2;
    }
}

template <typename T>
__device__ __forceinline__ void rot_fast1(T &a0, T &a1, T c, T s)
{
    const T t = s * a0;
    a0 = fma_t(a0, c, -(s * a1));
    a1 = fma_t(a1, c, t);
}

template <typename T>
__device__ __forceinline__ void givens1(T x, T y, T &c, T &s)
{
#pragma clang fp contract(off)
    const T nrm = sqrt_(x * x + y * y);
    const bool z = nrm == T(0);
    c = z ? T(1) : x / nrm;
    s = z ? T(0) : -(y / nrm);
}

// tmp = s*a0; a0 = a0*c - s*a1; a1 = a1*c + tmp   (_givens_apply_* :370-402)
template <typename T>
__device__ __forceinline__ void rot1(T &a0, T &a1, T c, T s)
{
#pragma clang fp contract(off)
    const T tmp = s * a0;
    a0 = a0 * c - s * a1;
    a1 = a1 * c + tmp;
}

// householder_ :55-69 on x[0..m), reflecting onto component `basis` (compile-time or not;
// the element is picked by a select so that registers are never indexed dynamically)
template <typename T, int NT, bool FAST = false>
__device__ __forceinline__ T householder1(T (&x)[Dim<NT>::MAX], int m, int basis)
{
#pragma clang fp contract(off)
    T xb = T(0);
#pragma unroll
    for (int i = 0; i < Dim<NT>::MAX; ++i)
        if (i < m) xb = (i == basis) ? x[i] : xb;
    T rho = (xb > T(0)) ? T(1) : ((xb < T(0)) ? T(-1) : T(0));
    rho = -rho;
    rho = (rho == T(0)) ? T(1) : rho;
    T ss = T(0);
#pragma unroll
    for (int i = 0; i < Dim<NT>::MAX; ++i)
        if (i < m) ss += x[i] * x[i];
    if constexpr (FAST) {
        // eig_sym's fast arithmetic (policy above): |x| = ss rsqrt(ss); the reflected component is
        // x_b - rho = sgn(x_b) (|x_b| + |x|), so the squared norm of the un-normalised reflector is
        // 2 |x| (|x| + |x_b|) -- two operations instead of a second sum of squares -- and
        // u = x rsqrt(that).  Sums of squares outside the safe range (zero vectors, denormals,
        // overflow, NaN) take the IEEE form on a uniform vote, as in givens_fast1.
        T ssf = T(0);
#pragma unroll
        for (int i = 0; i < Dim<NT>::MAX; ++i)
            if (i < m) ssf = fma_t(x[i], x[i], ssf);
        const T nrm = ssf * rsq_nr(ssf);
        const T ss2 = (nrm + fabs_(xb)) * (nrm + nrm);
        const bool ok = ssf > FastRange<T>::lo && ssf < FastRange<T>::hi && ss2 < FastRange<T>::hi;
        const T rhof = rho * nrm;
        const T inv = rsq_nr(ss2);
        if (__builtin_expect(__any(!ok), 0)) {
            T y[Dim<NT>::MAX];
#pragma unroll
            for (int i = 0; i < Dim<NT>::MAX; ++i) y[i] = x[i];
            const T rho2 = householder1<T, NT, false>(y, m, basis);
#pragma unroll
            for (int i = 0; i < Dim<NT>::MAX; ++i)
                if (i < m) x[i] = ok ? ((i == basis) ? x[i] - rhof : x[i]) * inv : y[i];
            return ok ? rhof : rho2;
        }
#pragma unroll
        for (int i = 0; i < Dim<NT>::MAX; ++i)
            if (i < m) x[i] = ((i == basis) ? x[i] - rhof : x[i]) * inv;
        return rhof;
    }
    rho *= sqrt_(ss);
#pragma unroll
    for (int i = 0; i < Dim<NT>::MAX; ++i)
        if (i < m) x[i] = (i == basis) ? x[i] - rho : x[i];
    ss = T(0);
#pragma unroll
    for (int i = 0; i < Dim<NT>::MAX; ++i)
        if (i < m) ss += x[i] * x[i];
    const T nrm = sqrt_(ss);
#pragma unroll
    for (int i = 0; i < Dim<NT>::MAX; ++i)
        if (i < m) {
            const T v = x[i] / nrm;
            x[i] = finite_(v) ? v : T(0);
        }
    return rho;
}

// hessenberg_ :117-141.  up[k][r]: reflector k (length n-1-k), kept when WITH_U.
template <typename T, int NT, bool WITH_U>
__device__ __forceinline__ void hessenberg1(T (&a)[Dim<NT>::MAX][Dim<NT>::MAX], int n,
                                            T (&up)[Dim<NT>::MAX][Dim<NT>::MAX])
{
#pragma clang fp contract(off)
    constexpr int MX = Dim<NT>::MAX;
#pragma unroll
    for (int k = 0; k < MX - 2; ++k) {
        if (k < n - 2) {
            const int m = n - k - 1;
            T u[MX];
#pragma unroll
            for (int r = 0; r < MX; ++r)
                if (r < m) u[r] = a[k + 1 + r][k];
            const T alpha = householder1<T, NT>(u, m, 0);
            if (WITH_U) {
#pragma unroll
                for (int r = 0; r < MX; ++r)
                    if (r < m) up[k][r] = u[r];
            }
#pragma unroll
            for (int c = k + 1; c < MX; ++c)
                if (c < n) {
                    T d = T(0);
#pragma unroll
                    for (int r = 0; r < MX; ++r)
                        if (r < m) d += u[r] * a[k + 1 + r][c];
#pragma unroll
                    for (int r = 0; r < MX; ++r)
                        if (r < m) a[k + 1 + r][c] -= T(2) * (u[r] * d);
                }
#pragma unroll
            for (int r = 0; r < MX; ++r)
                if (r < n) {
                    T d = T(0);
#pragma unroll
                    for (int c = 0; c < MX; ++c)
                        if (c < m) d += a[r][k + 1 + c] * u[c];
#pragma unroll
                    for (int c = 0; c < MX; ++c)
                        if (c < m) a[r][k + 1 + c] -= T(2) * (d * u[c]);
                }
            a[k + 1][k] = alpha;
#pragma unroll
            for (int r = k + 2; r < MX; ++r)
                if (r < n) a[r][k] = T(0);
        }
    }
}

// hessenberg_sym_lower_ :296-323 on a matrix whose LOWER triangle holds the data (the
// caller mirrors the requested triangle on load, which is what the reference's transposed
// view does for upper=True).  Output: symmetric tridiagonal, both halves filled.
template <typename T, int NT, bool WITH_U, bool FAST = false>
__device__ __forceinline__ void hessenberg_sym1(T (&a)[Dim<NT>::MAX][Dim<NT>::MAX], int n,
                                                T (&up)[Dim<NT>::MAX][Dim<NT>::MAX])
{
#pragma clang fp contract(off)
    constexpr int MX = Dim<NT>::MAX;
#pragma unroll
    for (int k = 0; k < MX - 2; ++k) {
        if (k < n - 2) {
            const int m = n - k - 1, o = k + 1;
            T u[MX], v[MX];
#pragma unroll
            for (int r = 0; r < MX; ++r)
                if (r < m) u[r] = a[o + r][k];
            const T alpha = householder1<T, NT, FAST>(u, m, 0);
            if (WITH_U) {
#pragma unroll
                for (int r = 0; r < MX; ++r)
                    if (r < m) up[k][r] = u[r];
            }
            if constexpr (FAST) {
                // the same rank-2 update A -= u v^T + v u^T, v = 2 (A u - (u.Au) u), contracted to fma
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < m) {
                        T y = T(0);
#pragma unroll
                        for (int j = 0; j < MX; ++j)
                            if (j < m) y = fma_t(i >= j ? a[o + i][o + j] : a[o + j][o + i], u[j], y);
                        v[i] = y;
                    }
                T d = T(0);
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < m) d = fma_t(u[i], v[i], d);
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < m) v[i] = fma_t(-u[i], d, v[i]) * T(2);
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < m) {
#pragma unroll
                        for (int j = 0; j <= i; ++j)
                            a[o + i][o + j] = fma_t(-u[j], v[i], fma_t(-v[j], u[i], a[o + i][o + j]));
                    }
            } else {
#pragma unroll
            for (int i = 0; i < MX; ++i)
                if (i < m) {
                    T y = T(0);
#pragma unroll
                    for (int j = 0; j < MX; ++j)
                        if (j < m) y += (i >= j ? a[o + i][o + j] : a[o + j][o + i]) * u[j];
                    v[i] = y;
                }
            T d = T(0);
#pragma unroll
            for (int i = 0; i < MX; ++i)
                if (i < m) d += u[i] * v[i];
#pragma unroll
            for (int i = 0; i < MX; ++i)
                if (i < m) v[i] = (v[i] - u[i] * d) * T(2);
#pragma unroll
            for (int i = 0; i < MX; ++i)
                if (i < m) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        const T w = (i == j) ? (u[i] * v[i]) * T(2) : (u[j] * v[i] + v[j] * u[i]);
                        a[o + i][o + j] -= w;
                    }
                }
            }
            a[o][k] = alpha;
#pragma unroll
            for (int r = k + 2; r < MX; ++r)
                if (r < n) a[r][k] = T(0);
        }
    }
#pragma unroll
    for (int i = 0; i < MX; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j)
            if (i < n) a[j][i] = a[i][j];
}

// qr_hessenberg_ :432-454
template <typename T, int NT>
__device__ __forceinline__ void qr_hessenberg1(T (&a)[Dim<NT>::MAX][Dim<NT>::MAX],
                                               T (&q)[Dim<NT>::MAX][Dim<NT>::MAX], int n)
{
    constexpr int MX = Dim<NT>::MAX;
#pragma unroll
    for (int i = 0; i < MX; ++i)
#pragma unroll
        for (int j = 0; j < MX; ++j) q[i][j] = (i == j) ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < MX - 1; ++k) {
        if (k < n - 1) {
            T c, s;
            givens1(a[k][k], a[k + 1][k], c, s);
#pragma unroll
            for (int j = k; j < MX; ++j)
                if (j < n) rot1(a[k][j], a[k + 1][j], c, s);
#pragma unroll
            for (int i = 0; i < k + 2; ++i) rot1(q[i][k], q[i][k + 1], c, s);
        }
    }
}

// One R Q step on the leading m x m block; WITH_U also rotates the columns of u (n rows).
// sym: the tridiagonal shortcut of _rq_hessenberg_jit_ :457-485; otherwise the full ranges.
// FM: float32 sweep arithmetic (FastSweeps above); only eig_sym's sweeps ask for it.
template <typename T, int NT, bool WITH_U, bool FM = false>
__device__ __forceinline__ void rq_step1(T (&a)[Dim<NT>::MAX][Dim<NT>::MAX],
                                         T (&u)[Dim<NT>::MAX][Dim<NT>::MAX], int n, int m, bool sym)
{
    constexpr int MX = Dim<NT>::MAX;
    T lc[MX], ls[MX];
    auto giv = [](T x, T y, T &c, T &s) {
        if constexpr (FM) givens_fast1(x, y, c, s);
        else givens1(x, y, c, s);
    };
    auto rot = [](T &a0, T &a1, T c, T s) {
        if constexpr (FM) rot_fast1(a0, a1, c, s);
        else rot1(a0, a1, c, s);
    };
#pragma unroll
    for (int k = 0; k < MX - 1; ++k) {
        if (k < m - 1) {
            giv(a[k][k], a[k + 1][k], lc[k], ls[k]);
#pragma unroll
            for (int j = k; j < MX; ++j)
                if (j < m && (!sym || j < k + 3)) rot(a[k][j], a[k + 1][j], lc[k], ls[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < MX - 1; ++k) {
        if (k < m - 1) {
#pragma unroll
            for (int i = 0; i < k + 2; ++i)
                if (!sym || i >= k - 1) rot(a[i][k], a[i][k + 1], lc[k], ls[k]);
            if (WITH_U) {
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < n) rot(u[i][k], u[i][k + 1], lc[k], ls[k]);
            }
        }
    }
}

// _wilkinson :558-569 on the trailing 2x2 of the active m x m block
template <typename T>
__device__ __forceinline__ T wilkinson1(T h0, T h1, T b)
{
#pragma clang fp contract(off)
    const T b2 = b * b;
    T d = (h0 - h1) / T(2);
    const T s = (d < T(0)) ? T(-1) : T(1);
    d = fabs_(d) + sqrt_(d * d + b2);
    d = (d == T(0)) ? T(1) : d;
    return h1 - s * b2 / d;
}

// the same shift for the fast sweeps: hardware sqrt / rcp (a shift only steers the iteration)
template <typename T>
__device__ __forceinline__ T wilkinson_fast1(T h0, T h1, T b)
{
    const T b2 = b * b;
    const T d = (h0 - h1) * T(0.5);
    const T sb2 = (d < T(0)) ? -b2 : b2;
    const T t = fma_t(d, d, b2);
    const bool ok = t > FastRange<T>::lo && t < FastRange<T>::hi;
    T sigma = fma_t(-sb2, hw_rcp(fabs_(d) + hw_sqrt(t)), h1); // (explicit: the packed twin must contract the same way)
    if (__builtin_expect(__any(!ok), 0)) {
        const T s2 = wilkinson1<T>(h0, h1, b);
        sigma = ok ? sigma : s2;
    }
    return sigma;
}

// One explicitly shifted QR step T <- R Q + sigma on the leading m x m block of a SYMMETRIC
// TRIDIAGONAL matrix, the fast sweeps' form of `rq_step1(..., sym = true)`: the same rotations
// (c_k, s_k) and, in exact arithmetic, the same T' -- but only the diagonal and the sub-diagonal
// are carried (11 operations per rotation instead of six 4-operation row / column rotations):
//   p_0 = d_0 - sigma, q_0 = e_0;   (C_k, S_k, r_k) rotate (p_k, e_k) onto (r_k, 0);
//   u_k = C_k q_k + S_k (d_{k+1} - sigma);   p_{k+1} = C_k (d_{k+1} - sigma) - S_k q_k;   q_{k+1} = C_k e_{k+1};
//   d'_k = C_{k-1} C_k r_k + S_k u_k + sigma;   e'_{k-1} = S_{k-1} r_k;   d'_{m-1} = C_{m-2} p_{m-1} + sigma.
// (derivation and a numerical check against the explicit form: DESIGN.md section 4.2).  The upper
// sub-diagonal is kept equal to the lower one so that the storage stays a symmetric matrix.
template <typename T, int NT, bool WITH_U>
__device__ __forceinline__ void tri_sweep_fast1(T (&h)[Dim<NT>::MAX][Dim<NT>::MAX],
                                                T (&u)[Dim<NT>::MAX][Dim<NT>::MAX], int n, int m, T sigma)
{
    constexpr int MX = Dim<NT>::MAX;
    T p = h[0][0] - sigma, q = h[1][0];
    T cprev = T(1), sprev = T(0);
#pragma unroll
    for (int k = 0; k < MX - 1; ++k) {
        if (k < m - 1) {
            const T b = h[k + 1][k];
            const T a1 = h[k + 1][k + 1] - sigma;
            T c, sr; // the reference's convention: sr = -S
            givens_fast1(p, b, c, sr);
            const T S = -sr;
            const T r = fma_t(c, p, S * b);
            const T uk = fma_t(c, q, S * a1);
            const T pn = fma_t(c, a1, -(S * q));
            T qn = T(0);
            if (k + 2 < MX)
                if (k + 2 < m) qn = c * h[k + 2][k + 1];
            h[k][k] = fma_t(c * cprev, r, S * uk) + sigma;
            if (k > 0) {
                const T e = sprev * r;
                h[k][k - 1] = e;
                h[k - 1][k] = e;
            }
            if (WITH_U) {
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < n) rot_fast1(u[i][k], u[i][k + 1], c, sr);
            }
            p = pn;
            q = qn;
            cprev = c;
            sprev = S;
        }
    }
    const T e = sprev * p;
    h[m - 1][m - 1] = fma_t(cprev, p, sigma);
    h[m - 1][m - 2] = e;
    h[m - 2][m - 1] = e;
}

// The last stage of the deflation (m == 2) in the fast sweeps: the leading 2 x 2 block [a b; b d]
// is diagonalised by ONE Jacobi rotation instead of being iterated on -- a lockstep wavefront pays
// 3-4 shifted sweeps for its slowest lane there, a third of all the instructions of a 3 x 3 problem.
//   delta = (d - a) / 2,  r = sqrt(delta^2 + b^2),  t = sgn(delta) b / (|delta| + r)   (|t| <= 1),
//   a' = a - t b,  d' = d + t b;   (c, s) = (1, t) / sqrt(1 + t^2) rotates columns 0, 1 of U.
// b == 0 gives t = +-0 and (c, s) = (1, 0) exactly: a diagonal block comes back bit for bit.
// Lanes whose delta^2 + b^2 is outside the safe range of the hardware rsqrt (0, denormal, inf, NaN)
// are left untouched and reported: they take the iterative path.
template <typename T, int NT, bool WITH_U>
__device__ __forceinline__ bool jacobi2_fast1(T (&h)[Dim<NT>::MAX][Dim<NT>::MAX], T (&u)[Dim<NT>::MAX][Dim<NT>::MAX],
                                              int n)
{
    constexpr int MX = Dim<NT>::MAX;
    const T a = h[0][0], d = h[1][1], b = h[1][0];
    const T dl = (d - a) * T(0.5);
    const T r2 = fma_t(dl, dl, b * b);
    const bool ok = r2 > FastRange<T>::lo && r2 < FastRange<T>::hi;
    const T den = fma_t(r2, rsq_nr(r2), fabs_(dl));
    T inv = hw_rcp(den);
    inv = fma_t(fma_t(-den, inv, T(1)), inv, inv);
    const T t = ((dl < T(0)) ? -b : b) * inv;
    h[0][0] = ok ? fma_t(-t, b, a) : a;
    h[1][1] = ok ? fma_t(t, b, d) : d;
    h[1][0] = ok ? T(0) : b;
    h[0][1] = h[1][0];
    if (WITH_U) {
        T c = rsq_nr(fma_t(t, t, T(1)));
        T s = t * c;
        c = ok ? c : T(1);
        s = ok ? s : T(0);
#pragma unroll
        for (int i = 0; i < MX; ++i)
            if (i < n) rot_fast1(u[i][0], u[i][1], c, s);
    }
    return ok;
}

// _qr_explicit(_vectors)_jit_ :572-656 with sym = True; convergence per lane (Q9)
template <typename T, int NT, bool WITH_U, bool FAST = false>
__device__ __forceinline__ void qr_explicit1(T (&h)[Dim<NT>::MAX][Dim<NT>::MAX],
                                             T (&u)[Dim<NT>::MAX][Dim<NT>::MAX], int n, int max_iter, double tol)
{
#pragma clang fp contract(off)
    constexpr int MX = Dim<NT>::MAX;
    constexpr bool FM = FAST && FastSweeps<T>::on;
    // The reference deflates when e^2 < tol (d0^2 + d1^2) with tol = 1e-32 by default: |e| < 1e-16 |d|,
    // the working precision of float64 -- but eight orders below float32's, where it costs one more
    // sweep per eigenvalue just to square an off-diagonal that is already below half an ulp.  The fast
    // sweeps floor the tolerance at the working precision of the dtype, |e| <= eps/4 |d| (the neglected
    // entry moves an eigenvalue by at most |e|: a quarter of an ulp); a larger caller tolerance is kept.
    if constexpr (FM) {
        const double floor_ = sizeof(T) == 4 ? 0x1p-52 : 0x1p-110; // (eps / 4)^2, eps = 2^-24 / 2^-53
        tol = tol > floor_ ? tol : floor_;
    }
    const T tol_t = (T)tol, stuck_t = (T)(tol * 1e-3);
    if (WITH_U) {
#pragma unroll
        for (int i = 0; i < MX; ++i)
#pragma unroll
            for (int j = 0; j < MX; ++j) u[i][j] = (i == j) ? T(1) : T(0);
    }
#pragma unroll
    for (int m = MX; m >= 2; --m) {
        if (m <= n) {
            int iters = max_iter;
            if constexpr (FM) {
                if (m == 2 && max_iter > 0)
                    if (jacobi2_fast1<T, NT, WITH_U>(h, u, n)) iters = 0;
            }
            double sos_prev = 0.0;
            T ratio_prev = T(0);
            for (int it = 0; it < iters; ++it) {
                T sigma;
                if constexpr (FM) sigma = wilkinson_fast1(h[m - 2][m - 2], h[m - 1][m - 1], h[m - 1][m - 2]);
                else sigma = wilkinson1(h[m - 2][m - 2], h[m - 1][m - 1], h[m - 1][m - 2]);
                if constexpr (FM) {
                    tri_sweep_fast1<T, NT, WITH_U>(h, u, n, m, sigma);
                } else {
#pragma unroll
                    for (int i = 0; i < m; ++i) h[i][i] -= sigma;
                    rq_step1<T, NT, WITH_U, false>(h, u, n, m, true);
#pragma unroll
                    for (int i = 0; i < m; ++i) h[i][i] += sigma;
                }
                const T bb = fabs_(h[m - 1][m - 2]), a0 = fabs_(h[m - 1][m - 1]), a1 = fabs_(h[m - 2][m - 2]);
                const T sos_lower = bb * bb, sos_diag = a0 * a0 + a1 * a1;
                // `<=` (upstream: `<`) and the NaN test only matter when nothing can change any
                // more: a zero off-diagonal (diagonal or zero blocks, padding lanes of the last
                // tile) or NaNs would otherwise spin through all max_iter identical iterations
                bool conv;
                if constexpr (FM) conv = sos_lower <= tol_t * sos_diag; // in T: tol_t >= (eps/4)^2 is a normal number
                else conv = (double)sos_lower <= tol * (double)sos_diag;
                if (conv || sos_lower != sos_lower) {
#pragma unroll
                    for (int j = 0; j < m - 1; ++j) h[m - 1][j] = T(0);
                    break;
                }
                if constexpr (FM && !WITH_U) { // the same exit in T (the ratio only detects a fixed point)
                    const T ratio = sos_lower * hw_rcp(sos_diag);
                    const T dif = fabs_(ratio_prev - ratio);
                    if (ratio_prev != T(0) && dif < stuck_t * ratio_prev) break;
                    ratio_prev = ratio;
                } else if (!WITH_U) { // the "stuck" exit exists only in the no-vectors variant :648-653
                    // |prev - new| / prev < tol * 1e-3, written without the fp64 division (prev > 0)
                    double sos_new;
                    if constexpr (FM) // the ratio only detects a fixed point of the iteration
                        sos_new = (double)(sos_lower * hw_rcp(sos_diag));
                    else sos_new = (double)(sos_lower / sos_diag);
                    const double dif = sos_prev - sos_new;
                    if (sos_prev != 0.0 && (dif < 0 ? -dif : dif) < (tol * 1e-3) * sos_prev) break;
                    sos_prev = sos_new;
                }
            }
        }
    }
}

// apply P = I - 2 w w^T (w of length m, acting on the trailing m rows) from the left
// householder_apply_ :72-106, side='left'
template <typename T, int NT, bool FAST = false>
__device__ __forceinline__ void reflect_left1(T (&a)[Dim<NT>::MAX][Dim<NT>::MAX], int n, int m,
                                              const T (&w)[Dim<NT>::MAX])
{
#pragma clang fp contract(off)
    constexpr int MX = Dim<NT>::MAX;
    const int k0 = n - m;
    if constexpr (FAST) { // eig_sym's fast arithmetic: the same update contracted to fma
#pragma unroll
        for (int c = 0; c < MX; ++c)
            if (c < n) {
                T d = T(0);
#pragma unroll
                for (int r = 0; r < MX; ++r)
                    if (r >= k0 && r < n) d = fma_t(w[r - k0 < 0 ? 0 : r - k0], a[r][c], d);
                d += d;
#pragma unroll
                for (int r = 0; r < MX; ++r)
                    if (r >= k0 && r < n) a[r][c] = fma_t(-w[r - k0 < 0 ? 0 : r - k0], d, a[r][c]);
            }
        return;
    }
#pragma unroll
    for (int c = 0; c < MX; ++c)
        if (c < n) {
            T d = T(0);
#pragma unroll
            for (int r = 0; r < MX; ++r)
                if (r >= k0 && r < n) d += w[r - k0 < 0 ? 0 : r - k0] * a[r][c];
#pragma unroll
            for (int r = 0; r < MX; ++r)
                if (r >= k0 && r < n) a[r][c] -= T(2) * (w[r - k0 < 0 ? 0 : r - k0] * d);
        }
}

// _fwd_eig_sym :665-681.  `a` holds the symmetric input (the requested triangle already
// mirrored); on return vals = diagonal, and for WITH_U the columns of u are the eigenvectors.
template <typename T, int NT, bool WITH_U, bool FAST = false>
__device__ __forceinline__ void eig_sym1(T (&a)[Dim<NT>::MAX][Dim<NT>::MAX], T (&u)[Dim<NT>::MAX][Dim<NT>::MAX],
                                         int n, int max_iter, double tol)
{
    constexpr int MX = Dim<NT>::MAX;
    T up[MX][MX];
    hessenberg_sym1<T, NT, WITH_U, FAST>(a, n, up);
    qr_explicit1<T, NT, WITH_U, FAST>(a, u, n, max_iter, tol);
    if (WITH_U) {
        // householder_apply_(u, q, side='left', inverse=True): reflectors in reverse order
#pragma unroll
        for (int k = MX - 3; k >= 0; --k)
            if (k < n - 2) {
                T w[MX];
#pragma unroll
                for (int r = 0; r < MX; ++r) w[r] = up[k][r];
                reflect_left1<T, NT, FAST && FastSweeps<T>::on>(u, n, n - k - 1, w);
            }
    }
}

} // namespace qr
} // namespace nfm

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

// MATRIX STORAGE.  The routines below only ever write `a[i][j]`: the matrix arguments are template
// parameters (`MA`, `MU`, ...), bound either to a register array T[MAX][MAX] (compile-time orders: every
// index is a literal after unrolling) or to an LDS image of the lane's matrix (`LdsMat`: run-time orders
// 9..16 whose matrices do not fit the lane's registers -- element (i, j) of lane t lives at
// image[(i * ld + j) * LANES + t], consecutive lanes in consecutive banks, so every access of a wavefront is
// conflict-free whatever (i, j) is).  Vectors (reflectors, rotation lists, the band of the QR sweeps) are
// always register arrays indexed by literals.
template <typename T, int LANES>
struct LdsMat {
    T *p;   // &image[lane]
    int ld; // row stride in elements
    struct Row {
        T *p;
        __device__ __forceinline__ T &operator[](int j) const { return p[j * LANES]; }
    };
    __device__ __forceinline__ Row operator[](int i) const { return Row{p + i * ld * LANES}; }
};
template <class M>
struct IsLdsMat : std::false_type {};
template <typename T, int LANES>
struct IsLdsMat<LdsMat<T, LANES>> : std::true_type {};
// reflector k of a symmetric tridiagonalisation stored in row k of the matrix, right of the diagonal
// (n - 1 - k slots for its n - 1 - k values: the sweeps of eig_sym never read the upper triangle)
template <typename T, class M>
struct UpperRows {
    M m;
    struct Row {
        M m;
        int k;
        __device__ __forceinline__ T &operator[](int r) const { return m[k][k + 1 + r]; }
    };
    __device__ __forceinline__ Row operator[](int k) const { return Row{m, k}; }
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

// ---------------------------------------------------------------------------------------
// Correctly rounded division and square root WITHOUT their range handling: what eig_sym's
// reference-order arithmetic (FAST = false: the default, bit-identical to the CPU restatement) runs.
//
// hipcc's IEEE sequences are, for float32: division = 2 v_div_scale + v_rcp + 2 fma (reciprocal) +
// mul + 3 fma + v_div_fmas + v_div_fixup (11 instructions), square root = scale-up select, v_sqrt,
// two neighbours tested with one fma each, scale-down select, class select (16); float64: 11 and 20.
// v_div_scale / v_div_fixup and the scaling selects only act on operands near the ends of the exponent
// range (denormals, quotients that underflow, zeros, inf, NaN).  For operands inside the ranges of
// `CrRange` they are the identity, and what is left is the SAME arithmetic -- so the same correctly
// rounded bits -- in 8 (one division), 3 + 5 per numerator (a shared denominator; the numerator part
// packs to v_pk_* for float32) and 9 / 10 (square root) instructions.  Whether every active lane of
// the wavefront is inside the ranges is one vote per rotation / shift (`__any`: a uniform branch, no
// exec-mask juggling on the common path); otherwise the whole wavefront runs hipcc's full sequences,
// whose result for the in-range lanes is the same by construction.
// Ranges (a lane outside them is correct, just slower): squared norms / sqrt arguments in
// [2^-96, 2^40] (float64: [2^-760, 2^120]) -- denominators in [2^-48, 2^20] -- and numerators of
// magnitude >= 2^-102 (2^-960): no operand or quotient is denormal, no residual fma loses bits
// (v_div_scale's own criterion: numerator exponent > 23 / 52), exponent differences stay below 96 / 768.
// Zero numerators take the full sequence too (the trimmed one returns +0 for -0 / n).
template <typename T>
struct CrRange;
template <>
struct CrRange<float> {
    static constexpr float r2_lo = 0x1p-96f, r2_hi = 0x1p40f, num_lo = 0x1p-102f;
};
template <>
struct CrRange<double> {
    static constexpr double r2_lo = 0x1p-760, r2_hi = 0x1p120, num_lo = 0x1p-960;
};
template <typename T>
using V2 = T __attribute__((ext_vector_type(2)));

// lo <= x <= hi for x >= 0 as ONE unsigned compare of the (high) dword; NaN, inf and negative values fail
__device__ __forceinline__ bool cr_in_range(float x)
{
    constexpr unsigned lo = 0x0f800000u /* 2^-96 */, hi = 0x53800000u /* 2^40 */;
    static_assert(CrRange<float>::r2_lo == 0x1p-96f && CrRange<float>::r2_hi == 0x1p40f, "bit patterns above");
    return (__builtin_bit_cast(unsigned, x) - lo) <= (hi - lo);
}
__device__ __forceinline__ bool cr_in_range(double x)
{
    constexpr unsigned lo = (1023u - 760u) << 20, hi = (1023u + 120u) << 20;
    static_assert(CrRange<double>::r2_lo == 0x1p-760 && CrRange<double>::r2_hi == 0x1p120, "bit patterns above");
    return ((unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32) - lo) <= (hi - lo);
}

// sqrt: hipcc's float32 sequence without the 2^32 scaling of arguments below 2^-96 and the class select
__device__ __forceinline__ float sqrt_cr(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x); // <= 1 ulp
    const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}
// ... and the float64 one (v_rsq_f64 + a coupled Newton step + two residual corrections) without its ldexp pair
__device__ __forceinline__ double sqrt_cr(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
// refined reciprocal of the (shared) denominator: v_rcp + one (float64: two) Newton steps, as in hipcc's division
__device__ __forceinline__ float rcp_cr(float n)
{
    const float r = __builtin_amdgcn_rcpf(n);
    return __builtin_fmaf(__builtin_fmaf(-n, r, 1.0f), r, r);
}
__device__ __forceinline__ double rcp_cr(double n)
{
    double r = __builtin_amdgcn_rcp(n);
    r = __builtin_fma(r, __builtin_fma(-n, r, 1.0), r);
    return __builtin_fma(r, __builtin_fma(-n, r, 1.0), r);
}
// x / n given r = rcp_cr(n): quotient, residual, correction (float32: twice), the last fma is v_div_fmas unscaled
__device__ __forceinline__ float div_cr(float x, float n, float r)
{
    float q = x * r;
    q = __builtin_fmaf(__builtin_fmaf(-n, q, x), r, q);
    return __builtin_fmaf(__builtin_fmaf(-n, q, x), r, q);
}
__device__ __forceinline__ double div_cr(double x, double n, double r)
{
    const double q = x * r;
    return __builtin_fma(__builtin_fma(-n, q, x), r, q);
}
// (x, y) / n: the same steps on a register pair -- v_pk_mul_f32 / v_pk_fma_f32 for float32
__device__ __forceinline__ void div2_cr(float x, float y, float n, float &qx, float &qy)
{
    const float r = rcp_cr(n);
    const V2<float> xy = {x, y}, nn = {-n, -n}, rr = {r, r};
    V2<float> q = xy * rr;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nn, q, xy), rr, q);
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nn, q, xy), rr, q);
    qx = q.x;
    qy = q.y;
}
__device__ __forceinline__ void div2_cr(double x, double y, double n, double &qx, double &qy)
{
    const double r = rcp_cr(n);
    qx = div_cr(x, n, r);
    qy = div_cr(y, n, r);
}

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

// givens1 for eig_sym's reference-order sweeps: the same bits; the trimmed sequences when every active lane is in range
template <typename T>
__device__ __forceinline__ void givens_cr1(T x, T y, T &c, T &s)
{
#pragma clang fp contract(off)
    const T r2 = x * x + y * y;
    const T lo = fabs_(x) < fabs_(y) ? fabs_(x) : fabs_(y);
    const bool ok = cr_in_range(r2) && lo >= CrRange<T>::num_lo;
    if (__builtin_expect(__any(!ok), 0)) {
        givens1<T>(x, y, c, s);
        return;
    }
    const T nrm = sqrt_cr(r2);
    T sy;
    div2_cr(x, y, nrm, c, sy);
    s = -sy;
}

// rot1 on a register pair (a0, a1) <- (a0 c - s a1, a1 c + s a0).  float32: two v_pk_mul_f32 and one
// v_pk_add_f32 that swaps the halves of its second operand and negates the one added to the low half
// (op_sel / neg_lo) -- separately rounded products and sums, the same bits as rot1.
template <typename T>
__device__ __forceinline__ void rot_pair1(T &a0, T &a1, T c, T s)
{
#pragma clang fp contract(off)
    if constexpr (sizeof(T) == 4) {
        const V2<float> v = {a0, a1};
        const V2<float> t = v * s, w = v * c;
        V2<float> o;
#ifdef NFM_EXP_NOASM
        o.x = w.x - t.y; o.y = w.y + t.x;
#else
        // (device inline asm is `convergent` by default, which keeps the loops around it from unrolling early
        // and the matrix from being promoted to registers; this statement touches no other lane)
        [[clang::noconvergent]] {
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o) : "v"(w), "v"(t));
        }
#endif
        a0 = o.x;
        a1 = o.y;
    } else {
        rot1(a0, a1, c, s);
    }
}

// householder_ :55-69 on x[0..m), reflecting onto component `basis` (compile-time or not;
// the element is picked by a select so that registers are never indexed dynamically)
// TRIM (eig_sym's reference-order arithmetic): the two square roots and the m divisions run the trimmed
// correctly rounded sequences (CrRange above) when every active lane is in range -- the same bits
template <typename T, int NT, bool FAST = false, bool TRIM = false>
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
    if constexpr (TRIM) {
        // in range: ss (and the second sum of squares, <= 4 ss) a valid sqrt argument / denominator, every
        // component a valid numerator (the reflected one becomes |x_b| + |x| >= the denominator's floor)
        T lo = fabs_(x[0]);
#pragma unroll
        for (int i = 1; i < Dim<NT>::MAX; ++i)
            if (i < m) lo = fabs_(x[i]) < lo ? fabs_(x[i]) : lo;
        const bool ok = cr_in_range(ss * T(4)) && cr_in_range(ss) && lo >= CrRange<T>::num_lo;
        if (!__builtin_expect(__any(!ok), 0)) {
            rho *= sqrt_cr(ss);
#pragma unroll
            for (int i = 0; i < Dim<NT>::MAX; ++i)
                if (i < m) x[i] = (i == basis) ? x[i] - rho : x[i];
            ss = T(0);
#pragma unroll
            for (int i = 0; i < Dim<NT>::MAX; ++i)
                if (i < m) ss += x[i] * x[i];
            const T nrm = sqrt_cr(ss), rn = rcp_cr(nrm);
#pragma unroll
            for (int i = 0; i < Dim<NT>::MAX; ++i)
                if (i < m) x[i] = div_cr(x[i], nrm, rn);
            return rho;
        }
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
template <typename T, int NT, bool WITH_U, class MA, class MU>
__device__ __forceinline__ void hessenberg1(MA &a, int n, MU &up)
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
// FILL = false (eig_sym): the upper triangle is left alone (the sweeps run on the band d, e taken from the
// lower half, and the LDS form keeps the reflectors there)
template <typename T, int NT, bool WITH_U, bool FAST = false, bool TRIM = false, bool FILL = true, class MA, class MU>
__device__ __forceinline__ void hessenberg_sym1(MA &a, int n, MU &up)
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
            const T alpha = householder1<T, NT, FAST, TRIM>(u, m, 0);
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
    if constexpr (FILL) {
#pragma unroll
        for (int i = 0; i < MX; ++i)
#pragma unroll
            for (int j = 0; j < i; ++j)
                if (i < n) a[j][i] = a[i][j];
    }
}

// qr_hessenberg_ :432-454
template <typename T, int NT, class MA, class MQ>
__device__ __forceinline__ void qr_hessenberg1(MA &a, MQ &q, int n)
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
template <typename T, int NT, bool WITH_U, bool FM = false, class MA, class MU>
__device__ __forceinline__ void rq_step1(MA &a, MU &u, int n, int m, bool sym)
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

// wilkinson1 with the trimmed sequences on a vote (same bits)
template <typename T>
__device__ __forceinline__ T wilkinson_cr1(T h0, T h1, T b)
{
#pragma clang fp contract(off)
    const T b2 = b * b;
    const T d = (h0 - h1) * T(0.5); // == / 2, exactly
    const T t = d * d + b2;
    const bool ok = cr_in_range(t) && b2 >= CrRange<T>::num_lo;
    if (__builtin_expect(__any(!ok), 0)) return wilkinson1<T>(h0, h1, b);
    const T den = fabs_(d) + sqrt_cr(t); // > 0 in range
    const T sb2 = (d < T(0)) ? -b2 : b2;
    return h1 - div_cr(sb2, den, rcp_cr(den));
}

// rq_step1(..., sym = true) of eig_sym's reference-order sweeps on BAND storage.  The tridiagonal shortcut
// of _rq_hessenberg_jit_ :457-485 rotates rows k, k+1 over columns k..k+2 and columns k, k+1 over rows
// k-1..k+1: only four diagonals of the matrix are ever read or written --
//   d[i] = a[i][i],  l[i] = a[i+1][i],  u1[i] = a[i][i+1],  u2[i] = a[i][i+2]
// (l and u1 drift apart by rounding and u2 fills with rounding residue: all three are carried because
// the reference carries them).  The same operations on the same entries, so the same bits, in 4 n
// registers instead of n^2; rotations by givens_cr1 / rot_pair1.  Loop bounds are literal (the vote in
// givens_cr1 is a convergent operation: loops around one only unroll early when their trip count is).
template <typename T, int NT, bool WITH_U, class MU>
__device__ __forceinline__ void band_sweep_cr1(T (&d)[Dim<NT>::MAX], T (&l)[Dim<NT>::MAX], T (&u1)[Dim<NT>::MAX],
                                               T (&u2)[Dim<NT>::MAX], MU &u, int n, int m)
{
    constexpr int MX = Dim<NT>::MAX;
    T lc[MX], ls[MX];
#pragma unroll
    for (int k = 0; k < MX - 1; ++k) {
        if (k < m - 1) {
            givens_cr1(d[k], l[k], lc[k], ls[k]);
            rot_pair1(d[k], l[k], lc[k], ls[k]);          // column k:   (a[k][k],   a[k+1][k])
            rot_pair1(u1[k], d[k + 1], lc[k], ls[k]);     // column k+1: (a[k][k+1], a[k+1][k+1])
            if (k + 2 < MX)
                if (k + 2 < m) rot_pair1(u2[k], u1[k + 1], lc[k], ls[k]); // column k+2: (a[k][k+2], a[k+1][k+2])
        }
    }
    // U <- U Q: rotation k combines columns k, k+1.  float64: column k+1 is carried in registers from one rotation
    // to the next (uc), so that a U kept in LDS is read and written once per column and sweep instead of twice (for
    // a U in registers this is the same dataflow as rotating in place, and the form that allocates fewer registers:
    // 8x8 float32 with vectors 252 against 323, two wavefronts per SIMD against one).  A float32 U kept in LDS is
    // rotated in place: measured, the carried form of the packed rotation was 1.7x SLOWER there.
    if constexpr (WITH_U && sizeof(T) == 4 && IsLdsMat<MU>::value) {
#pragma unroll
        for (int k = 0; k < MX - 1; ++k) {
            if (k < m - 1) {
                if (k >= 1) rot_pair1(u1[k - 1 < 0 ? 0 : k - 1], u2[k - 1 < 0 ? 0 : k - 1], lc[k], ls[k]); // row k-1
                rot_pair1(d[k], u1[k], lc[k], ls[k]);         // row k:   (a[k][k],   a[k][k+1])
                rot_pair1(l[k], d[k + 1], lc[k], ls[k]);      // row k+1: (a[k+1][k], a[k+1][k+1])
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < n) rot_pair1(u[i][k], u[i][k + 1], lc[k], ls[k]);
            }
        }
        return;
    }
    T uc[MX];
    if (WITH_U) {
#pragma unroll
        for (int i = 0; i < MX; ++i) uc[i] = (i < n) ? u[i][0] : T(0);
    }
#pragma unroll
    for (int k = 0; k < MX - 1; ++k) {
        if (k < m - 1) {
            if (k >= 1) rot_pair1(u1[k - 1 < 0 ? 0 : k - 1], u2[k - 1 < 0 ? 0 : k - 1], lc[k], ls[k]); // row k-1
            rot_pair1(d[k], u1[k], lc[k], ls[k]);         // row k:   (a[k][k],   a[k][k+1])
            rot_pair1(l[k], d[k + 1], lc[k], ls[k]);      // row k+1: (a[k+1][k], a[k+1][k+1])
            if (WITH_U) {
                T un[MX];
#pragma unroll
                for (int i = 0; i < MX; ++i) un[i] = (i < n) ? u[i][k + 1] : T(0);
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < n) {
                        rot_pair1(uc[i], un[i], lc[k], ls[k]);
                        u[i][k] = uc[i];
                        uc[i] = un[i];
                    }
            }
        }
    }
    if (WITH_U) { // the last column the sweep touched
#pragma unroll
        for (int i = 0; i < MX; ++i)
            if (i < n) u[i][m - 1] = uc[i];
    }
}

// _qr_explicit(_vectors)_jit_ :572-656 with sym = True in the reference's operation order on band storage
// (band_sweep_cr1); convergence per lane (Q9).  d, e: the diagonal and the sub-diagonal of the symmetric
// tridiagonal matrix (the reference's upper sub-diagonal starts as a copy of e, its second one as zeros);
// on return d holds the eigenvalues.
template <typename T, int NT, bool WITH_U, class MU>
__device__ __forceinline__ void qr_explicit_band1(T (&d)[Dim<NT>::MAX], T (&l)[Dim<NT>::MAX], MU &u, int n, int max_iter,
                                                  double tol)
{
#pragma clang fp contract(off)
    constexpr int MX = Dim<NT>::MAX;
    // the exact "stuck" test can be screened by an estimate when its threshold is below half an ulp of T
    const bool screen_stuck = tol * 1e-3 < (sizeof(T) == 4 ? 0x1p-25 : 0x1p-54);
    T u1[MX], u2[MX];
#pragma unroll
    for (int i = 0; i < MX; ++i) {
        u1[i] = l[i];
        u2[i] = T(0);
    }
    if (WITH_U) {
#pragma unroll
        for (int i = 0; i < MX; ++i)
#pragma unroll
            for (int j = 0; j < MX; ++j)
                if (i < n && j < n) u[i][j] = (i == j) ? T(1) : T(0);
    }
#pragma unroll
    for (int m = MX; m >= 2; --m) {
        if (m <= n) {
            double sos_prev = 0.0;
            T ratio_prev = T(0), low_prev = T(0), diag_prev = T(1);
            for (int it = 0; it < max_iter; ++it) {
                const T sigma = wilkinson_cr1(d[m - 2], d[m - 1], l[m - 2]);
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < m) d[i] -= sigma;
                band_sweep_cr1<T, NT, WITH_U>(d, l, u1, u2, u, n, m);
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < m) d[i] += sigma;
                const T bb = fabs_(l[m - 2]), a0 = fabs_(d[m - 1]), a1 = fabs_(d[m - 2]);
                const T sos_lower = bb * bb, sos_diag = a0 * a0 + a1 * a1;
                // `<=` (upstream: `<`) and the NaN test only matter when nothing can change any
                // more: a zero off-diagonal (diagonal or zero blocks, padding lanes of the last
                // tile) or NaNs would otherwise spin through all max_iter identical iterations
                if ((double)sos_lower <= tol * (double)sos_diag || sos_lower != sos_lower) {
                    l[m - 2] = T(0); // h[m-1][:m-1] = 0: the only entry of that row the band holds
                    break;
                }
                if (!WITH_U) { // the "stuck" exit exists only in the no-vectors variant :648-653
                    // |prev - new| / prev < tol * 1e-3 with new = sos_lower / sos_diag correctly rounded in T,
                    // written without the fp64 division (prev > 0).  Below a relative threshold of one ulp of T
                    // the exit can only fire when the two quotients are EQUAL, so the division itself is only
                    // run when an estimate (v_rcp: a few ulp) says they may be: a wavefront vote, the
                    // reference's decision bit for bit either way.
                    if (screen_stuck) {
                        const T ratio = sos_lower * hw_rcp(sos_diag);
                        const bool far = fabs_(ratio - ratio_prev) > ratio_prev * T(0x1p-18); // NaN / inf: not far
                        const T lp = low_prev, dp = diag_prev;
                        ratio_prev = ratio;
                        low_prev = sos_lower;
                        diag_prev = sos_diag;
                        if (__builtin_expect(__any(!far), 0)) {
                            const double snew = (double)(sos_lower / sos_diag);
                            const double sprev = it > 0 ? (double)(lp / dp) : 0.0;
                            const double dif = sprev - snew;
                            if (!far && sprev != 0.0 && (dif < 0 ? -dif : dif) < (tol * 1e-3) * sprev) break;
                        }
                    } else {
                        const double sos_new = (double)(sos_lower / sos_diag);
                        const double dif = sos_prev - sos_new;
                        if (sos_prev != 0.0 && (dif < 0 ? -dif : dif) < (tol * 1e-3) * sos_prev) break;
                        sos_prev = sos_new;
                    }
                }
            }
        }
    }
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
// (c_k, s_k) and, in exact arithmetic, the same T' -- but only the diagonal d and the sub-diagonal e
// are carried (11 operations per rotation instead of six 4-operation row / column rotations):
//   p_0 = d_0 - sigma, q_0 = e_0;   (C_k, S_k, r_k) rotate (p_k, e_k) onto (r_k, 0);
//   u_k = C_k q_k + S_k (d_{k+1} - sigma);   p_{k+1} = C_k (d_{k+1} - sigma) - S_k q_k;   q_{k+1} = C_k e_{k+1};
//   d'_k = C_{k-1} C_k r_k + S_k u_k + sigma;   e'_{k-1} = S_{k-1} r_k;   d'_{m-1} = C_{m-2} p_{m-1} + sigma.
// (derivation and a numerical check against the explicit form: DESIGN.md section 4.2)
template <typename T, int NT, bool WITH_U, class MU>
__device__ __forceinline__ void tri_sweep_fast1(T (&d)[Dim<NT>::MAX], T (&e)[Dim<NT>::MAX], MU &u, int n, int m, T sigma)
{
    constexpr int MX = Dim<NT>::MAX;
    T p = d[0] - sigma, q = e[0];
    T cprev = T(1), sprev = T(0);
    T uc[MX]; // column k of U, carried from one rotation to the next (band_sweep_cr1)
    if (WITH_U) {
#pragma unroll
        for (int i = 0; i < MX; ++i) uc[i] = (i < n) ? u[i][0] : T(0);
    }
#pragma unroll
    for (int k = 0; k < MX - 1; ++k) {
        if (k < m - 1) {
            const T b = e[k];
            const T a1 = d[k + 1] - sigma;
            T c, sr; // the reference's convention: sr = -S
            givens_fast1(p, b, c, sr);
            const T S = -sr;
            const T r = fma_t(c, p, S * b);
            const T uk = fma_t(c, q, S * a1);
            const T pn = fma_t(c, a1, -(S * q));
            T qn = T(0);
            if (k + 2 < MX)
                if (k + 2 < m) qn = c * e[k + 1];
            d[k] = fma_t(c * cprev, r, S * uk) + sigma;
            if (k > 0) e[k - 1 < 0 ? 0 : k - 1] = sprev * r;
            if (WITH_U) {
                T un[MX];
#pragma unroll
                for (int i = 0; i < MX; ++i) un[i] = (i < n) ? u[i][k + 1] : T(0);
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < n) {
                        rot_fast1(uc[i], un[i], c, sr);
                        u[i][k] = uc[i];
                        uc[i] = un[i];
                    }
            }
            p = pn;
            q = qn;
            cprev = c;
            sprev = S;
        }
    }
    if (WITH_U) {
#pragma unroll
        for (int i = 0; i < MX; ++i)
            if (i < n) u[i][m - 1] = uc[i];
    }
    d[m - 1] = fma_t(cprev, p, sigma);
    e[m - 2] = sprev * p;
}

// The last stage of the deflation (m == 2) in the fast sweeps: the leading 2 x 2 block [a b; b d]
// is diagonalised by ONE Jacobi rotation instead of being iterated on -- a lockstep wavefront pays
// 3-4 shifted sweeps for its slowest lane there, a third of all the instructions of a 3 x 3 problem.
//   delta = (d - a) / 2,  r = sqrt(delta^2 + b^2),  t = sgn(delta) b / (|delta| + r)   (|t| <= 1),
//   a' = a - t b,  d' = d + t b;   (c, s) = (1, t) / sqrt(1 + t^2) rotates columns 0, 1 of U.
// b == 0 gives t = +-0 and (c, s) = (1, 0) exactly: a diagonal block comes back bit for bit.
// Lanes whose delta^2 + b^2 is outside the safe range of the hardware rsqrt (0, denormal, inf, NaN)
// are left untouched and reported: they take the iterative path.
template <typename T, int NT, bool WITH_U, class MU>
__device__ __forceinline__ bool jacobi2_fast1(T (&dg)[Dim<NT>::MAX], T (&e)[Dim<NT>::MAX], MU &u, int n)
{
    constexpr int MX = Dim<NT>::MAX;
    const T a = dg[0], d = dg[1], b = e[0];
    const T dl = (d - a) * T(0.5);
    const T r2 = fma_t(dl, dl, b * b);
    const bool ok = r2 > FastRange<T>::lo && r2 < FastRange<T>::hi;
    const T den = fma_t(r2, rsq_nr(r2), fabs_(dl));
    T inv = hw_rcp(den);
    inv = fma_t(fma_t(-den, inv, T(1)), inv, inv);
    const T t = ((dl < T(0)) ? -b : b) * inv;
    dg[0] = ok ? fma_t(-t, b, a) : a;
    dg[1] = ok ? fma_t(t, b, d) : d;
    e[0] = ok ? T(0) : b;
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

// the fast sweeps of eig_sym (policy above) on band storage d, e; convergence per lane (Q9)
template <typename T, int NT, bool WITH_U, class MU>
__device__ __forceinline__ void qr_fast_band1(T (&d)[Dim<NT>::MAX], T (&e)[Dim<NT>::MAX], MU &u, int n, int max_iter,
                                              double tol)
{
    constexpr int MX = Dim<NT>::MAX;
    // The reference deflates when e^2 < tol (d0^2 + d1^2) with tol = 1e-32 by default: |e| < 1e-16 |d|,
    // the working precision of float64 -- but eight orders below float32's, where it costs one more
    // sweep per eigenvalue just to square an off-diagonal that is already below half an ulp.  The fast
    // sweeps floor the tolerance at the working precision of the dtype, |e| <= eps/4 |d| (the neglected
    // entry moves an eigenvalue by at most |e|: a quarter of an ulp); a larger caller tolerance is kept.
    const double floor_ = sizeof(T) == 4 ? 0x1p-52 : 0x1p-110; // (eps / 4)^2, eps = 2^-24 / 2^-53
    tol = tol > floor_ ? tol : floor_;
    const T tol_t = (T)tol, stuck_t = (T)(tol * 1e-3);
    if (WITH_U) {
#pragma unroll
        for (int i = 0; i < MX; ++i)
#pragma unroll
            for (int j = 0; j < MX; ++j)
                if (i < n && j < n) u[i][j] = (i == j) ? T(1) : T(0);
    }
#pragma unroll
    for (int m = MX; m >= 2; --m) {
        if (m <= n) {
            int iters = max_iter;
            if (m == 2 && max_iter > 0)
                if (jacobi2_fast1<T, NT, WITH_U>(d, e, u, n)) iters = 0;
            T ratio_prev = T(0);
            for (int it = 0; it < iters; ++it) {
                const T sigma = wilkinson_fast1(d[m - 2], d[m - 1], e[m - 2]);
                tri_sweep_fast1<T, NT, WITH_U>(d, e, u, n, m, sigma);
                const T bb = fabs_(e[m - 2]), a0 = fabs_(d[m - 1]), a1 = fabs_(d[m - 2]);
                const T sos_lower = bb * bb, sos_diag = a0 * a0 + a1 * a1;
                // `<=` (upstream: `<`) and the NaN test only matter when nothing can change any
                // more: a zero off-diagonal (diagonal or zero blocks, padding lanes of the last
                // tile) or NaNs would otherwise spin through all max_iter identical iterations
                const bool conv = sos_lower <= tol_t * sos_diag; // in T: tol_t >= (eps/4)^2 is a normal number
                if (conv || sos_lower != sos_lower) {
                    e[m - 2] = T(0);
                    break;
                }
                if constexpr (!WITH_U) { // the "stuck" exit :648-653 in T (the ratio only detects a fixed point)
                    const T ratio = sos_lower * hw_rcp(sos_diag);
                    const T dif = fabs_(ratio_prev - ratio);
                    if (ratio_prev != T(0) && dif < stuck_t * ratio_prev) break;
                    ratio_prev = ratio;
                }
            }
        }
    }
}

// apply P = I - 2 w w^T (w of length m, acting on the trailing m rows) from the left
// householder_apply_ :72-106, side='left'.  K0 = n - m, the first row touched, is a LITERAL for the caller
// (eig_sym1: reflector k acts on rows k+1..), so that w is indexed by literals.
template <typename T, int NT, bool FAST = false, class MA>
__device__ __forceinline__ void reflect_left1(MA &a, int n, int k0, const T (&w)[Dim<NT>::MAX])
{
#pragma clang fp contract(off)
    constexpr int MX = Dim<NT>::MAX;
    // (every column is read ONCE into `col`, updated and written back: a matrix kept in LDS is not read twice)
    if constexpr (FAST) { // eig_sym's fast arithmetic: the same update contracted to fma
#pragma unroll
        for (int c = 0; c < MX; ++c)
            if (c < n) {
                T col[MX];
#pragma unroll
                for (int r = 0; r < MX; ++r) col[r] = (r >= k0 && r < n) ? a[r][c] : T(0);
                T d = T(0);
#pragma unroll
                for (int r = 0; r < MX; ++r)
                    if (r >= k0 && r < n) d = fma_t(w[r - k0 < 0 ? 0 : r - k0], col[r], d);
                d += d;
#pragma unroll
                for (int r = 0; r < MX; ++r)
                    if (r >= k0 && r < n) a[r][c] = fma_t(-w[r - k0 < 0 ? 0 : r - k0], d, col[r]);
            }
        return;
    }
#pragma unroll
    for (int c = 0; c < MX; ++c)
        if (c < n) {
            T col[MX];
#pragma unroll
            for (int r = 0; r < MX; ++r) col[r] = (r >= k0 && r < n) ? a[r][c] : T(0);
            T d = T(0);
#pragma unroll
            for (int r = 0; r < MX; ++r)
                if (r >= k0 && r < n) d += w[r - k0 < 0 ? 0 : r - k0] * col[r];
#pragma unroll
            for (int r = 0; r < MX; ++r)
                if (r >= k0 && r < n) a[r][c] = col[r] - T(2) * (w[r - k0 < 0 ? 0 : r - k0] * d);
        }
}

// _fwd_eig_sym :665-681.  `a` holds the symmetric input in its LOWER triangle (the caller mirrors the
// requested one on load; the upper triangle is never read); `up` is where the reflectors go (WITH_U).  On
// return vals = eigenvalues in deflation order, and for WITH_U the columns of u are the eigenvectors.
template <typename T, int NT, bool WITH_U, bool FAST = false, class MA, class MU, class MP>
__device__ __forceinline__ void eig_sym1(MA &a, MU &u, MP &up, T (&vals)[Dim<NT>::MAX], int n, int max_iter, double tol)
{
    constexpr int MX = Dim<NT>::MAX;
    constexpr bool FM = FAST && FastSweeps<T>::on;
    hessenberg_sym1<T, NT, WITH_U, FAST, !FAST, false>(a, n, up);
    T e[MX];
#pragma unroll
    for (int i = 0; i < MX; ++i) {
        vals[i] = (i < n) ? a[i][i] : T(0);
        e[i] = T(0);
        if (i + 1 < MX)
            if (i + 1 < n) e[i] = a[i + 1 < MX ? i + 1 : i][i];
    }
    if constexpr (FM) qr_fast_band1<T, NT, WITH_U>(vals, e, u, n, max_iter, tol);
    else qr_explicit_band1<T, NT, WITH_U>(vals, e, u, n, max_iter, tol);
    if (WITH_U) {
        // householder_apply_(u, q, side='left', inverse=True): reflectors in reverse order
#pragma unroll
        for (int k = MX - 3; k >= 0; --k)
            if (k < n - 2) {
                T w[MX];
#pragma unroll
                for (int r = 0; r < MX; ++r) w[r] = (r < n - 1 - k) ? up[k][r] : T(0);
                reflect_left1<T, NT, FM>(u, n, k + 1, w);
            }
    }
}

} // namespace qr
} // namespace nfm

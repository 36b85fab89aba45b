// nfm_sym.hip -- compact-symmetric entry points (sym_solve / matvec / invert / det /
// to_full / outer / matmul) for orders 1..8 in registers; orders 9..16 are in
// nfm_big.hip.  See include/nfm_hip.h for the ABI and the reference lines replaced.
#include "nfm_record_kernel.hpp"
#include "nfm_smallmat.hpp"
#include "nfm_big.hpp"

namespace nfm {

constexpr int mat_comps(int kind, int M)
{
    return kind == NFM_MAT_SYM ? sym_k(M) : kind == NFM_MAT_DIAG ? M : kind == NFM_MAT_SCAL ? 1 : M * M;
}

template <int KIND, int M>
using MatRec = Rec<(KIND == NFM_MAT_FULL ? M : 1), (KIND == NFM_MAT_FULL ? M : mat_comps(KIND, M))>;

struct SolveParams {
    double eps[NFM_MAX_DIM];
    int has_eps;
};

// ---- x = A \ v ---------------------------------------------------------------------
template <typename T, int M, int KIND>
struct SolveOp {
    using RA = MatRec<KIND, M>;
    using RB = Rec<1, M>;
    using RC = NoRec;
    using RO = Rec<1, M>;
    using Params = SolveParams;
    static constexpr int TILE = pick_tile((RA::C + RB::C) * (int)sizeof(T) + 16);
    static __device__ __forceinline__ void apply(T (&a)[RA::Cs], const T (&v)[M], const T (&)[1], T (&x)[M],
                                                 const Params &p)
    {
        if (p.has_eps) { // smoothing term on the diagonal (_impl/sym.py:356-357)
            if constexpr (KIND == NFM_MAT_FULL) {
#pragma unroll
                for (int i = 0; i < M; ++i) a[i * M + i] += (T)p.eps[i];
            } else if constexpr (KIND == NFM_MAT_SCAL) {
                // a scaled identity plus per-component eps is a diagonal: handled below
            } else {
#pragma unroll
                for (int i = 0; i < M; ++i) a[i] += (T)p.eps[i];
            }
        }
        if constexpr (KIND == NFM_MAT_SYM) {
            if constexpr (M <= 4) {
                sym_solve_closed<T, M>(a, v, x);
            } else {
                T f[M][M], b[M][1];
                sym_expand<T, M>(a, f);
#pragma unroll
                for (int i = 0; i < M; ++i) b[i][0] = v[i];
                ge_solve<T, M, 1>(f, b);
#pragma unroll
                for (int i = 0; i < M; ++i) x[i] = b[i][0];
            }
        } else if constexpr (KIND == NFM_MAT_DIAG) {
#pragma unroll
            for (int i = 0; i < M; ++i) x[i] = v[i] / a[i];
        } else if constexpr (KIND == NFM_MAT_SCAL) {
#pragma unroll
            for (int i = 0; i < M; ++i) x[i] = v[i] / (p.has_eps ? a[0] + (T)p.eps[i] : a[0]);
        } else {
            T f[M][M], b[M][1];
#pragma unroll
            for (int i = 0; i < M; ++i) {
#pragma unroll
                for (int j = 0; j < M; ++j) f[i][j] = a[i * M + j];
                b[i][0] = v[i];
            }
            ge_solve<T, M, 1>(f, b);
#pragma unroll
            for (int i = 0; i < M; ++i) x[i] = b[i][0];
        }
    }
};

// ---- y = [inp +/-] A v ---------------------------------------------------------------
struct MatvecParams {
    int mode;
};

template <typename T, int M, int KIND>
struct MatvecOp {
    using RA = MatRec<KIND, M>;
    using RB = Rec<1, M>;
    using RC = Rec<1, M>;
    using RO = Rec<1, M>;
    using Params = MatvecParams;
    static constexpr int TILE = pick_tile((RA::C + 2 * M) * (int)sizeof(T) + 32);
    static __device__ __forceinline__ void apply(const T (&a)[RA::Cs], const T (&v)[M], const T (&inp)[M],
                                                 T (&y)[M], const Params &p)
    {
#pragma clang fp contract(off)
        T r[M];
        if constexpr (KIND == NFM_MAT_SYM) {
            sym_matvec_compact<T, M>(a, v, r);
        } else if constexpr (KIND == NFM_MAT_DIAG) {
#pragma unroll
            for (int i = 0; i < M; ++i) r[i] = a[i] * v[i];
        } else if constexpr (KIND == NFM_MAT_SCAL) {
#pragma unroll
            for (int i = 0; i < M; ++i) r[i] = a[0] * v[i];
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                T s = a[i * M] * v[0];
#pragma unroll
                for (int j = 1; j < M; ++j) s = s + a[i * M + j] * v[j];
                r[i] = s;
            }
        }
#pragma unroll
        for (int i = 0; i < M; ++i) y[i] = p.mode > 0 ? inp[i] + r[i] : (p.mode < 0 ? inp[i] - r[i] : r[i]);
    }
};

// ---- compact inverse -----------------------------------------------------------------
struct NoParams {
    int unused;
};

template <typename T, int M, bool DIAG>
struct InvertOp {
    using RA = Rec<1, sym_k(M)>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, (DIAG ? M : sym_k(M))>;
    using Params = NoParams;
    static constexpr int TILE = pick_tile((RA::C + (DIAG ? M : 0)) * (int)sizeof(T) + 16);
    static __device__ __forceinline__ void apply(const T (&a)[RA::Cs], const T (&)[1], const T (&)[1],
                                                 T (&r)[RO::Cs], const Params &)
    {
        T inv[sym_k(M)];
        if constexpr (M <= 4) {
            sym_invert_closed<T, M>(a, inv);
        } else {
            T f[M][M];
            sym_expand<T, M>(a, f);
            gj_inverse<T, M>(f);
            // the reference fills entry (i, j), i < j, from column i of the inverse
            // (solve against e_i, element j): that is inv[j][i]
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = i; j < M; ++j) inv[sym_idx(M, i, j)] = f[j][i];
        }
#pragma unroll
        for (int i = 0; i < RO::C; ++i) r[i] = inv[i];
    }
};

// ---- determinant ---------------------------------------------------------------------
template <typename T, int M>
struct DetOp {
    using RA = Rec<1, sym_k(M)>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, 1>;
    using Params = NoParams;
    static constexpr int TILE = pick_tile(RA::C * (int)sizeof(T) + 16);
    static __device__ __forceinline__ void apply(const T (&a)[RA::Cs], const T (&)[1], const T (&)[1], T (&r)[1],
                                                 const Params &)
    {
        if constexpr (M == 1) r[0] = a[0];
        else if constexpr (M == 2) r[0] = sym_det2(&a[0], &a[2]);
        else if constexpr (M == 3) r[0] = sym_det3(&a[0], &a[3]);
        else if constexpr (M == 4) r[0] = sym_det4(&a[0], &a[4]);
        else {
            T f[M][M];
            sym_expand<T, M>(a, f);
            r[0] = lu_det<T, M>(f);
        }
    }
};

// ---- compact -> full -----------------------------------------------------------------
template <typename T, int M>
struct ToFullOp {
    using RA = Rec<1, sym_k(M)>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<M, M>;
    using Params = NoParams;
    static constexpr int TILE = pick_tile((RA::C + RO::C) * (int)sizeof(T) + 32);
    static __device__ __forceinline__ void apply(const T (&a)[RA::Cs], const T (&)[1], const T (&)[1],
                                                 T (&r)[RO::Cs], const Params &)
    {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) r[i * M + j] = a[sym_idx(M, i, j)];
    }
};

// ---- x x^T ---------------------------------------------------------------------------
template <typename T, int M>
struct OuterOp {
    using RA = Rec<1, M>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, sym_k(M)>;
    using Params = NoParams;
    static constexpr int TILE = pick_tile((RA::C + RO::C) * (int)sizeof(T) + 32);
    static __device__ __forceinline__ void apply(const T (&x)[M], const T (&)[1], const T (&)[1], T (&r)[RO::Cs],
                                                 const Params &)
    {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) r[sym_idx(M, i, j)] = x[i] * x[j];
    }
};

// ---- x y^T + y x^T in "gradient of a compact matrix" convention ------------------------
// out_ii = x_i y_i, out_ij = x_i y_j + x_j y_i (i < j): the pull-back of a full-matrix
// cotangent x y^T onto compact storage, where one stored entry stands for both (i, j) and
// (j, i).  Used by the backward passes of sym_matvec / sym_solve; `neg` flips the sign.
struct Outer2Params {
    int neg;
};

template <typename T, int M>
struct Outer2Op {
    using RA = Rec<1, M>;
    using RB = Rec<1, M>;
    using RC = NoRec;
    using RO = Rec<1, sym_k(M)>;
    using Params = Outer2Params;
    static constexpr int TILE = pick_tile((2 * M + RO::C) * (int)sizeof(T) + 48);
    static __device__ __forceinline__ void apply(const T (&x)[M], const T (&y)[M], const T (&)[1], T (&r)[RO::Cs],
                                                 const Params &p)
    {
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) {
                const T v = (i == j) ? x[i] * y[i] : x[i] * y[j] + x[j] * y[i];
                r[sym_idx(M, i, j)] = p.neg ? -v : v;
            }
    }
};

// ---- J^T H J (compact) ---------------------------------------------------------------
// _impl/sym.py:531-670.  jac (K x D) row-major record, hess compact (HK = SYM) or
// diagonal (HK = DIAG).  K == D in {1, 2, 3} with a compact hess follow jhj1/2/3 to the
// letter (including their J H J^T convention, quirk Q16); everything else follows jhjn.
template <typename T, int K, int D, int HK>
struct MatmulOp {
    using RA = Rec<K, D>;
    using RB = Rec<1, (HK == NFM_MAT_SYM ? sym_k(K) : K)>;
    using RC = NoRec;
    using RO = Rec<1, sym_k(D)>;
    using Params = NoParams;
    static constexpr int TILE = pick_tile((RA::C + RB::C + RO::C) * (int)sizeof(T) + 48);
    static __device__ __forceinline__ void apply(const T (&J)[RA::Cs], const T (&H)[RB::Cs], const T (&)[1],
                                                 T (&o)[RO::Cs], const Params &)
    {
#pragma clang fp contract(off)
        if constexpr (K == 1 && D == 1) {
            o[0] = (J[0] * J[0]) * H[0];
        } else if constexpr (K == 2 && D == 2 && HK == NFM_MAT_SYM) {
            const T h00 = H[0], h11 = H[1], h01 = H[2];
            const T j00 = J[0], j01 = J[1], j10 = J[2], j11 = J[3];
            o[0] = ((j00 * j00) * h00 + (j01 * j01) * h11) + ((T(2) * j00) * j01) * h01;
            o[1] = ((j10 * j10) * h00 + (j11 * j11) * h11) + ((T(2) * j10) * j11) * h01;
            o[2] = ((j00 * j10) * h00 + (j01 * j11) * h11) + (j01 * j10 + j00 * j11) * h01;
        } else if constexpr (K == 3 && D == 3 && HK == NFM_MAT_SYM) {
            const T h00 = H[0], h11 = H[1], h22 = H[2], h01 = H[3], h02 = H[4], h12 = H[5];
            auto dg = [&](T a, T b, T c) {
                return (((((h00 * a) * a + ((T(2) * h01) * a) * b) + ((T(2) * h02) * a) * c) + (h11 * b) * b) +
                        ((T(2) * h12) * b) * c) +
                       (h22 * c) * c;
            };
            auto off = [&](T p, T q, T r, T a, T b, T c) {
                return (p * ((h00 * a + h01 * b) + h02 * c) + q * ((h01 * a + h11 * b) + h12 * c)) +
                       r * ((h02 * a + h12 * b) + h22 * c);
            };
            o[0] = dg(J[0], J[1], J[2]);
            o[1] = dg(J[3], J[4], J[5]);
            o[2] = dg(J[6], J[7], J[8]);
            o[3] = off(J[3], J[4], J[5], J[0], J[1], J[2]);
            o[4] = off(J[6], J[7], J[8], J[0], J[1], J[2]);
            o[5] = off(J[6], J[7], J[8], J[3], J[4], J[5]);
        } else {
            // jhjn :600-634, accumulation order preserved
#pragma unroll
            for (int d = 0; d < D; ++d) {
                T acc = T(0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    acc += H[k] * (J[k * D + d] * J[k * D + d]);
                    if constexpr (HK == NFM_MAT_SYM) {
#pragma unroll
                        for (int l = k + 1; l < K; ++l)
                            acc += ((T(2) * H[sym_idx(K, k, l)]) * J[k * D + d]) * J[l * D + d];
                    }
                }
                o[d] = acc;
#pragma unroll
                for (int e = d + 1; e < D; ++e) {
                    T ac = T(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        ac += (H[k] * J[k * D + d]) * J[k * D + e];
                        if constexpr (HK == NFM_MAT_SYM) {
#pragma unroll
                            for (int l = k + 1; l < K; ++l)
                                ac += H[sym_idx(K, k, l)] *
                                      (J[k * D + d] * J[l * D + e] + J[l * D + d] * J[k * D + e]);
                        }
                    }
                    o[sym_idx(D, d, e)] = ac;
                }
            }
        }
    }
};

// ------------------------------------------------------------------- dispatch helpers
#define NFM_CASE_M(Mv, ...) \
    case Mv: {              \
        constexpr int M = Mv; \
        __VA_ARGS__;        \
    } break;

#define NFM_SWITCH_M8(Mexpr, ...)        \
    switch (Mexpr) {                     \
        NFM_CASE_M(1, __VA_ARGS__)       \
        NFM_CASE_M(2, __VA_ARGS__)       \
        NFM_CASE_M(3, __VA_ARGS__)       \
        NFM_CASE_M(4, __VA_ARGS__)       \
        NFM_CASE_M(5, __VA_ARGS__)       \
        NFM_CASE_M(6, __VA_ARGS__)       \
        NFM_CASE_M(7, __VA_ARGS__)       \
        NFM_CASE_M(8, __VA_ARGS__)       \
    default:                             \
        return NFM_ESIZE;                \
    }

template <typename T>
static int sym_solve_t(int M, int kind, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                       const nfm_operand *out, const SolveParams &p, void *stream)
{
    if (M > 8) return big_sym_solve<T>(M, kind, no, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
    switch (kind) {
    case NFM_MAT_SYM:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_SYM>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    case NFM_MAT_DIAG:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_DIAG>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    case NFM_MAT_SCAL:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_SCAL>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    case NFM_MAT_FULL:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_FULL>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    default:
        return NFM_EINVAL;
    }
    return NFM_EINVAL;
}

template <typename T>
static int sym_matvec_t(int M, int kind, int mode, int64_t no, int64_t ni, const nfm_operand *mat,
                        const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out, void *stream)
{
    if (M > 8) return big_sym_matvec<T>(M, kind, mode, no, ni, mat, vec, inp, out, stream);
    MatvecParams p{mode};
    switch (kind) {
    case NFM_MAT_SYM:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_SYM>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    case NFM_MAT_DIAG:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_DIAG>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    case NFM_MAT_SCAL:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_SCAL>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    case NFM_MAT_FULL:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_FULL>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    default:
        return NFM_EINVAL;
    }
    return NFM_EINVAL;
}

template <typename T>
static int sym_invert_t(int M, int diag_only, int64_t no, int64_t ni, const nfm_operand *mat,
                        const nfm_operand *out, void *stream)
{
    if (M > 8) return big_sym_invert<T>(M, diag_only, no, ni, mat, out, stream);
    NoParams p{0};
    if (diag_only) {
        NFM_SWITCH_M8(M, return (rec_launch<T, InvertOp<T, M, true>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    } else {
        NFM_SWITCH_M8(M, return (rec_launch<T, InvertOp<T, M, false>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    }
    return NFM_EINVAL;
}

template <typename T>
static int sym_det_t(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    if (M > 8) return big_sym_det<T>(M, no, ni, mat, out, stream);
    NoParams p{0};
    NFM_SWITCH_M8(M, return (rec_launch<T, DetOp<T, M>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int sym_to_full_t(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out,
                         void *stream)
{
    if (M > 8) return big_sym_to_full<T>(M, no, ni, mat, out, stream);
    NoParams p{0};
    NFM_SWITCH_M8(M, return (rec_launch<T, ToFullOp<T, M>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int sym_outer_t(int M, int64_t no, int64_t ni, const nfm_operand *x, const nfm_operand *out, void *stream)
{
    if (M > 8) return big_sym_outer<T>(M, no, ni, x, out, stream);
    NoParams p{0};
    NFM_SWITCH_M8(M, return (rec_launch<T, OuterOp<T, M>>(x, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int sym_outer2_t(int M, int neg, int64_t no, int64_t ni, const nfm_operand *x, const nfm_operand *y,
                        const nfm_operand *out, void *stream)
{
    if (M > 8) return big_sym_outer2<T>(M, neg, no, ni, x, y, out, stream);
    Outer2Params p{neg};
    NFM_SWITCH_M8(M, return (rec_launch<T, Outer2Op<T, M>>(x, y, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T, int HK>
static int sym_matmul_t(int K, int D, int64_t no, int64_t ni, const nfm_operand *jac, const nfm_operand *hess,
                        const nfm_operand *out, void *stream)
{
    NoParams p{0};
#define NFM_MM(Kv, Dv)                                                                              \
    if (K == Kv && D == Dv)                                                                         \
        return (rec_launch<T, MatmulOp<T, Kv, Dv, HK>>(jac, hess, nullptr, out, no, ni, p, stream));
    NFM_MM(1, 1) NFM_MM(1, 2) NFM_MM(1, 3) NFM_MM(1, 4)
    NFM_MM(2, 1) NFM_MM(2, 2) NFM_MM(2, 3) NFM_MM(2, 4)
    NFM_MM(3, 1) NFM_MM(3, 2) NFM_MM(3, 3) NFM_MM(3, 4)
    NFM_MM(4, 1) NFM_MM(4, 2) NFM_MM(4, 3) NFM_MM(4, 4)
#undef NFM_MM
    return big_sym_matmul<T>(K, D, HK, no, ni, jac, hess, out, stream);
}

} // namespace nfm

using namespace nfm;

extern "C" {

int nfm_sym_solve(int dtype, int M, int mat_kind, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                  const nfm_operand *vec, const nfm_operand *out, const double *eps, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    if (mat_kind < 0 || mat_kind > 3) return NFM_EINVAL;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(vec, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    SolveParams p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? eps[i] : 0.0;
    return dtype == NFM_F32 ? sym_solve_t<float>(M, mat_kind, n_outer, n_inner, mat, vec, out, p, stream)
                            : sym_solve_t<double>(M, mat_kind, n_outer, n_inner, mat, vec, out, p, stream);
}

int nfm_sym_matvec(int dtype, int M, int mat_kind, int mode, int64_t n_outer, int64_t n_inner,
                   const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out,
                   void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    if (mat_kind < 0 || mat_kind > 3 || mode < -1 || mode > 1) return NFM_EINVAL;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(vec, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    if (mode != 0 && (rc = check_operand(inp, dtype, nonempty))) return rc;
    if (mode == 0) inp = nullptr;
    return dtype == NFM_F32 ? sym_matvec_t<float>(M, mat_kind, mode, n_outer, n_inner, mat, vec, inp, out, stream)
                            : sym_matvec_t<double>(M, mat_kind, mode, n_outer, n_inner, mat, vec, inp, out, stream);
}

int nfm_sym_invert(int dtype, int M, int diag_only, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                   const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_invert_t<float>(M, diag_only, n_outer, n_inner, mat, out, stream)
                            : sym_invert_t<double>(M, diag_only, n_outer, n_inner, mat, out, stream);
}

int nfm_sym_det(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *mat, const nfm_operand *out,
                void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_det_t<float>(M, n_outer, n_inner, mat, out, stream)
                            : sym_det_t<double>(M, n_outer, n_inner, mat, out, stream);
}

int nfm_sym_to_full(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                    const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_to_full_t<float>(M, n_outer, n_inner, mat, out, stream)
                            : sym_to_full_t<double>(M, n_outer, n_inner, mat, out, stream);
}

int nfm_sym_outer(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *x, const nfm_operand *out,
                  void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(x, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_outer_t<float>(M, n_outer, n_inner, x, out, stream)
                            : sym_outer_t<double>(M, n_outer, n_inner, x, out, stream);
}

int nfm_sym_outer2(int dtype, int M, int neg, int64_t n_outer, int64_t n_inner, const nfm_operand *x,
                   const nfm_operand *y, const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(x, dtype, nonempty))) return rc;
    if ((rc = check_operand(y, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_outer2_t<float>(M, neg ? 1 : 0, n_outer, n_inner, x, y, out, stream)
                            : sym_outer2_t<double>(M, neg ? 1 : 0, n_outer, n_inner, x, y, out, stream);
}

int nfm_sym_matmul(int dtype, int K, int D, int hess_kind, int64_t n_outer, int64_t n_inner,
                   const nfm_operand *jac, const nfm_operand *hess, const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (K < 1 || K > NFM_MAX_DIM || D < 1 || D > NFM_MAX_DIM) return NFM_ESIZE;
    if (hess_kind != NFM_MAT_SYM && hess_kind != NFM_MAT_DIAG) return NFM_EINVAL;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(jac, dtype, nonempty))) return rc;
    if ((rc = check_operand(hess, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    if (dtype == NFM_F32)
        return hess_kind == NFM_MAT_SYM
                   ? sym_matmul_t<float, NFM_MAT_SYM>(K, D, n_outer, n_inner, jac, hess, out, stream)
                   : sym_matmul_t<float, NFM_MAT_DIAG>(K, D, n_outer, n_inner, jac, hess, out, stream);
    return hess_kind == NFM_MAT_SYM
               ? sym_matmul_t<double, NFM_MAT_SYM>(K, D, n_outer, n_inner, jac, hess, out, stream)
               : sym_matmul_t<double, NFM_MAT_DIAG>(K, D, n_outer, n_inner, jac, hess, out, stream);
}

} // extern "C"

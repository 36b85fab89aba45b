// nfm_sym_ops.hpp -- per-lane operations on compact symmetric matrices (the `Op` structs
// plugged into rec_kernel).  Shared by nfm_sym.hip (orders 1..8) and nfm_large.hip
// (orders 9..16 in registers).
#pragma once
#include "nfm_record_kernel.hpp"
#include "nfm_smallmat.hpp"

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
    // contiguous 4x4 / 6x6 fp32 systems (configurations C2 / C5): 512-lane tiles, +2-4 % in same-box
    // A/B runs at 1e8 systems (see KindTile); the other orders measured flat within run-to-run noise
    static constexpr int kAosTile = ((M == 4 || M == 6) && KIND == NFM_MAT_SYM && sizeof(T) == 4) ? 512 : TILE;
    static constexpr bool kNoTile = KIND == NFM_MAT_SYM && large_no_tile(sizeof(T) == 8, M, LN_SOLVE);
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
    static constexpr bool kNoTile = large_no_tile(sizeof(T) == 8, M, LN_DET);
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

} // namespace nfm

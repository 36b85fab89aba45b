// nfm_sym.hip -- compact-symmetric entry points (sym_solve / matvec / invert / det /
// to_full / outer / matmul) for orders 1..8 in registers; orders 9..16 are in
// nfm_big.hip.  See include/nfm_hip.h for the ABI and the reference lines replaced.
#include "nfm_sym_ops.hpp"
#include "nfm_big.hpp"
#include "nfm_large.hpp"
#include "nfm_rowwave.hpp"

namespace nfm {

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

// One matrix (per outer slab) solved against many vectors (M <= 8).  Closed forms (M <= 4): the cofactors and
// the determinant are derived once per lane and applied to V right-hand sides -- the same operations
// in the same order as SolveOp (sym_solve_prepare / sym_solve_apply are the two halves of
// sym_solve_closed), so the results are bit for bit the per-record kernel's, at 16 multiply-adds and
// 4 divisions per system instead of ~250 instructions: the kernel becomes a stream over the vectors.
template <typename T, int M, int V>
__global__ __launch_bounds__(256) void sym_solve_bcast_kernel(Opnd mat, Opnd vec, Opnd out, int64_t n_inner,
                                                              SolveParams p)
{
    constexpr int K = sym_k(M);
    using RV = Rec<1, M>;
    const int64_t o = blockIdx.y;
    const T *pm = reinterpret_cast<const T *>(mat.ptr) + o * mat.so;
    T a[K];
#pragma unroll
    for (int c = 0; c < K; ++c) a[c] = pm[c * mat.sc];
    if (p.has_eps) {
#pragma unroll
        for (int i = 0; i < M; ++i) a[i] += (T)p.eps[i];
    }
    const int64_t base = (int64_t)blockIdx.x * (256 * V) + threadIdx.x;
    T v[V][M];
#pragma unroll
    for (int q = 0; q < V; ++q) rec_direct_load<T, RV>(vec, vec.tiled, o, base + q * 256, base + q * 256 < n_inner, v[q]);
    if constexpr (M <= 4) {
        T co[SymCofLen<M>::value], det;
        sym_solve_prepare<T, M>(a, co, det);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            T r[M];
            sym_solve_apply<T, M>(co, det, v[q], r);
            rec_direct_store<T, RV>(out, out.tiled, o, base + q * 256, base + q * 256 < n_inner, r);
        }
    } else {
        // orders 5..8: the elimination with partial pivoting of SolveOp, once, with the V vectors as the columns
        // of one right-hand side (ge_solve treats every column independently: the same bits per system)
        T f[M][M], b[M][V];
        sym_expand<T, M>(a, f);
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int q = 0; q < V; ++q) b[i][q] = v[q][i];
        ge_solve<T, M, V>(f, b);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            T r[M];
#pragma unroll
            for (int i = 0; i < M; ++i) r[i] = b[i][q];
            rec_direct_store<T, RV>(out, out.tiled, o, base + q * 256, base + q * 256 < n_inner, r);
        }
    }
}

template <typename T, int M>
static int sym_solve_bcast(int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                           const nfm_operand *out, const SolveParams &p, void *stream)
{
    constexpr int V = 4;
    const int64_t nblk = (ni + 256 * V - 1) / (256 * V);
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    auto mode = [](const nfm_operand *op) {
        return packed_ok(op, M, 1, M, sizeof(T)) ? (int)MODE_PACKED : (int)MODE_STRIDED;
    };
    hipLaunchKernelGGL((sym_solve_bcast_kernel<T, M, V>), dim3((unsigned)nblk, (unsigned)no, 1), dim3(256, 1, 1), 0,
                       static_cast<hipStream_t>(stream), make_opnd(mat, MODE_STRIDED), make_opnd(vec, mode(vec)),
                       make_opnd(out, mode(out)), ni, p);
    return launch_status();
}

template <typename T>
static int sym_solve_t(int M, int kind, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                       const nfm_operand *out, const SolveParams &p, void *stream)
{
    // a matrix that is the same along the inner batch level (stride 0: one Hessian, many gradients)
    if (kind == NFM_MAT_SYM && M >= 2 && M <= 8 && mat->stride_inner == 0 && ni >= 1024) {
        switch (M) {
        case 2: return sym_solve_bcast<T, 2>(no, ni, mat, vec, out, p, stream);
        case 3: return sym_solve_bcast<T, 3>(no, ni, mat, vec, out, p, stream);
        case 4: return sym_solve_bcast<T, 4>(no, ni, mat, vec, out, p, stream);
        case 5: return sym_solve_bcast<T, 5>(no, ni, mat, vec, out, p, stream);
        case 6: return sym_solve_bcast<T, 6>(no, ni, mat, vec, out, p, stream);
        case 7: return sym_solve_bcast<T, 7>(no, ni, mat, vec, out, p, stream);
        default: return sym_solve_bcast<T, 8>(no, ni, mat, vec, out, p, stream);
        }
    }
    if (M > 8) {
        if (kind == NFM_MAT_SYM && no == 1) { // contiguous operands: registers; else LDS-resident
            if (rowwave_first<T>(M, RWW_SOLVE)) { // one matrix per 16 lanes (nfm_rowwave.hip)
                const int rc = RowWave<T>::sym_solve(M, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
                if (rc != NFM_EFALLBACK) return rc;
            }
            const int rc = Large<T>::sym_solve(M, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_solve<T>(M, kind, no, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
    }
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
    if (M > 8) {
        if (kind == NFM_MAT_SYM && no == 1) {
            const int rc = Large<T>::sym_matvec(M, mode, ni, mat, vec, inp, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_matvec<T>(M, kind, mode, no, ni, mat, vec, inp, out, stream);
    }
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
    if (M > 8) {
        if (no == 1 && rowwave_first<T>(M, diag_only ? RWW_INVDIAG_SYM : RWW_INV_SYM)) {
            const int rc = RowWave<T>::sym_invert(M, diag_only, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (!diag_only && no == 1) {
            const int rc = Large<T>::sym_invert(M, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_invert<T>(M, diag_only, no, ni, mat, out, stream);
    }
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
    if (M > 8) {
        if (no == 1 && rowwave_first<T>(M, RWW_DET_SYM)) {
            const int rc = RowWave<T>::sym_det(M, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1) {
            const int rc = Large<T>::sym_det(M, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_det<T>(M, no, ni, mat, out, stream);
    }
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

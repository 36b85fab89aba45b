// nfm_batched.hip -- general small matrices: batchinv / batchdet / batchmatvec
// (reference `_impl/batched.py`).  One matrix per lane; N <= 3 use the reference's
// adjugate closed forms, 4 <= N <= 8 Gauss-Jordan / LU with partial pivoting in
// registers, N > 8 the LDS-resident kernels of nfm_big.hpp.
#include "nfm_batched_ops.hpp"
#include "nfm_big.hpp"
#include "nfm_large.hpp"
#include "nfm_rowwave.hpp"
#include "nfm_spd.hpp"

namespace nfm {

#define NFM_CASE_N(Nv, ...)   \
    case Nv: {                \
        constexpr int N = Nv; \
        __VA_ARGS__;          \
    } break;
#define NFM_SWITCH_N8(Nexpr, ...)  \
    switch (Nexpr) {               \
        NFM_CASE_N(1, __VA_ARGS__) \
        NFM_CASE_N(2, __VA_ARGS__) \
        NFM_CASE_N(3, __VA_ARGS__) \
        NFM_CASE_N(4, __VA_ARGS__) \
        NFM_CASE_N(5, __VA_ARGS__) \
        NFM_CASE_N(6, __VA_ARGS__) \
        NFM_CASE_N(7, __VA_ARGS__) \
        NFM_CASE_N(8, __VA_ARGS__) \
    default:                       \
        return NFM_ESIZE;          \
    }

template <typename T>
static int batch_inv_t(int N, int flags, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *out,
                       void *stream)
{
    if (N > 8) {
        if (no == 1) { // diagonal pivots first, the pivoted elimination for the wavefronts that need it (nfm_spd.hip)
            const int rc = Spd<T>::batch_inv(N, ni, a, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1 && rowwave_first<T>(N, RWW_INV_GEN)) { // one matrix per 16 lanes (nfm_rowwave.hip)
            const int rc = RowWave<T>::batch_inv(N, ni, a, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1) {
            const int rc = Large<T>::batch_inv(N, ni, a, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_batch_inv<T>(N, no, ni, a, out, stream);
    }
    InvParams p{(flags & NFM_FLAG_TS_PERTURB) ? 1 : 0};
    NFM_SWITCH_N8(N, return (rec_launch<T, BatchInvOp<T, N>>(a, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int batch_det_t(int N, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (N > 8) {
        if (no == 1) { // diagonal pivots first (nfm_spd.hip)
            const int rc = Spd<T>::batch_det(N, ni, a, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1 && rowwave_first<T>(N, RWW_DET_GEN)) {
            const int rc = RowWave<T>::batch_det(N, ni, a, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1) {
            const int rc = Large<T>::batch_det(N, ni, a, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_batch_det<T>(N, no, ni, a, out, stream);
    }
    NoParamsB p{0};
    NFM_SWITCH_N8(N, return (rec_launch<T, BatchDetOp<T, N>>(a, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int batch_matvec_t(int R, int C, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *v,
                          const nfm_operand *out, void *stream)
{
    NoParamsB p{0};
#define NFM_MV(Rv, Cv) \
    if (R == Rv && C == Cv) return (rec_launch<T, BatchMatvecOp<T, Rv, Cv>>(a, v, nullptr, out, no, ni, p, stream));
    NFM_MV(1, 1) NFM_MV(2, 2) NFM_MV(3, 3) NFM_MV(4, 4) NFM_MV(5, 5) NFM_MV(6, 6) NFM_MV(7, 7) NFM_MV(8, 8)
    NFM_MV(2, 3) NFM_MV(3, 2) NFM_MV(3, 4) NFM_MV(4, 3) NFM_MV(4, 5) NFM_MV(5, 4)
#undef NFM_MV
    return big_batch_matvec<T>(R, C, no, ni, a, v, out, stream);
}

} // namespace nfm

using namespace nfm;

extern "C" {

int nfm_batch_inv(int dtype, int N, int flags, int64_t n_outer, int64_t n_inner, const nfm_operand *a,
                  const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (N < 1 || N > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? batch_inv_t<float>(N, flags, n_outer, n_inner, a, out, stream)
                            : batch_inv_t<double>(N, flags, n_outer, n_inner, a, out, stream);
}

int nfm_batch_det(int dtype, int N, int64_t n_outer, int64_t n_inner, const nfm_operand *a, const nfm_operand *out,
                  void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (N < 1 || N > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? batch_det_t<float>(N, n_outer, n_inner, a, out, stream)
                            : batch_det_t<double>(N, n_outer, n_inner, a, out, stream);
}

int nfm_batch_matvec(int dtype, int rows, int cols, int64_t n_outer, int64_t n_inner, const nfm_operand *a,
                     const nfm_operand *v, const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (rows < 1 || rows > NFM_MAX_DIM || cols < 1 || cols > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(v, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? batch_matvec_t<float>(rows, cols, n_outer, n_inner, a, v, out, stream)
                            : batch_matvec_t<double>(rows, cols, n_outer, n_inner, a, v, out, stream);
}

} // extern "C"

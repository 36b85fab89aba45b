// nfm_rowwave.hpp -- front end of the one-matrix-per-16-lanes kernels (nfm_rowwave.hip) for
// orders 9..16 on contiguous batch-major operands.  Every function answers NFM_EFALLBACK_ when
// the order or the layout is not covered; the caller then takes the lane-per-matrix kernels of
// nfm_large.hip / the LDS-resident ones of nfm_big.hpp.
#pragma once
#include "nfm_common.hpp"

namespace nfm {

constexpr int NFM_EFALLBACK_RW = -100; // == NFM_EFALLBACK of nfm_record_kernel.hpp

template <typename T>
struct RowWave {
    static int sym_solve(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                         const double *eps, void *stream);
    static int sym_invert(int M, int diag_only, int64_t ni, const nfm_operand *mat, const nfm_operand *out,
                          void *stream);
    static int sym_det(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream);
    static int batch_inv(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);
    static int batch_det(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);
};

// Which (dtype, order) pairs go to the row-wave kernels first (measured, profiles/r02/rowwave_table.md):
// the lane-per-matrix kernels win while a matrix fits the register file with room to spare.
// NFM_ROWWAVE_MIN_F64 / _F32 (environment, read once) override the thresholds for experiments.
// what: which op asks (the cross-over order differs per op and dtype); the diagonal of the compact
// inverse has no lane-per-matrix kernel beyond order 8, so the row-wave kernel takes every order 9..16
enum { RWW_SOLVE = 0, RWW_INV_SYM, RWW_INVDIAG_SYM, RWW_DET_SYM, RWW_INV_GEN, RWW_DET_GEN };
int rowwave_min_order(int dtype_is_f64, int what);
template <typename T>
inline bool rowwave_first(int N, int what)
{
    return N >= rowwave_min_order(sizeof(T) == 8, what);
}

} // namespace nfm

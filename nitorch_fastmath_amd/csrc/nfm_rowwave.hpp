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

// Which cases the row-wave kernels serve, and in which form (measured on MI355X,
// profiles/r03/rowwave_table.md -- re-measured after the pivot step lost its branch: the compact inverses now take
// the pivot row through the LDS slot (float32 12..16: +10-20 %, float64 10..12: up to +24 %) -- and r02/rowwave_table.md: every op x order 9..16 x dtype x {lane-per-matrix, 1 / 2 / 4 rows
// per lane, pivot row through ds_bpermute or through an LDS slot}).  rows == 0: the lane-per-matrix
// register kernel of nfm_large.hip is faster and keeps the case.  Cases listed here are NOT
// instantiated in nfm_large.hip any more (they were the kernels that needed scratch memory:
// float64 solve 13..16, inverses 12..16, determinants 14..16; float32 batchdet 16).
enum { RWW_SOLVE = 0, RWW_INV_SYM, RWW_INVDIAG_SYM, RWW_DET_SYM, RWW_INV_GEN, RWW_DET_GEN };
struct RwChoice {
    int rows;  // rows per lane (16 / rows lanes per matrix); 0 = not a row-wave case
    bool lds;  // pivot row through an LDS slot instead of ds_bpermute
};
constexpr RwChoice rowwave_choice(bool f64, int N, int what)
{
    if (N < 9 || N > 16) return {0, false};
    if (what == RWW_INVDIAG_SYM) return {(f64 && N >= 13) ? 2 : 4, false}; // no lane-per-matrix form exists
    if (f64) {
        switch (what) {
        case RWW_SOLVE: return N >= 13 ? RwChoice{2, false} : N == 12 ? RwChoice{4, false} : RwChoice{0, false};
        case RWW_INV_SYM: return N >= 13 ? RwChoice{2, true} : N >= 10 ? RwChoice{4, true} : RwChoice{0, false};
        case RWW_DET_SYM: return N >= 14 ? RwChoice{2, false} : RwChoice{0, false};
        case RWW_INV_GEN: return N >= 15 ? RwChoice{1, true} : N >= 11 ? RwChoice{2, false} : RwChoice{0, false};
        case RWW_DET_GEN: return N >= 13 ? RwChoice{1, true} : RwChoice{0, false};
        default: return {0, false};
        }
    }
    switch (what) {
    case RWW_INV_SYM: return N >= 16 ? RwChoice{2, true} : N >= 12 ? RwChoice{4, true} : RwChoice{0, false};
    case RWW_INV_GEN: return N >= 14 ? RwChoice{2, true} : N == 12 ? RwChoice{4, true} : RwChoice{0, false};
    case RWW_DET_GEN: return N >= 16 ? RwChoice{2, false} : RwChoice{0, false};
    default: return {0, false};
    }
}
// run-time form for the dispatchers (nfm_sym.hip, nfm_batched.hip).  NFM_ROWWAVE_MIN_F64 / _F32
// (environment, read once per process) force every order >= the value onto the row-wave kernels:
// scripts/bench_rowwave.py uses it to time the forms side by side.
bool rowwave_forced(bool f64, int N);
template <typename T>
inline bool rowwave_first(int N, int what)
{
    return rowwave_choice(sizeof(T) == 8, N, what).rows != 0 || rowwave_forced(sizeof(T) == 8, N);
}

} // namespace nfm

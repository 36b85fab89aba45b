// nfm_spd.hpp -- front end of the no-exchange-first kernels (nfm_spd.hip): orders 9..16 of sym_solve / sym_invert /
// sym_det on contiguous compact records (positive definite first) and of batchinv / batchdet on contiguous
// row-major matrices (diagonal pivots first).  NFM_EFALLBACK_RW when the layout is not covered.
#pragma once
#include "nfm_common.hpp"
#include "nfm_rowwave.hpp"

namespace nfm {

enum { SP_SOLVE = 0, SP_INV, SP_INVDIAG, SP_DET, SP_GINV, SP_GDET }; // G*: general (full, row-major) matrices

// one object per (dtype, pair of orders): part = dtype * 4 + q holds orders 9 + 2 q and 10 + 2 q
#define NFM_SPD_DECL(S, Q)                                                                                                      \
    int spd_call_##S##_q##Q(int op, int M, int64_t n, const void *a, const void *b, void *o, const double *eps, void *stream); \
    int spd_call_strided_##S##_q##Q(int op, int M, int64_t no, int64_t n, const nfm_operand *a, const nfm_operand *b,          \
                                    const nfm_operand *o, const double *eps, void *stream);                                   \
    int spd_matvec_strided_##S##_q##Q(int M, int mode, int64_t no, int64_t n, const nfm_operand *a, const nfm_operand *b,      \
                                      const nfm_operand *c, const nfm_operand *o, void *stream);                              \
    int spd_matvec_tiled_##S##_q##Q(int M, int mode, int64_t n, const void *a, const void *b, const void *c, void *o,          \
                                    void *stream);
NFM_SPD_DECL(f32, 0) NFM_SPD_DECL(f32, 1) NFM_SPD_DECL(f32, 2) NFM_SPD_DECL(f32, 3)
NFM_SPD_DECL(f64, 0) NFM_SPD_DECL(f64, 1) NFM_SPD_DECL(f64, 2) NFM_SPD_DECL(f64, 3)
#undef NFM_SPD_DECL

template <typename T>
struct Spd {
    static int sym_solve(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                         const double *eps, void *stream);
    static int sym_invert(int M, int diag_only, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream);
    static int sym_det(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream);
    // the same for ANY strides of the compact operands (channel-first fields, padded / interleaved records, one vector
    // for every matrix) and a second batch level: every lane addresses its own record, element by element
    static int sym_solve_strided(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                                 const nfm_operand *out, const double *eps, void *stream);
    static int sym_invert_strided(int M, int diag_only, int64_t no, int64_t ni, const nfm_operand *mat,
                                  const nfm_operand *out, void *stream);
    static int sym_det_strided(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream);
    // y = [inp +/-] mat * vec at any strides (no factorisation, nothing to fall back to: it lives here for the strided
    // record access it shares with the kernels above)
    // contiguous operands, float64 orders 15, 16 (the lane-by-lane kernel of nfm_large.hip ran at 0.49 of the roofline
    // at 16x16): the records through LDS images, like the kernels above
    static int sym_matvec(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                          const nfm_operand *inp, const nfm_operand *out, void *stream);
    static int sym_matvec_strided(int M, int mode, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                                  const nfm_operand *inp, const nfm_operand *out, void *stream);
    static int batch_inv(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);
    static int batch_det(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);
};

} // namespace nfm

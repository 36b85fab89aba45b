// nfm_large.hpp -- orders 9..16 held in registers (the same Ops as orders 1..8, compiled
// only in their contiguous-operand FAST form).  Each function answers NFM_EFALLBACK when
// the order / dtype / layout is not covered; the caller then uses the LDS-resident kernels
// of nfm_big.hpp.  Coverage is set by the 512-register budget of a lane:
//   sym_solve, sym_det, sym_matvec, sym_invert (full), batch_inv, batch_det: orders 9..16, f32 and f64
#pragma once
#include "nfm_common.hpp"

namespace nfm {

constexpr int NFM_EFALLBACK_ = -100; // == NFM_EFALLBACK of nfm_record_kernel.hpp


#define NFM_LARGE_DECL(T)                                                                                          \
    int large_sym_solve_##T(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,                     \
                            const nfm_operand *out, const double *eps, void *stream);                             \
    int large_sym_det_##T(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream);        \
    int large_sym_invert_##T(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream);     \
    int large_sym_matvec_##T(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,          \
                             const nfm_operand *inp, const nfm_operand *out, void *stream);                       \
    int large_batch_inv_##T(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);        \
    int large_batch_det_##T(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);
NFM_LARGE_DECL(f32)
NFM_LARGE_DECL(f64)
#undef NFM_LARGE_DECL

// type-dispatched front ends used by nfm_sym.hip / nfm_batched.hip
template <typename T>
struct Large;
template <>
struct Large<float> {
    static constexpr auto sym_solve = large_sym_solve_f32;
    static constexpr auto sym_det = large_sym_det_f32;
    static constexpr auto sym_invert = large_sym_invert_f32;
    static constexpr auto sym_matvec = large_sym_matvec_f32;
    static constexpr auto batch_inv = large_batch_inv_f32;
    static constexpr auto batch_det = large_batch_det_f32;
};
template <>
struct Large<double> {
    static constexpr auto sym_solve = large_sym_solve_f64;
    static constexpr auto sym_det = large_sym_det_f64;
    static constexpr auto sym_invert = large_sym_invert_f64;
    static constexpr auto sym_matvec = large_sym_matvec_f64;
    static constexpr auto batch_inv = large_batch_inv_f64;
    static constexpr auto batch_det = large_batch_det_f64;
};

} // namespace nfm

// nfm_large.hpp -- orders 9..16 held in registers (the same Ops as orders 1..8, compiled
// only in their contiguous-operand FAST form).  Each function answers NFM_EFALLBACK when
// the order / dtype / layout is not covered; the caller then uses the LDS-resident kernels
// of nfm_big.hpp.  Coverage is set by the 512-register budget of a lane:
//   sym_solve, sym_det, sym_matvec, sym_invert (full), batch_inv, batch_det: orders 9..16, f32 and f64
#pragma once
#include "nfm_common.hpp"

namespace nfm {

constexpr int NFM_EFALLBACK_ = -100; // == NFM_EFALLBACK of nfm_record_kernel.hpp


#define NFM_LARGE_DECL(T, Q)                                                                                        \
    int large_sym_solve_##T##_q##Q(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,                 \
                                   const nfm_operand *out, const double *eps, void *stream);                         \
    int large_sym_det_##T##_q##Q(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream);    \
    int large_sym_invert_##T##_q##Q(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream); \
    int large_sym_matvec_##T##_q##Q(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,      \
                                    const nfm_operand *inp, const nfm_operand *out, void *stream);                   \
    int large_batch_inv_##T##_q##Q(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);    \
    int large_batch_det_##T##_q##Q(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream);
NFM_LARGE_DECL(f32, 0) NFM_LARGE_DECL(f32, 1) NFM_LARGE_DECL(f32, 2) NFM_LARGE_DECL(f32, 3)
NFM_LARGE_DECL(f64, 0) NFM_LARGE_DECL(f64, 1) NFM_LARGE_DECL(f64, 2) NFM_LARGE_DECL(f64, 3)
#undef NFM_LARGE_DECL

// type-dispatched front ends used by nfm_sym.hip / nfm_batched.hip: object "q" holds orders
// 9+2q and 10+2q
template <typename T>
struct Large;
#define NFM_LARGE_FRONT(T, S)                                                                                                                                        \
    template <>                                                                                                                                                      \
    struct Large<T> {                                                                                                                                                \
        static int sym_solve(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out, const double *eps, void *st)                 \
        {                                                                                                                                                            \
            switch ((M - 9) >> 1) {                                                                                                                                  \
            case 0: return large_sym_solve_##S##_q0(M, ni, mat, vec, out, eps, st);                                                                                  \
            case 1: return large_sym_solve_##S##_q1(M, ni, mat, vec, out, eps, st);                                                                                  \
            case 2: return large_sym_solve_##S##_q2(M, ni, mat, vec, out, eps, st);                                                                                  \
            case 3: return large_sym_solve_##S##_q3(M, ni, mat, vec, out, eps, st);                                                                                  \
            default: return NFM_EFALLBACK;                                                                                                                           \
            }                                                                                                                                                        \
        }                                                                                                                                                            \
        static int sym_det(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *st)                                                              \
        {                                                                                                                                                            \
            switch ((M - 9) >> 1) {                                                                                                                                  \
            case 0: return large_sym_det_##S##_q0(M, ni, mat, out, st);                                                                                              \
            case 1: return large_sym_det_##S##_q1(M, ni, mat, out, st);                                                                                              \
            case 2: return large_sym_det_##S##_q2(M, ni, mat, out, st);                                                                                              \
            case 3: return large_sym_det_##S##_q3(M, ni, mat, out, st);                                                                                              \
            default: return NFM_EFALLBACK;                                                                                                                           \
            }                                                                                                                                                        \
        }                                                                                                                                                            \
        static int sym_invert(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *st)                                                           \
        {                                                                                                                                                            \
            switch ((M - 9) >> 1) {                                                                                                                                  \
            case 0: return large_sym_invert_##S##_q0(M, ni, mat, out, st);                                                                                           \
            case 1: return large_sym_invert_##S##_q1(M, ni, mat, out, st);                                                                                           \
            case 2: return large_sym_invert_##S##_q2(M, ni, mat, out, st);                                                                                           \
            case 3: return large_sym_invert_##S##_q3(M, ni, mat, out, st);                                                                                           \
            default: return NFM_EFALLBACK;                                                                                                                           \
            }                                                                                                                                                        \
        }                                                                                                                                                            \
        static int sym_matvec(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out, void *st) \
        {                                                                                                                                                            \
            switch ((M - 9) >> 1) {                                                                                                                                  \
            case 0: return large_sym_matvec_##S##_q0(M, mode, ni, mat, vec, inp, out, st);                                                                           \
            case 1: return large_sym_matvec_##S##_q1(M, mode, ni, mat, vec, inp, out, st);                                                                           \
            case 2: return large_sym_matvec_##S##_q2(M, mode, ni, mat, vec, inp, out, st);                                                                           \
            case 3: return large_sym_matvec_##S##_q3(M, mode, ni, mat, vec, inp, out, st);                                                                           \
            default: return NFM_EFALLBACK;                                                                                                                           \
            }                                                                                                                                                        \
        }                                                                                                                                                            \
        static int batch_inv(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *st)                                                              \
        {                                                                                                                                                            \
            switch ((N - 9) >> 1) {                                                                                                                                  \
            case 0: return large_batch_inv_##S##_q0(N, ni, a, out, st);                                                                                              \
            case 1: return large_batch_inv_##S##_q1(N, ni, a, out, st);                                                                                              \
            case 2: return large_batch_inv_##S##_q2(N, ni, a, out, st);                                                                                              \
            case 3: return large_batch_inv_##S##_q3(N, ni, a, out, st);                                                                                              \
            default: return NFM_EFALLBACK;                                                                                                                           \
            }                                                                                                                                                        \
        }                                                                                                                                                            \
        static int batch_det(int N, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *st)                                                              \
        {                                                                                                                                                            \
            switch ((N - 9) >> 1) {                                                                                                                                  \
            case 0: return large_batch_det_##S##_q0(N, ni, a, out, st);                                                                                              \
            case 1: return large_batch_det_##S##_q1(N, ni, a, out, st);                                                                                              \
            case 2: return large_batch_det_##S##_q2(N, ni, a, out, st);                                                                                              \
            case 3: return large_batch_det_##S##_q3(N, ni, a, out, st);                                                                                              \
            default: return NFM_EFALLBACK;                                                                                                                           \
            }                                                                                                                                                        \
        }                                                                                                                                                            \
    };
NFM_LARGE_FRONT(float, f32)
NFM_LARGE_FRONT(double, f64)
#undef NFM_LARGE_FRONT

} // namespace nfm

// nfm_large.hip -- register-resident kernels for orders 9..16 (see nfm_large.hpp).
// Compiled eight times (-DNFM_LARGE_PART=0..7), one object per (function group, dtype), so
// that the big fully-unrolled eliminations build in parallel.
#include "nfm_sym_ops.hpp"
#include "nfm_batched_ops.hpp"
#include "nfm_large.hpp"

#ifndef NFM_LARGE_PART
#error "compile with -DNFM_LARGE_PART=0..7"
#endif

namespace nfm {

#define NFM_LCASE(Nv, ...)    \
    case Nv: {                \
        constexpr int N = Nv; \
        __VA_ARGS__;          \
    } break;
#define NFM_LC4(a, b, c, d, ...) NFM_LCASE(a, __VA_ARGS__) NFM_LCASE(b, __VA_ARGS__) NFM_LCASE(c, __VA_ARGS__) NFM_LCASE(d, __VA_ARGS__)
#define NFM_LSWITCH_9_12(Nexpr, ...) \
    switch (Nexpr) { NFM_LC4(9, 10, 11, 12, __VA_ARGS__) default: break; }
#define NFM_LSWITCH_9_13(Nexpr, ...) \
    switch (Nexpr) { NFM_LC4(9, 10, 11, 12, __VA_ARGS__) NFM_LCASE(13, __VA_ARGS__) default: break; }
#define NFM_LSWITCH_9_14(Nexpr, ...) \
    switch (Nexpr) { NFM_LC4(9, 10, 11, 12, __VA_ARGS__) NFM_LCASE(13, __VA_ARGS__) NFM_LCASE(14, __VA_ARGS__) default: break; }
#define NFM_LSWITCH_14_16(Nexpr, ...) \
    switch (Nexpr) { NFM_LCASE(14, __VA_ARGS__) NFM_LCASE(15, __VA_ARGS__) NFM_LCASE(16, __VA_ARGS__) default: break; }
#define NFM_LSWITCH16(Nexpr, ...) \
    switch (Nexpr) { NFM_LC4(9, 10, 11, 12, __VA_ARGS__) NFM_LC4(13, 14, 15, 16, __VA_ARGS__) default: break; }

// Every op covers orders 9..16 in both dtypes.  The pivoting code uses OPAQUE selects here
// (Sel<true>, nfm_smallmat.hpp): no data-dependent control flow is left, so whatever spill
// code the big eliminations need runs under a full EXEC mask, and most kernels need none
// (-Rpass-analysis=kernel-resource-usage: every f32 kernel and f64 up to 13 are spill-free).
//   inverse (compact symmetric and general): LU + column-by-column unit solves (InvStreamOp)
//   solve / determinants: Gaussian elimination with partial pivoting (SolveOp, DetOp, BatchDetOp)
// tests/test_gpu_large_orders.py checks every order of every op against the CPU restatement
// on thousands of matrices, twice.

#if NFM_LARGE_PART == 0 || NFM_LARGE_PART == 1
#if NFM_LARGE_PART == 0
using TS = float;
#else
using TS = double;
#endif
static int large_sym_solve_impl(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                                const nfm_operand *out, const double *eps, void *stream)
{
    SolveParams p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? eps[i] : 0.0;
    NFM_LSWITCH16(M, return (rec_launch<TS, SolveOp<TS, N, NFM_MAT_SYM>, true>(mat, vec, nullptr, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
static int large_sym_det_impl(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    NoParams p{0};
    NFM_LSWITCH16(M, return (rec_launch<TS, DetOp<TS, N>, true>(mat, nullptr, nullptr, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
#if NFM_LARGE_PART == 0
int large_sym_solve_f32(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                        const double *eps, void *stream) { return large_sym_solve_impl(M, ni, mat, vec, out, eps, stream); }
int large_sym_det_f32(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream) { return large_sym_det_impl(M, ni, mat, out, stream); }
#else
int large_sym_solve_f64(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                        const double *eps, void *stream) { return large_sym_solve_impl(M, ni, mat, vec, out, eps, stream); }
int large_sym_det_f64(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream) { return large_sym_det_impl(M, ni, mat, out, stream); }
#endif
#endif

#if NFM_LARGE_PART == 2 || NFM_LARGE_PART == 3
#if NFM_LARGE_PART == 2
using TI = float;
#else
using TI = double;
#endif
static int large_sym_invert_impl(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    InvParams p{0};
#if NFM_LARGE_PART == 2
    NFM_LSWITCH16(M, return (rec_launch<TI, InvStreamOp<TI, N, true>, true>(mat, nullptr, nullptr, out, 1, ni, p, stream)))
#else
    NFM_LSWITCH16(M, return (rec_launch<TI, InvStreamOp<TI, N, true>, true>(mat, nullptr, nullptr, out, 1, ni, p, stream)))
#endif
    return NFM_EFALLBACK;
}
#if NFM_LARGE_PART == 2
int large_sym_invert_f32(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream) { return large_sym_invert_impl(M, ni, mat, out, stream); }
#else
int large_sym_invert_f64(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream) { return large_sym_invert_impl(M, ni, mat, out, stream); }
#endif
#endif

#if NFM_LARGE_PART == 4 || NFM_LARGE_PART == 5
#if NFM_LARGE_PART == 4
using TB = float;
#else
using TB = double;
#endif
static int large_batch_inv_impl(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    InvParams p{0};
    NFM_LSWITCH16(N_, return (rec_launch<TB, InvStreamOp<TB, N, false>, true>(a, nullptr, nullptr, out, 1, ni, p, stream)))

    return NFM_EFALLBACK;
}
#if NFM_LARGE_PART == 4
int large_batch_inv_f32(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream) { return large_batch_inv_impl(N_, ni, a, out, stream); }
#else
int large_batch_inv_f64(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream) { return large_batch_inv_impl(N_, ni, a, out, stream); }
#endif
#endif

#if NFM_LARGE_PART == 6 || NFM_LARGE_PART == 7
#if NFM_LARGE_PART == 6
using TM = float;
#else
using TM = double;
#endif
static int large_sym_matvec_impl(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                                 const nfm_operand *inp, const nfm_operand *out, void *stream)
{
    MatvecParams p{mode};
    NFM_LSWITCH16(M, return (rec_launch<TM, MatvecOp<TM, N, NFM_MAT_SYM>, true>(mat, vec, inp, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
static int large_batch_det_impl(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    NoParamsB p{0};
    #if NFM_LARGE_PART == 6
    NFM_LSWITCH16(N_, return (rec_launch<TM, BatchDetOp<TM, N>, true>(a, nullptr, nullptr, out, 1, ni, p, stream)))
#else
    NFM_LSWITCH16(N_, return (rec_launch<TM, BatchDetOp<TM, N>, true>(a, nullptr, nullptr, out, 1, ni, p, stream)))
#endif
    return NFM_EFALLBACK;
}
#if NFM_LARGE_PART == 6
int large_sym_matvec_f32(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out, void *stream) { return large_sym_matvec_impl(M, mode, ni, mat, vec, inp, out, stream); }
int large_batch_det_f32(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream) { return large_batch_det_impl(N_, ni, a, out, stream); }
#else
int large_sym_matvec_f64(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out, void *stream) { return large_sym_matvec_impl(M, mode, ni, mat, vec, inp, out, stream); }
int large_batch_det_f64(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream) { return large_batch_det_impl(N_, ni, a, out, stream); }
#endif
#endif

} // namespace nfm

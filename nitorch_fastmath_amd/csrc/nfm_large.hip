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

// Coverage is set by what stays in registers (checked with -Rpass-analysis=kernel-resource-usage
// and timed on MI355X, profiles/r01c): beyond these ranges the spills make the register kernels
// slower than the LDS-resident ones, which take over.
//   inverse, compact symmetric : column-by-column LU (InvStreamOp)     f32 9..12, f64 9..14
//   inverse, general           : in-place Gauss-Jordan (BatchInvOp)    f32 9..13, f64 9..13
//   determinant, general       : LU (BatchDetOp)                       f32 9..16, f64 9..14
// NB (toolchain): with hipcc 7.2 the f32 column-by-column inverse gives WRONG, run-to-run
// varying results for N >= 13 (the f64 instantiation of the same source is correct; the
// kernels are far beyond 256 live registers and the symptom is that of a spill placed under
// a partial EXEC mask).  scripts/dbg_stream.hip reproduces it stand-alone.  Those orders
// therefore stay on the LDS-resident kernels, and tests/test_gpu_large_orders.py checks
// every order 9..16 of every op against the CPU restatement on thousands of matrices.

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
    NFM_LSWITCH_9_12(M, return (rec_launch<TI, InvStreamOp<TI, N, true>, true>(mat, nullptr, nullptr, out, 1, ni, p, stream)))
#else
    NFM_LSWITCH_9_14(M, return (rec_launch<TI, InvStreamOp<TI, N, true>, true>(mat, nullptr, nullptr, out, 1, ni, p, stream)))
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
    NFM_LSWITCH_9_13(N_, return (rec_launch<TB, BatchInvOp<TB, N>, true>(a, nullptr, nullptr, out, 1, ni, p, stream)))

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
    NFM_LSWITCH_9_14(N_, return (rec_launch<TM, BatchDetOp<TM, N>, true>(a, nullptr, nullptr, out, 1, ni, p, stream)))
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

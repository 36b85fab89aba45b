// nfm_large.hip -- register-resident kernels for orders 9..16 (see nfm_large.hpp).
// Compiled 32 times (-DNFM_LARGE_PART=0..31), one object per (function group, dtype, quarter
// of the order range), so that the big fully-unrolled eliminations build in parallel:
// PART = group * 8 + dtype * 4 + quarter; quarter q holds orders 9+2q and 10+2q.
#include "nfm_sym_ops.hpp"
#include "nfm_batched_ops.hpp"
#include "nfm_large.hpp"
#include "nfm_rowwave.hpp"

#ifndef NFM_LARGE_PART
#error "compile with -DNFM_LARGE_PART=0..31"
#endif

#define NFM_LGROUP (NFM_LARGE_PART / 8)
#define NFM_LF64 ((NFM_LARGE_PART / 4) % 2)
#define NFM_LQUART (NFM_LARGE_PART % 4)

namespace nfm {

#define NFM_LCASE(Nv, ...)    \
    case Nv: {                \
        constexpr int N = Nv; \
        __VA_ARGS__;          \
    } break;
#if NFM_LQUART == 0
#define NFM_LQ 0
#define NFM_LSWITCH16(Nexpr, ...) \
    switch (Nexpr) { NFM_LCASE(9, __VA_ARGS__) NFM_LCASE(10, __VA_ARGS__) default: break; }
#elif NFM_LQUART == 1
#define NFM_LQ 1
#define NFM_LSWITCH16(Nexpr, ...) \
    switch (Nexpr) { NFM_LCASE(11, __VA_ARGS__) NFM_LCASE(12, __VA_ARGS__) default: break; }
#elif NFM_LQUART == 2
#define NFM_LQ 2
#define NFM_LSWITCH16(Nexpr, ...) \
    switch (Nexpr) { NFM_LCASE(13, __VA_ARGS__) NFM_LCASE(14, __VA_ARGS__) default: break; }
#else
#define NFM_LQ 3
#define NFM_LSWITCH16(Nexpr, ...) \
    switch (Nexpr) { NFM_LCASE(15, __VA_ARGS__) NFM_LCASE(16, __VA_ARGS__) default: break; }
#endif

// Every op covers the orders 9..16 of both dtypes that rowwave_choice (nfm_rowwave.hpp) leaves
// to the lane-per-matrix form; the others (the ones that did not fit the register file) answer
// NFM_EFALLBACK without instantiating a kernel.  The pivoting code uses OPAQUE selects here
// (Sel<true>, nfm_smallmat.hpp): no data-dependent control flow is left, so whatever spill
// code the big eliminations need runs under a full EXEC mask, and most kernels need none
// (-Rpass-analysis=kernel-resource-usage: every f32 kernel and f64 up to 13 are spill-free).
//   inverse (compact symmetric and general): LU + column-by-column unit solves (InvStreamOp)
//   solve / determinants: Gaussian elimination with partial pivoting (SolveOp, DetOp, BatchDetOp)
// tests/test_gpu_large_orders.py checks every order of every op against the CPU restatement
// on thousands of matrices, twice.

#if NFM_LF64 == 0
using TL = float;
#define NFM_LNAME(op) NFM_LNAME2(op, f32, NFM_LQ)
#else
using TL = double;
#define NFM_LNAME(op) NFM_LNAME2(op, f64, NFM_LQ)
#endif
#define NFM_LNAME2(op, t, h) NFM_LNAME3(op, t, h)
// does the lane-per-matrix form keep (dtype TL, order N, op `what`)?
#define NFM_LKEEP(N, what) (rowwave_choice(sizeof(TL) == 8, N, what).rows == 0)
#define NFM_LNAME3(op, t, h) large_##op##_##t##_q##h

#if NFM_LGROUP == 0
int NFM_LNAME(sym_solve)(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                         const double *eps, void *stream)
{
    SolveParams p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? eps[i] : 0.0;
    NFM_LSWITCH16(M, if constexpr (NFM_LKEEP(N, RWW_SOLVE))
                     return (rec_launch<TL, SolveOp<TL, N, NFM_MAT_SYM>, true>(mat, vec, nullptr, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
int NFM_LNAME(sym_det)(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    NoParams p{0};
    NFM_LSWITCH16(M, if constexpr (NFM_LKEEP(N, RWW_DET_SYM))
                     return (rec_launch<TL, DetOp<TL, N>, true>(mat, nullptr, nullptr, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
#endif

#if NFM_LGROUP == 1
int NFM_LNAME(sym_invert)(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    InvParams p{0};
    NFM_LSWITCH16(M, if constexpr (NFM_LKEEP(N, RWW_INV_SYM))
                     return (rec_launch<TL, InvStreamOp<TL, N, true>, true>(mat, nullptr, nullptr, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
#endif

#if NFM_LGROUP == 2
int NFM_LNAME(batch_inv)(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    InvParams p{0};
    NFM_LSWITCH16(N_, if constexpr (NFM_LKEEP(N, RWW_INV_GEN))
                      return (rec_launch<TL, InvStreamOp<TL, N, false>, true>(a, nullptr, nullptr, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
#endif

#if NFM_LGROUP == 3
int NFM_LNAME(sym_matvec)(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                          const nfm_operand *inp, const nfm_operand *out, void *stream)
{
    MatvecParams p{mode};
    NFM_LSWITCH16(M, return (rec_launch<TL, MatvecOp<TL, N, NFM_MAT_SYM>, true>(mat, vec, inp, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
int NFM_LNAME(batch_det)(int N_, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    NoParamsB p{0};
    NFM_LSWITCH16(N_, if constexpr (NFM_LKEEP(N, RWW_DET_GEN))
                      return (rec_launch<TL, BatchDetOp<TL, N>, true>(a, nullptr, nullptr, out, 1, ni, p, stream)))
    return NFM_EFALLBACK;
}
#endif

} // namespace nfm

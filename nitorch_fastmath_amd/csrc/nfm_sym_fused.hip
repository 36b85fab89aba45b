// nfm_sym_fused.hip -- the Gauss-Newton step callers of the reference chain out of two calls,
// `sym_solve(sym_matmul(J, H), g)` (`_impl/sym.py:637-670` then `:327-398`), as ONE kernel: the
// compact (D x D) product never travels to HBM.  The arithmetic is the two
// Ops' arithmetic back to back in registers, so the result is bit-identical to the chained calls.
#include "nfm_sym_ops.hpp"

namespace nfm {

template <typename T, int K, int D, int HK>
struct MatmulSolveOp {
    using RA = Rec<K, D>;
    using RB = Rec<1, (HK == NFM_MAT_SYM ? sym_k(K) : K)>;
    using RC = Rec<1, D>;
    using RO = Rec<1, D>;
    using Params = SolveParams;
    static constexpr int TILE = pick_tile((RA::C + RB::C + 2 * D) * (int)sizeof(T) + 48);
    static __device__ __forceinline__ void apply(const T (&J)[RA::Cs], const T (&H)[RB::Cs], const T (&g)[D],
                                                 T (&x)[D], const Params &p)
    {
        T a[sym_k(D)], none[1] = {T(0)};
        const NoParams np{0};
        MatmulOp<T, K, D, HK>::apply(J, H, none, a, np);
        SolveOp<T, D, NFM_MAT_SYM>::apply(a, g, none, x, p);
    }
};

template <typename T, int HK>
static int matmul_solve_t(int K, int D, int64_t no, int64_t ni, const nfm_operand *jac, const nfm_operand *hess,
                          const nfm_operand *grad, const nfm_operand *out, const SolveParams &p, void *stream)
{
#define NFM_MS(Kv, Dv)                                                                                   \
    if (K == Kv && D == Dv)                                                                              \
        return (rec_launch<T, MatmulSolveOp<T, Kv, Dv, HK>>(jac, hess, grad, out, no, ni, p, stream));
    NFM_MS(1, 1) NFM_MS(1, 2) NFM_MS(1, 3) NFM_MS(1, 4)
    NFM_MS(2, 1) NFM_MS(2, 2) NFM_MS(2, 3) NFM_MS(2, 4)
    NFM_MS(3, 1) NFM_MS(3, 2) NFM_MS(3, 3) NFM_MS(3, 4)
    NFM_MS(4, 1) NFM_MS(4, 2) NFM_MS(4, 3) NFM_MS(4, 4)
#undef NFM_MS
    return NFM_ESIZE;
}

} // namespace nfm

using namespace nfm;

extern "C" int nfm_sym_matmul_solve(int dtype, int K, int D, int hess_kind, int64_t n_outer, int64_t n_inner,
                                    const nfm_operand *jac, const nfm_operand *hess, const nfm_operand *grad,
                                    const nfm_operand *out, const double *eps, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (K < 1 || K > 4 || D < 1 || D > 4) return NFM_ESIZE;
    if (hess_kind != NFM_MAT_SYM && hess_kind != NFM_MAT_DIAG) return NFM_EINVAL;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(jac, dtype, nonempty))) return rc;
    if ((rc = check_operand(hess, dtype, nonempty))) return rc;
    if ((rc = check_operand(grad, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    SolveParams p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < D) ? eps[i] : 0.0;
    if (dtype == NFM_F32)
        return hess_kind == NFM_MAT_SYM
                   ? matmul_solve_t<float, NFM_MAT_SYM>(K, D, n_outer, n_inner, jac, hess, grad, out, p, stream)
                   : matmul_solve_t<float, NFM_MAT_DIAG>(K, D, n_outer, n_inner, jac, hess, grad, out, p, stream);
    return hess_kind == NFM_MAT_SYM
               ? matmul_solve_t<double, NFM_MAT_SYM>(K, D, n_outer, n_inner, jac, hess, grad, out, p, stream)
               : matmul_solve_t<double, NFM_MAT_DIAG>(K, D, n_outer, n_inner, jac, hess, grad, out, p, stream);
}

/* nfm_oracle.c -- CPU restatement of the reference's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * backend: it may be loaded by tests/, by __graft_entry__.smoke() and by
 * bench.py's `cpu_baseline` leg, and by nothing else.  The product package
 * (nitorch_fastmath_amd) never imports, links or calls it and has no CPU
 * fallback of any kind.
 *
 * What it restates (file:line relative to /root/reference/nitorch_fastmath/):
 *   _impl/sym.py:16-60     sym_to_full
 *   _impl/sym.py:87-172    sym_matvec (+ add/sub forms named in sym.py:31-32)
 *   _impl/sym.py:186-398   sym_solve: closed forms M<=4, full + LU for M>4
 *   _impl/sym.py:401-452   sym_det (with quirk Q2 fixed: M from the compact dim)
 *   _impl/sym.py:455-493   sym_invert = M solves against basis vectors
 *   _impl/sym.py:496-670   sym_outer, sym_matmul (jhjn order)
 *   _impl/batched.py:21-190 batchdet / batchinv / batchmatvec (CPU = torch LU
 *                          fallbacks; TorchScript adjugate forms behind `closed`)
 *   reduce.py:255-510      nansum / nanmax / nanmin / sum / max / min
 *   _impl/qr.py:55-681     givens, givens_apply, householder(_apply), hessenberg(_sym),
 *                          qr_hessenberg, rq_hessenberg, eig_sym (nfm_oracle_qr.inc)
 *
 * Pinning: checked against golden vectors produced by the real reference in
 * the build container (tests/golden/make_golden.py -> tests/golden/ npz files);
 * see tests/test_oracle_golden.py.  The closed forms (M <= 4) follow the
 * reference's operation order exactly and are compiled with -ffp-contract=off.
 *
 * Layout: every operand is contiguous, batch-major ("AoS"): mat (n, K),
 * vec (n, M), full matrices (n, N, N) row-major.  dtype: 0 = f32, 1 = f64.
 */
#include <stdint.h>
#include <math.h>
#include <stddef.h>

#define NFM_MAXM 16

#define T float
#define SUF f32
#define FABS fabsf
#define SQRT sqrtf
#define ISFINITE(x) isfinite(x)
#define ADDCMUL(self, a, b) fmaf((a), (b), (self))
#include "nfm_oracle_body.inc"
#undef T
#undef SUF
#undef FABS
#undef SQRT
#undef ISFINITE
#undef ADDCMUL

#define T double
#define SUF f64
#define FABS fabs
#define SQRT sqrt
#define ISFINITE(x) isfinite(x)
#define ADDCMUL(self, a, b) fma((a), (b), (self))
#include "nfm_oracle_body.inc"
#undef T
#undef SUF
#undef FABS
#undef SQRT
#undef ISFINITE
#undef ADDCMUL

#define BAD_DTYPE (-2)
#define BAD_SIZE (-3)
#define CHECK_M(M) do { if ((M) < 1 || (M) > NFM_MAXM) return BAD_SIZE; } while (0)

int nfm_oracle_sym_solve(int dtype, int M, int64_t n, int kind, const void *mat, const void *vec, void *out)
{
    CHECK_M(M);
    if (dtype == 0) return nfm_oracle_sym_solve_f32(M, n, kind, mat, vec, out);
    if (dtype == 1) return nfm_oracle_sym_solve_f64(M, n, kind, mat, vec, out);
    return BAD_DTYPE;
}

int nfm_oracle_sym_matvec(int dtype, int M, int64_t n, int kind, int sign, const void *mat, const void *vec,
                          const void *inp, void *out)
{
    CHECK_M(M);
    if (dtype == 0) return nfm_oracle_sym_matvec_f32(M, n, kind, sign, mat, vec, inp, out);
    if (dtype == 1) return nfm_oracle_sym_matvec_f64(M, n, kind, sign, mat, vec, inp, out);
    return BAD_DTYPE;
}

int nfm_oracle_sym_invert(int dtype, int M, int64_t n, int diag_only, const void *mat, void *out)
{
    CHECK_M(M);
    if (dtype == 0) return nfm_oracle_sym_invert_f32(M, n, diag_only, mat, out);
    if (dtype == 1) return nfm_oracle_sym_invert_f64(M, n, diag_only, mat, out);
    return BAD_DTYPE;
}

int nfm_oracle_sym_det(int dtype, int M, int64_t n, const void *mat, void *out)
{
    CHECK_M(M);
    if (dtype == 0) return nfm_oracle_sym_det_f32(M, n, mat, out);
    if (dtype == 1) return nfm_oracle_sym_det_f64(M, n, mat, out);
    return BAD_DTYPE;
}

int nfm_oracle_sym_to_full(int dtype, int M, int64_t n, const void *mat, void *out)
{
    CHECK_M(M);
    if (dtype == 0) return nfm_oracle_sym_to_full_f32(M, n, mat, out);
    if (dtype == 1) return nfm_oracle_sym_to_full_f64(M, n, mat, out);
    return BAD_DTYPE;
}

int nfm_oracle_sym_outer(int dtype, int M, int64_t n, const void *x, void *out)
{
    CHECK_M(M);
    if (dtype == 0) return nfm_oracle_sym_outer_f32(M, n, x, out);
    if (dtype == 1) return nfm_oracle_sym_outer_f64(M, n, x, out);
    return BAD_DTYPE;
}

int nfm_oracle_sym_matmul(int dtype, int K, int D, int64_t n, int hess_diag, const void *jac, const void *hess,
                          void *out)
{
    CHECK_M(K);
    CHECK_M(D);
    if (dtype == 0) return nfm_oracle_sym_matmul_f32(K, D, n, hess_diag, jac, hess, out);
    if (dtype == 1) return nfm_oracle_sym_matmul_f64(K, D, n, hess_diag, jac, hess, out);
    return BAD_DTYPE;
}

int nfm_oracle_batch_inv(int dtype, int N, int64_t n, int closed, const void *a, void *out)
{
    CHECK_M(N);
    if (dtype == 0) return nfm_oracle_batch_inv_f32(N, n, closed, a, out);
    if (dtype == 1) return nfm_oracle_batch_inv_f64(N, n, closed, a, out);
    return BAD_DTYPE;
}

int nfm_oracle_batch_det(int dtype, int N, int64_t n, int closed, const void *a, void *out)
{
    CHECK_M(N);
    if (dtype == 0) return nfm_oracle_batch_det_f32(N, n, closed, a, out);
    if (dtype == 1) return nfm_oracle_batch_det_f64(N, n, closed, a, out);
    return BAD_DTYPE;
}

int nfm_oracle_batch_matvec(int dtype, int rows, int cols, int64_t n, const void *a, const void *v, void *out)
{
    CHECK_M(rows);
    CHECK_M(cols);
    if (dtype == 0) return nfm_oracle_batch_matvec_f32(rows, cols, n, a, v, out);
    if (dtype == 1) return nfm_oracle_batch_matvec_f64(rows, cols, n, a, v, out);
    return BAD_DTYPE;
}

int nfm_oracle_reduce(int dtype, int op, int64_t outer, int64_t red, int64_t inner, const void *x, void *out,
                      int out_f64)
{
    if (op < 0 || op > 5) return BAD_SIZE;
    if (dtype == 0) return nfm_oracle_reduce_f32(op, outer, red, inner, x, out, out_f64);
    if (dtype == 1) return nfm_oracle_reduce_f64(op, outer, red, inner, x, out, out_f64);
    return BAD_DTYPE;
}

#define DISPATCH(call32, call64) do { if (dtype == 0) return call32; if (dtype == 1) return call64; return BAD_DTYPE; } while (0)

int nfm_oracle_qr_givens(int dtype, int64_t n, const void *x, const void *y, void *c, void *s)
{
    DISPATCH(nfm_oracle_qr_givens_f32(n, x, y, c, s), nfm_oracle_qr_givens_f64(n, x, y, c, s));
}

int nfm_oracle_qr_givens_apply(int dtype, int N, int64_t n, int side, int i, int j, void *a, const void *c,
                               const void *s)
{
    CHECK_M(N);
    if (i < 0 || j < 0 || i >= N || j >= N) return BAD_SIZE;
    DISPATCH(nfm_oracle_qr_givens_apply_f32(N, n, side, i, j, a, c, s),
             nfm_oracle_qr_givens_apply_f64(N, n, side, i, j, a, c, s));
}

int nfm_oracle_qr_householder(int dtype, int N, int64_t n, int basis, void *x, void *alpha)
{
    CHECK_M(N);
    DISPATCH(nfm_oracle_qr_householder_f32(N, n, basis, x, alpha), nfm_oracle_qr_householder_f64(N, n, basis, x, alpha));
}

int nfm_oracle_qr_householder_apply(int dtype, int N, int m, int64_t n, int side, void *a, const void *u)
{
    CHECK_M(N);
    if (m < 1 || m > N) return BAD_SIZE;
    DISPATCH(nfm_oracle_qr_householder_apply_f32(N, m, n, side, a, u),
             nfm_oracle_qr_householder_apply_f64(N, m, n, side, a, u));
}

int nfm_oracle_qr_hessenberg(int dtype, int N, int64_t n, void *a, void *upack)
{
    CHECK_M(N);
    DISPATCH(nfm_oracle_qr_hessenberg_f32(N, n, a, upack), nfm_oracle_qr_hessenberg_f64(N, n, a, upack));
}

int nfm_oracle_qr_hessenberg_sym(int dtype, int N, int64_t n, int upper, int fill, void *a, void *upack)
{
    CHECK_M(N);
    DISPATCH(nfm_oracle_qr_hessenberg_sym_f32(N, n, upper, fill, a, upack),
             nfm_oracle_qr_hessenberg_sym_f64(N, n, upper, fill, a, upack));
}

int nfm_oracle_qr_qr_hessenberg(int dtype, int N, int64_t n, void *a, void *q)
{
    CHECK_M(N);
    DISPATCH(nfm_oracle_qr_qr_hessenberg_f32(N, n, a, q), nfm_oracle_qr_qr_hessenberg_f64(N, n, a, q));
}

int nfm_oracle_qr_rq_hessenberg(int dtype, int N, int64_t n, int sym, int true_rq, void *a, void *u)
{
    CHECK_M(N);
    DISPATCH(nfm_oracle_qr_rq_hessenberg_f32(N, n, sym, true_rq, a, u),
             nfm_oracle_qr_rq_hessenberg_f64(N, n, sym, true_rq, a, u));
}

int nfm_oracle_qr_eig_sym(int dtype, int N, int64_t n, int upper, int compute_u, int max_iter, double tol, void *a,
                          void *vals, void *vecs)
{
    CHECK_M(N);
    DISPATCH(nfm_oracle_qr_eig_sym_f32(N, n, upper, compute_u, max_iter, tol, a, vals, vecs, NULL),
             nfm_oracle_qr_eig_sym_f64(N, n, upper, compute_u, max_iter, tol, a, vals, vecs, NULL));
}

/* same, also reporting the QR sweeps spent per matrix and per active block size: sweeps (n, N) int32,
   entry m - 1 = sweeps on the m x m block (the divergence profile of the per-lane GPU kernel) */
int nfm_oracle_qr_eig_sym_sweeps(int dtype, int N, int64_t n, int upper, int compute_u, int max_iter, double tol,
                                 void *a, void *vals, void *vecs, int *sweeps)
{
    CHECK_M(N);
    DISPATCH(nfm_oracle_qr_eig_sym_f32(N, n, upper, compute_u, max_iter, tol, a, vals, vecs, sweeps),
             nfm_oracle_qr_eig_sym_f64(N, n, upper, compute_u, max_iter, tol, a, vals, vecs, sweeps));
}

int nfm_oracle_version(void) { return 1; }

/* nfm_hip.h -- C ABI of libnfm_hip.so, the MI355X (gfx950) backend for the
 * per-element small-matrix hot path of nitorch-fastmath.
 *
 * The reference has no FFI of its own for this path: its public functions are
 * plain Python (`nitorch_fastmath/sym.py:28-37`, `batched.py:16-17`,
 * `qr.py:1-11`, `reduce.py:38-41`) and the shipped `sym_*` implementation comes
 * from the external `jitfields.sym` module (`sym.py:37`).  This header is the
 * boundary a maintainer binds instead (ctypes stub in INTEGRATION.md): one
 * entry point per reference function family, plain pointers / sizes / strides,
 * no torch types.  Every entry point
 *   - launches asynchronously on the `hipStream_t` passed as `stream`
 *     (NULL = the default stream) on the CURRENT device, never synchronises,
 *     never allocates or frees memory (graph-capture safe);
 *   - returns 0 on success, a negative NFM_E* code for a rejected argument,
 *     or a positive hipError_t if the launch itself failed;
 *   - is reentrant and stateless.
 *
 * Batch model.  The facade flattens the broadcast batch shape to two levels,
 * n_outer x n_inner (n_outer == 1 for ordinary contiguous tensors; two levels
 * cover channel-first fields (B, C, *spatial) without a copy).  Strides are in
 * ELEMENTS; 0 = broadcast.  An operand is read as
 *     ptr[o * stride_outer + i * stride_inner + r * stride_row + c * stride_col]
 * with (r, c) the matrix row/column for full matrices and r = 0, c = component
 * for vectors and compact-symmetric storage.
 *
 * Compact symmetric layout (reference `sym.py:7-14`, `_impl/sym.py:21-27`):
 * K = M(M+1)/2 components, the diagonal first, then the strict upper triangle
 * row by row: [a00 a11 .. a(M-1)(M-1) | a01 a02 .. a0(M-1) a12 ..].
 */
#ifndef NFM_HIP_H
#define NFM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFM_VERSION 5 /* 5: NFM_MAT_PIVOTED / NFM_INVERT_PIVOTED; 2: nfm_qr_eig_sym takes flags (NFM_EIG_*); nfm_reduce_median added; 3: nfm_reduce_median_mid;
                         4: nfm_qr_eig_sym flags: bit 2 is NFM_EIG_FAST (flags == with_u is the reference order again) */
#define NFM_MAX_DIM 16 /* largest matrix order handled (3x3 .. 16x16 and below) */

/* dtype codes */
#define NFM_F32 0
#define NFM_F64 1

/* error codes (negative) */
#define NFM_OK 0
#define NFM_EINVAL (-1)   /* null pointer / negative size / bad flag */
#define NFM_EDTYPE (-2)   /* unsupported dtype code */
#define NFM_ESIZE (-3)    /* matrix order outside 1..NFM_MAX_DIM, or batch too large */
#define NFM_EALIGN (-4)   /* pointer not aligned to the element size */
#define NFM_EWORKSPACE (-5) /* workspace too small */

/* `mat_kind` of the sym_* entry points: how the matrix operand's last dim of
 * length NN is interpreted, reference `sym.py:16-24`. */
#define NFM_MAT_SYM 0  /* NN = M(M+1)/2 compact symmetric */
#define NFM_MAT_DIAG 1 /* NN = M       diagonal           */
#define NFM_MAT_SCAL 2 /* NN = 1       scaled identity    */
#define NFM_MAT_FULL 3 /* NN = M*M     full, row-major via stride_row/stride_col */
/* flag, OR-ed into `mat_kind` of nfm_sym_solve (and bit 1 = value 2 of `diag_only` of nfm_sym_invert): orders 9..16 go
 * straight to the pivoted elimination, without the attempt that serves positive definite matrices -- for callers who
 * know their matrices are indefinite (a batch of those pays the attempt for nothing) */
#define NFM_MAT_PIVOTED 16
#define NFM_INVERT_DIAG 1
#define NFM_INVERT_PIVOTED 2

typedef struct nfm_operand {
    void *ptr;            /* device pointer */
    int64_t stride_outer; /* elements */
    int64_t stride_inner; /* elements */
    int64_t stride_row;   /* elements; full matrices only */
    int64_t stride_col;   /* elements; component stride for vectors / compact storage */
} nfm_operand;

/* ------------------------------------------------------------------ sym ---- */

/* x = mat \ vec.  Replaces `sym_solve` / `sym_solve_` (`sym.py:33`; in-repo
 * implementation `_impl/sym.py:327-398`).  M <= 4: the reference's closed forms,
 * evaluated in its operation order (bit-identical to its CPU path); M > 4: LU with
 * partial pivoting of the full matrix in registers / LDS, like the
 * `torch.linalg.solve` branch (`_impl/sym.py:392-396`); contiguous operands at
 * M = 9..16 (here, in nfm_sym_invert and in nfm_sym_det): the unpivoted LDL^T of the
 * compact record first, the pivoted elimination for every wavefront that holds a matrix
 * which is not positive definite -- same answers within rounding.  `eps` (NULL or M doubles on
 * the HOST) is added to the diagonal first (documented intent of `_impl/sym.py:356-357`).
 * `out` may alias `vec` (in-place variant). */
int nfm_sym_solve(int dtype, int M, int mat_kind, int64_t n_outer, int64_t n_inner,
                  const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                  const double *eps, void *stream);

/* out = mat * vec            (mode  0)  `sym_matvec`        `_impl/sym.py:134-172`
 * out = inp + mat * vec      (mode +1)  `sym_addmatvec(_)`  `sym.py:31`
 * out = inp - mat * vec      (mode -1)  `sym_submatvec(_)`  `sym.py:32`
 * `inp` is ignored for mode 0; `out` may alias `inp`. */
int nfm_sym_matvec(int dtype, int M, int mat_kind, int mode, int64_t n_outer, int64_t n_inner,
                   const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *inp,
                   const nfm_operand *out, void *stream);

/* out = compact inverse of a compact symmetric matrix (diag_only & NFM_INVERT_DIAG: its M
 * diagonal entries; diag_only & NFM_INVERT_PIVOTED: see NFM_MAT_PIVOTED).  Replaces
 * `sym_invert` / `sym_invert_` (`sym.py:34`, `_impl/sym.py:455-493`).  One factorisation per
 * matrix (the reference runs M full solves).  `out` may alias `mat` when the diagonal flag
 * is not set. */
int nfm_sym_invert(int dtype, int M, int diag_only, int64_t n_outer, int64_t n_inner,
                   const nfm_operand *mat, const nfm_operand *out, void *stream);

/* determinant of compact symmetric matrices, `sym_det` `_impl/sym.py:401-452` (quirk Q2
 * fixed: M comes from the compact dim).  out: one element per matrix (stride_col unused). */
int nfm_sym_det(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                const nfm_operand *out, void *stream);

/* compact -> full (M x M), `sym_to_full` `_impl/sym.py:16-60`. */
int nfm_sym_to_full(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                    const nfm_operand *out, void *stream);

/* compact x x^T of a vector, `sym_outer` `_impl/sym.py:496-528`. */
int nfm_sym_outer(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *x,
                  const nfm_operand *out, void *stream);

/* out_ii = x_i y_i, out_ij = x_i y_j + x_j y_i (i < j), negated when neg != 0: the pull-back
 * of the full-matrix cotangent x y^T onto compact storage.  No reference counterpart: it is
 * the building block of the backward passes of sym_matvec / sym_solve (autograd). */
int nfm_sym_outer2(int dtype, int M, int neg, int64_t n_outer, int64_t n_inner, const nfm_operand *x,
                   const nfm_operand *y, const nfm_operand *out, void *stream);

/* compact J^T H J, `sym_matmul` `_impl/sym.py:637-670`; jac is (K x D) full, hess compact
 * (hess_kind NFM_MAT_SYM) or diagonal (NFM_MAT_DIAG).  For K == D in {2, 3} the reference
 * evaluates J H J^T (quirk Q16, `jhj2`/`jhj3` `_impl/sym.py:540-597`); so does this. */
int nfm_sym_matmul(int dtype, int K, int D, int hess_kind, int64_t n_outer, int64_t n_inner,
                   const nfm_operand *jac, const nfm_operand *hess, const nfm_operand *out,
                   void *stream);

/* EXTENSION (no counterpart in the reference): x = (J^T H J)^-1 g in one kernel -- the
 * Gauss-Newton step its callers chain as `sym_solve(sym_matmul(jac, hess), grad, eps)`
 * (`_impl/sym.py:637-670` then `:327-398`), without the HBM round trip of the compact
 * (D x D) product.  Same arithmetic as the two calls: bit-identical results.  K, D in 1..4
 * (NFM_ESIZE otherwise: chain the two calls); eps as for nfm_sym_solve, length D. */
int nfm_sym_matmul_solve(int dtype, int K, int D, int hess_kind, int64_t n_outer, int64_t n_inner,
                         const nfm_operand *jac, const nfm_operand *hess, const nfm_operand *grad,
                         const nfm_operand *out, const double *eps, void *stream);

/* -------------------------------------------------------------- batched ---- */

#define NFM_FLAG_TS_PERTURB 1 /* add (max|A| - min|A|) * 1e-12 to det for N in {2,3}:
                                 the TorchScript forms `inv2`/`inv3`, `_impl/batched.py:74-76,94-96` */

/* out = a^-1 for general N x N matrices, `batchinv` `_impl/batched.py:101-130`.
 * N <= 3: adjugate / det; N > 3: in-register Gauss-Jordan with partial pivoting
 * (the reference falls back to LAPACK getrf/getri there); contiguous operands at
 * N = 9..16 (float64: 9..13), here and in nfm_batch_det: the elimination without row
 * exchanges first, accepted while every diagonal pivot is within 1/8 of its column's
 * maximum (threshold pivoting), the pivoted elimination for the groups of matrices that
 * hold one which needs an exchange -- same answers within rounding.  out may alias a. */
int nfm_batch_inv(int dtype, int N, int flags, int64_t n_outer, int64_t n_inner,
                  const nfm_operand *a, const nfm_operand *out, void *stream);

/* det(a), `batchdet` `_impl/batched.py:35-63`. */
int nfm_batch_det(int dtype, int N, int64_t n_outer, int64_t n_inner, const nfm_operand *a,
                  const nfm_operand *out, void *stream);

/* out = a v for (rows x cols) matrices, `batchmatvec` `_impl/batched.py:154-190`. */
int nfm_batch_matvec(int dtype, int rows, int cols, int64_t n_outer, int64_t n_inner,
                     const nfm_operand *a, const nfm_operand *v, const nfm_operand *out,
                     void *stream);

/* ------------------------------------------------------------ reductions ---- */

#define NFM_RED_NANSUM 0 /* `nansum` reduce.py:471-510 : NaN -> 0              */
#define NFM_RED_NANMAX 1 /* `nanmax` reduce.py:255-316 : NaN -> -inf           */
#define NFM_RED_NANMIN 2 /* `nanmin` reduce.py:319-380 : NaN -> +inf           */
#define NFM_RED_SUM 3    /* `sum`    reduce.py:431-468 : NaN propagates        */
#define NFM_RED_MAX 4    /* `max`    reduce.py:145-197                         */
#define NFM_RED_MIN 5    /* `min`    reduce.py:200-252                         */
#define NFM_RED_NANCOUNT 6 /* number of non-NaN elements (weights of `nanmean` reduce.py:591-594) */
#define NFM_RED_NANSUMSQ 7 /* sum of squares of the non-NaN elements (`nanvar` reduce.py:679-680) */

/* bytes of device workspace `nfm_reduce_all` needs (independent of n). */
size_t nfm_reduce_workspace_bytes(void);

/* Full reduction of n contiguous elements to ONE scalar.  Sums are accumulated in
 * double whatever the input dtype; `out_dtype` is the dtype of *out (the `dtype=`
 * argument of the reference functions).  n may exceed 2^32.  workspace: device
 * memory of nfm_reduce_workspace_bytes() bytes, 16-byte aligned. */
int nfm_reduce_all(int dtype, int op, int out_dtype, int64_t n, const void *x, void *workspace,
                   size_t workspace_bytes, void *out, void *stream);

/* Reduction over the middle axis of a contiguous (outer, red, inner) view
 * (`_reduce_index` / `sum` with `dim=`, `reduce.py:49-142`, `:431-510`); out is (outer, inner)
 * of `out_dtype`; idx (may be NULL; max/min ops only) receives the int64 position along `red`
 * of the selected element (first occurrence).  Shapes with few outputs and a long reduced
 * axis are cut into chunks reduced in parallel and folded in a fixed order: they need
 * nfm_reduce_dim_workspace_bytes(...) bytes of device workspace (0 for the other shapes;
 * workspace may then be NULL).  The plan depends on the shape only: results are reproducible. */
size_t nfm_reduce_dim_workspace_bytes(int dtype, int op, int64_t outer, int64_t red, int64_t inner,
                                      int want_idx);
int nfm_reduce_dim(int dtype, int op, int out_dtype, int64_t outer, int64_t red, int64_t inner,
                   const void *x, void *workspace, size_t workspace_bytes, void *out, int64_t *idx,
                   void *stream);

/* One pass over a contiguous (outer, red, inner) view producing, per (outer, inner) entry,
 * four doubles [count, sum(x - K), sum((x - K)^2), K] over the non-NaN elements of the
 * reduced axis (K = a finite element of that slice, chosen by the kernel).  The raw material
 * of `nanmean` / `nanvar` / `nanstd` / `mean` / `var` / `std` (`reduce.py:513-763`).
 * workspace: nfm_reduce_moments_workspace_bytes(...) bytes. */
size_t nfm_reduce_moments_workspace_bytes(int dtype, int64_t outer, int64_t red, int64_t inner);
int nfm_reduce_moments(int dtype, int64_t outer, int64_t red, int64_t inner, const void *x,
                       void *workspace, size_t workspace_bytes, double *out, void *stream);

/* mean / var / std over the middle axis in ONE pass (the moments above, finished in the
 * kernel): `mean` `reduce.py:513-550`, `nanmean :553-594`, `var :597-635`, `nanvar :638-685`,
 * `std :688-726`, `nanstd :729-763`.  stat = NFM_STAT_MEAN|VAR|STD, or-ed with
 * NFM_STAT_OMITNAN (ignore NaNs; otherwise a NaN in the slice gives NaN) and
 * NFM_STAT_UNBIASED (var/std: multiply by n / (n - 1), `reduce.py:682-684`).
 * out is (outer, inner) of `out_dtype`; workspace as for nfm_reduce_moments. */
#define NFM_STAT_MEAN 0
#define NFM_STAT_VAR 1
#define NFM_STAT_STD 2
#define NFM_STAT_OMITNAN 4
#define NFM_STAT_UNBIASED 8
int nfm_reduce_stat(int dtype, int stat, int out_dtype, int64_t outer, int64_t red, int64_t inner,
                    const void *x, void *workspace, size_t workspace_bytes, void *out, void *stream);

/* Median of every row of a contiguous (rows, red) array, `median` `reduce.py:384-428` (which
 * moves the reduced dims last and calls torch.median; the facade does the same move): the
 * LOWER median (rank (count - 1) / 2), on order-preserving integer keys: a register sorting network
 * per row for rows of up to 128 elements (float64: 64), radix selection for longer ones.
 * omitnan = 0: a NaN in a row makes its median NaN (idx: the first NaN) -- what the reference
 * computes; omitnan = 1: the median of the non-NaN elements (all NaN: NaN, idx 0) -- what its
 * docstring promises (quirk Q14).  val: (rows) of `dtype`; idx: (rows) int64 or NULL = the first
 * position holding the median value.  Rows of more than 1024 elements need a workspace of
 * nfm_reduce_median_workspace_bytes(rows, red) bytes (0 for shorter rows) and rows <= 65535. */
size_t nfm_reduce_median_workspace_bytes(int64_t rows, int64_t red);
int nfm_reduce_median(int dtype, int omitnan, int64_t rows, int64_t red, const void *x, void *workspace,
                      size_t workspace_bytes, void *val, void *idx, void *stream);

/* The same median over the MIDDLE dim of a contiguous (outer, red, inner) array -- the channel dim of a
 * channel-first field -- without the transposing copy that moving the reduced dim last would cost (the
 * reference makes that copy: `reduce.py:112-113`).  One row per lane, so only for
 * 2 <= red <= nfm_reduce_median_lane_max(dtype) (128 for float32, 64 for float64; NFM_ESIZE beyond).
 * val / idx: (outer, inner), idx = position along the reduced dim. */
int nfm_reduce_median_lane_max(int dtype);
int nfm_reduce_median_mid(int dtype, int omitnan, int64_t outer, int64_t red, int64_t inner, const void *x, void *val,
                          void *idx, void *stream);

/* ------------------------------------------------------------------- qr ---- */
/* Real dtypes.  Multi-output routines write ONE packed, contiguous output record per
 * matrix into `out` (n_outer * n_inner records, batch-major); the layout of the record
 * is given with each entry point.  Inputs are ordinary strided operands. */

#define NFM_SIDE_LEFT 0
#define NFM_SIDE_RIGHT 1
#define NFM_SIDE_BOTH 2

/* c = x / r, s = -y / r, r = sqrt(x^2 + y^2); r == 0 -> (1, 0).  `givens` `_impl/qr.py:326-369`.
 * x, y: one element per batch entry; out record: [c, s]. */
int nfm_qr_givens(int dtype, int64_t n_outer, int64_t n_inner, const nfm_operand *x,
                  const nfm_operand *y, void *out, void *stream);

/* IN PLACE on `a` (N x N): rotate rows (left), columns (right) or both i and j.
 * `givens_apply_` `_impl/qr.py:370-429`.  c, s: N components per batch entry (stride_col 0 =
 * one coefficient for the whole row/column, the usual case). */
int nfm_qr_givens_apply(int dtype, int N, int side, int i, int j, int64_t n_outer, int64_t n_inner,
                        const nfm_operand *a, const nfm_operand *c, const nfm_operand *s, void *stream);

/* Householder vector of x (length N) reflecting onto component `basis`, and the projection
 * alpha.  `householder_` `_impl/qr.py:55-69`.  out record: [u (N) | alpha]. */
int nfm_qr_householder(int dtype, int N, int basis, int64_t n_outer, int64_t n_inner,
                       const nfm_operand *x, void *out, void *stream);

/* IN PLACE on `a` (N x N): apply P = I - 2 u u^T, u of length m acting on the trailing m
 * rows (left) / columns (right).  One reflector of `householder_apply_` `_impl/qr.py:72-106`. */
int nfm_qr_householder_apply(int dtype, int N, int m, int side, int64_t n_outer, int64_t n_inner,
                             const nfm_operand *a, const nfm_operand *u, void *stream);

/* Householder reduction to Hessenberg form (sym == 0: `hessenberg_` `_impl/qr.py:117-141`) or
 * of a symmetric matrix to tridiagonal form reading only the `upper` / lower triangle
 * (sym != 0: `hessenberg_sym_upper_/lower_` `:280-323`; output filled symmetric).
 * out record: [H (N*N row-major) | with_u: (N-2) reflectors, reflector k in a slot of N-1
 * elements, its N-1-k entries first, zero padded]. */
int nfm_qr_hessenberg(int dtype, int N, int sym, int upper, int with_u, int64_t n_outer,
                      int64_t n_inner, const nfm_operand *a, void *out, void *stream);

/* Q, R of an upper-Hessenberg matrix by N-1 Givens rotations, `qr_hessenberg_`
 * `_impl/qr.py:432-454`.  out record: [Q (N*N) | R (N*N)]. */
int nfm_qr_qr_hessenberg(int dtype, int N, int64_t n_outer, int64_t n_inner, const nfm_operand *h,
                         void *out, void *stream);

/* One QR step H <- R Q (and U <- U Q when u != NULL), `rq_hessenberg_` `_impl/qr.py:457-530`.
 * sym != 0: the tridiagonal shortcut of the reference; sym == 0: the true R Q for any
 * Hessenberg input (quirk Q8 fixed).  out record: [H' (N*N) | U' (N*N) if u]. */
int nfm_qr_rq_hessenberg(int dtype, int N, int sym, int64_t n_outer, int64_t n_inner,
                         const nfm_operand *h, const nfm_operand *u, void *out, void *stream);

/* Eigenvalues (unsorted, deflation order) and optionally eigenvectors of symmetric matrices:
 * tridiagonalisation + explicit QR with Wilkinson shifts, `eig_sym` `qr.py:30-100`,
 * `_fwd_eig_sym` `_impl/qr.py:665-681`.  Convergence is judged per matrix with the
 * reference's criterion (quirk Q9).  out record: [vals (N) | NFM_EIG_VECTORS: vecs (N*N
 * row-major, eigenvectors in columns)].
 * flags: NFM_EIG_VECTORS  also compute the eigenvectors (`compute_u`); flags == 0 / 1 is exactly the
 *                          reference's `compute_u` argument and selects the reference's arithmetic:
 *                          its operation order, its tolerance, correctly rounded division and
 *                          square root -- bit-identical to the CPU restatement (same deflation
 *                          order, same eigenvector signs).
 *        NFM_EIG_FAST     opt-in: the sweeps use v_rsq + Newton steps (one for float32, two for
 *                          float64) and fma contraction, the last 2x2 block is diagonalised by one
 *                          Jacobi rotation, and `tol` is floored at the working precision of the
 *                          dtype, max(tol, (eps/4)^2): as accurate against the exact eigenvalues
 *                          and 2-3x the throughput, but the deflation ORDER and the eigenvector
 *                          SIGNS differ from the reference's for a share of the matrices (float32:
 *                          2 % at 3x3, 35 % at 8x8) and float32 values by up to 1.4e-6. */
#define NFM_EIG_VECTORS 1
#define NFM_EIG_FAST 2
int nfm_qr_eig_sym(int dtype, int N, int upper, int flags, int max_iter, double tol,
                   int64_t n_outer, int64_t n_inner, const nfm_operand *a, void *out, void *stream);

/* ------------------------------------------------------------------- misc ---- */

const char *nfm_strerror(int code);
int nfm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NFM_HIP_H */

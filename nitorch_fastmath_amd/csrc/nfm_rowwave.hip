// nfm_rowwave.hip -- orders 9..16 with ONE MATRIX PER 16 / R LANES (R = 1, 2 or 4 rows per lane).
//
// A 16x16 float64 matrix is 512 dwords: the whole register file of a lane.  The lane-per-matrix
// kernels of nfm_large.hip therefore spilled at float64 orders 13..16 (0.4-1.5 TB/s) and run at
// one wave per SIMD long before that.  Here a matrix is spread over the 16 / R lanes of a DPP row:
// a lane holds R rows (row ids lane + t * 16 / R), a wavefront works on 4 R matrices, a workgroup
// of 256 / R lanes on 16.
//
//   * pivot search: max over the unused rows of an integer key that orders |a_rk| -- over the
//     lane's own rows first, then over the lanes of the matrix by DPP (row_ror / quad_perm /
//     row_half_mirror) with a v_max_u32 each; the pivot lane is the lowest set bit of the
//     matrix's bits of a ballot;
//   * elimination: Gauss-Jordan WITHOUT row exchanges -- the pivot row stays where it is and
//     reaches the other lanes value by value, by ds_bpermute (the LDS crossbar, no LDS memory) or
//     through a per-matrix LDS slot (template LB); every row is updated with one fma per value
//     (the pivot row's own multiplier is 0).  All lanes work in every step, so the Gauss-Jordan
//     form (no back substitution) is free.  One broadcast serves the R rows of every lane: time is
//     proportional to the broadcasts per matrix, which is why R = 2 / 4 beat R = 1 until
//     registers bite (profiles/r02/rowwave_table.md, rowwave_counters.md);
//   * inverse: the in-place Gauss-Jordan on [A | I], rows divided by their pivots at the end; the
//     lane that pivoted at step k ends up with row k of A^-1 whose l-th entry belongs to the
//     column = row id of the pivot of step l, undone while writing the LDS image;
//   * determinant: product of the pivots, sign from the Lehmer code of the pivot sequence;
//   * HBM traffic: the tile's records are contiguous, so they are streamed with 16-byte accesses
//     through an LDS image (rows padded to an odd number of 16-byte slots), exactly the
//     algorithmic bytes; a base that is only element-aligned takes element accesses.
//   The form (R, LB) of every (dtype, order, op) is rowwave_choice (nfm_rowwave.hpp).
//
// 1 / pivot is v_rcp + Newton steps (2 for float64, 1 for float32: <= 1.5 ulp) instead of the
// IEEE division sequence; pivots are the same as the CPU restatement's (partial pivoting, first
// maximum), the rounding differs -- parity is asserted through the error model of the tests
// (tests/test_gpu_large_orders.py), like every LU-based result beyond the closed forms.
//
// Reference paths replaced: `torch.linalg.solve` of the densified matrix (`_impl/sym.py:392-396`),
// `a.inverse()` / `a.det()` (`_impl/batched.py:119-120`, `:53-54`).
#include "nfm_rowwave_core.hpp"

namespace nfm {
namespace roww {

// contiguous batch-major records (what the facade allocates; the base need not be 16-byte aligned)
static bool rec_contig(const nfm_operand *o, int64_t rec, int rows, int cols, size_t elem)
{
    if (o == nullptr || o->ptr == nullptr) return false;
    if (reinterpret_cast<uintptr_t>(o->ptr) % elem != 0) return false;
    if (o->stride_inner != rec) return false;
    if (cols > 1 && o->stride_col != 1) return false;
    if (rows > 1 && o->stride_row != cols) return false;
    return true;
}

template <typename T, int N, int OP, int R, bool LB>
static int launch_r(const void *a, const void *b, void *o, int64_t n, const RowParams<T> &p, void *stream)
{
    constexpr size_t lds = tile_lds_bytes<T, N, OP, LB>();
    static_assert(lds <= 64 * 1024, "row-wave tile must fit the default dynamic LDS limit");
    if (n == 0) return NFM_OK;
    const int64_t nblk = (n + MPB - 1) / MPB;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    hipLaunchKernelGGL((roww_kernel<T, N, OP, R, LB>), dim3((unsigned)nblk), dim3(256 / R), lds,
                       static_cast<hipStream_t>(stream), static_cast<const T *>(a), static_cast<const T *>(b),
                       static_cast<T *>(o), n, p);
    return launch_status();
}

constexpr int rww_of(int op)
{
    return op == RW_SOLVE_SYM ? RWW_SOLVE : op == RW_INV_SYM ? RWW_INV_SYM : op == RW_INVDIAG_SYM ? RWW_INVDIAG_SYM
           : op == RW_DET_SYM ? RWW_DET_SYM : op == RW_INV_GEN ? RWW_INV_GEN : RWW_DET_GEN;
}

// form of the kernel: rowwave_choice (nfm_rowwave.hpp); NFM_ROWWAVE_ROWS (1 / 2 / 4) and
// NFM_ROWWAVE_LDS (0 / 1) override it for the side-by-side measurements
template <typename T, int N, int OP>
static int launch(const void *a, const void *b, void *o, int64_t n, const RowParams<T> &p, void *stream)
{
    static const int rows = [] { const char *e = dbg_env("NFM_ROWWAVE_ROWS"); return e ? atoi(e) : 0; }();
    static const int ldsb = [] { const char *e = dbg_env("NFM_ROWWAVE_LDS"); return e ? atoi(e) : -1; }();
    constexpr RwChoice c = rowwave_choice(sizeof(T) == 8, N, rww_of(OP));
    const int r = rows ? rows : (c.rows ? c.rows : 2);
    const bool lb = ldsb >= 0 ? ldsb != 0 : c.lds;
    if (lb) {
        if (r == 1) return launch_r<T, N, OP, 1, true>(a, b, o, n, p, stream);
        if (r == 4) return launch_r<T, N, OP, 4, true>(a, b, o, n, p, stream);
        return launch_r<T, N, OP, 2, true>(a, b, o, n, p, stream);
    }
    if (r == 1) return launch_r<T, N, OP, 1, false>(a, b, o, n, p, stream);
    if (r == 4) return launch_r<T, N, OP, 4, false>(a, b, o, n, p, stream);
    return launch_r<T, N, OP, 2, false>(a, b, o, n, p, stream);
}

#define NFM_RW_SWITCH(Nexpr, ...)                                                                  \
    switch (Nexpr) {                                                                               \
    case 9: { constexpr int N = 9; __VA_ARGS__; } break;                                           \
    case 10: { constexpr int N = 10; __VA_ARGS__; } break;                                         \
    case 11: { constexpr int N = 11; __VA_ARGS__; } break;                                         \
    case 12: { constexpr int N = 12; __VA_ARGS__; } break;                                         \
    case 13: { constexpr int N = 13; __VA_ARGS__; } break;                                         \
    case 14: { constexpr int N = 14; __VA_ARGS__; } break;                                         \
    case 15: { constexpr int N = 15; __VA_ARGS__; } break;                                         \
    case 16: { constexpr int N = 16; __VA_ARGS__; } break;                                         \
    default: break;                                                                                \
    }

} // namespace roww

using namespace roww;

template <typename T>
int RowWave<T>::sym_solve(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                          const double *eps, void *stream)
{
    const int K = M * (M + 1) / 2;
    if (!rec_contig(mat, K, 1, K, sizeof(T)) || !rec_contig(vec, M, 1, M, sizeof(T)) ||
        !rec_contig(out, M, 1, M, sizeof(T)))
        return NFM_EFALLBACK_RW;
    RowParams<T> p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? (T)eps[i] : T(0);
    NFM_RW_SWITCH(M, return (launch<T, N, RW_SOLVE_SYM>(mat->ptr, vec->ptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::sym_invert(int M, int diag_only, int64_t ni, const nfm_operand *mat, const nfm_operand *out,
                           void *stream)
{
    const int K = M * (M + 1) / 2;
    const int RO = diag_only ? M : K;
    if (!rec_contig(mat, K, 1, K, sizeof(T)) || !rec_contig(out, RO, 1, RO, sizeof(T))) return NFM_EFALLBACK_RW;
    RowParams<T> p{};
    if (diag_only) {
        NFM_RW_SWITCH(M, return (launch<T, N, RW_INVDIAG_SYM>(mat->ptr, nullptr, out->ptr, ni, p, stream)))
    } else {
        NFM_RW_SWITCH(M, return (launch<T, N, RW_INV_SYM>(mat->ptr, nullptr, out->ptr, ni, p, stream)))
    }
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::sym_det(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    const int K = M * (M + 1) / 2;
    if (!rec_contig(mat, K, 1, K, sizeof(T)) || out == nullptr || out->ptr == nullptr || out->stride_inner != 1)
        return NFM_EFALLBACK_RW;
    RowParams<T> p{};
    NFM_RW_SWITCH(M, return (launch<T, N, RW_DET_SYM>(mat->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::batch_inv(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (!rec_contig(a, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)) || !rec_contig(out, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)))
        return NFM_EFALLBACK_RW;
    RowParams<T> p{};
    NFM_RW_SWITCH(Nn, return (launch<T, N, RW_INV_GEN>(a->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::batch_det(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (!rec_contig(a, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)) || out == nullptr || out->ptr == nullptr ||
        out->stride_inner != 1)
        return NFM_EFALLBACK_RW;
    RowParams<T> p{};
    NFM_RW_SWITCH(Nn, return (launch<T, N, RW_DET_GEN>(a->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

#if NFM_ROWW_F64
template struct RowWave<double>;
#else
template struct RowWave<float>;

bool rowwave_forced(bool f64, int N)
{
    static const int env64 = [] { const char *e = dbg_env("NFM_ROWWAVE_MIN_F64"); return e ? atoi(e) : 0; }();
    static const int env32 = [] { const char *e = dbg_env("NFM_ROWWAVE_MIN_F32"); return e ? atoi(e) : 0; }();
    const int m = f64 ? env64 : env32;
    return m > 0 && N >= m;
}
#endif

} // namespace nfm

// nfm_sym.hip -- compact-symmetric entry points (sym_solve / matvec / invert / det /
// to_full / outer / matmul) for orders 1..8 in registers; orders 9..16 are in
// nfm_big.hip.  See include/nfm_hip.h for the ABI and the reference lines replaced.
#include "nfm_sym_ops.hpp"
#include "nfm_big.hpp"
#include "nfm_large.hpp"
#include "nfm_rowwave.hpp"
#include "nfm_spd.hpp"

namespace nfm {

// ------------------------------------------------------------------- dispatch helpers
#define NFM_CASE_M(Mv, ...) \
    case Mv: {              \
        constexpr int M = Mv; \
        __VA_ARGS__;        \
    } break;

#define NFM_SWITCH_M8(Mexpr, ...)        \
    switch (Mexpr) {                     \
        NFM_CASE_M(1, __VA_ARGS__)       \
        NFM_CASE_M(2, __VA_ARGS__)       \
        NFM_CASE_M(3, __VA_ARGS__)       \
        NFM_CASE_M(4, __VA_ARGS__)       \
        NFM_CASE_M(5, __VA_ARGS__)       \
        NFM_CASE_M(6, __VA_ARGS__)       \
        NFM_CASE_M(7, __VA_ARGS__)       \
        NFM_CASE_M(8, __VA_ARGS__)       \
    default:                             \
        return NFM_ESIZE;                \
    }

// One matrix (per outer slab) solved against many vectors (M <= 8).  Closed forms (M <= 4): the cofactors and
// the determinant are derived once per lane and applied to V right-hand sides -- the same operations
// in the same order as SolveOp (sym_solve_prepare / sym_solve_apply are the two halves of
// sym_solve_closed), so the results are bit for bit the per-record kernel's, at 16 multiply-adds and
// 4 divisions per system instead of ~250 instructions: the kernel becomes a stream over the vectors.
template <typename T, int M, int V>
__global__ __launch_bounds__(256) void sym_solve_bcast_kernel(Opnd mat, Opnd vec, Opnd out, int64_t n_inner,
                                                              SolveParams p)
{
    constexpr int K = sym_k(M);
    using RV = Rec<1, M>;
    const int64_t o = blockIdx.y;
    const T *pm = reinterpret_cast<const T *>(mat.ptr) + o * mat.so;
    T a[K];
#pragma unroll
    for (int c = 0; c < K; ++c) a[c] = pm[c * mat.sc];
    if (p.has_eps) {
#pragma unroll
        for (int i = 0; i < M; ++i) a[i] += (T)p.eps[i];
    }
    const int64_t base = (int64_t)blockIdx.x * (256 * V) + threadIdx.x;
    T v[V][M];
#pragma unroll
    for (int q = 0; q < V; ++q) rec_direct_load<T, RV>(vec, vec.tiled, o, base + q * 256, base + q * 256 < n_inner, v[q]);
    if constexpr (M <= 4) {
        T co[SymCofLen<M>::value], det;
        sym_solve_prepare<T, M>(a, co, det);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            T r[M];
            sym_solve_apply<T, M>(co, det, v[q], r);
            rec_direct_store<T, RV>(out, out.tiled, o, base + q * 256, base + q * 256 < n_inner, r);
        }
    } else {
        // orders 5..8: the elimination with partial pivoting of SolveOp, once, with the V vectors as the columns
        // of one right-hand side (ge_solve treats every column independently: the same bits per system)
        T f[M][M], b[M][V];
        sym_expand<T, M>(a, f);
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int q = 0; q < V; ++q) b[i][q] = v[q][i];
        ge_solve<T, M, V>(f, b);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            T r[M];
#pragma unroll
            for (int i = 0; i < M; ++i) r[i] = b[i][q];
            rec_direct_store<T, RV>(out, out.tiled, o, base + q * 256, base + q * 256 < n_inner, r);
        }
    }
}

template <typename T, int M>
static int sym_solve_bcast(int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                           const nfm_operand *out, const SolveParams &p, void *stream)
{
    constexpr int V = 4;
    const int64_t nblk = (ni + 256 * V - 1) / (256 * V);
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    auto mode = [](const nfm_operand *op) {
        return packed_ok(op, M, 1, M, sizeof(T)) ? (int)MODE_PACKED : (int)MODE_STRIDED;
    };
    hipLaunchKernelGGL((sym_solve_bcast_kernel<T, M, V>), dim3((unsigned)nblk, (unsigned)no, 1), dim3(256, 1, 1), 0,
                       static_cast<hipStream_t>(stream), make_opnd(mat, MODE_STRIDED), make_opnd(vec, mode(vec)),
                       make_opnd(out, mode(out)), ni, p);
    return launch_status();
}

// One matrix (per outer slab) against many vectors at orders 9..16 (`mat` broadcast along the inner batch level:
// one Hessian, a field of gradients -- `_impl/sym.py:371` broadcasts the batch dims of `mat` and `vec`).
// The workgroup factors the matrix ONCE -- LU with partial pivoting, element (i, j) in the register of thread
// 16 i + j, rows and columns of a step exchanged through LDS -- and leaves L, U, 1 / diag(U) and the row
// permutation in LDS; then every lane streams vectors through them: the permuted right-hand side is fetched
// component by component (any strides), forward and back substitution run on registers with the factor
// entries read as LDS broadcasts (one read serves the V vectors of a lane), 16-byte stores when the output
// records are contiguous.  M^2 multiply-adds per vector against 2 M elements moved: the kernel is a stream
// over vec / out.  (The LDS-resident fallback this replaces redid the whole elimination per vector.)
// MATVEC: y = (inp +-) A v with the same streaming part and no factorisation.
enum { BB_SOLVE = 0, BB_MATVEC = 1 };
template <typename T, int OP, int V>
__global__ __launch_bounds__(256) void sym_bcast_big_kernel(Opnd mat, Opnd vec, Opnd inp, Opnd out, int64_t n_inner,
                                                            int M, int kind, int mode, SolveParams p)
{
    constexpr int MX = NFM_MAX_DIM;
    __shared__ T lu[MX * MX];
    __shared__ T rdiag[MX];
    __shared__ int perm[MX];
    const int tid = threadIdx.x;
    const int64_t o = blockIdx.y;
    const T *pm = reinterpret_cast<const T *>(mat.ptr) + o * mat.so;
    // the full matrix: thread (i, j) = (tid / 16, tid % 16) owns element (i, j)
    const int ti = tid >> 4, tj = tid & 15;
    const bool in = ti < M && tj < M;
    T a = T(0);
    if (in) {
        a = kind == NFM_MAT_FULL ? pm[ti * mat.sr + tj * mat.sc] : pm[sym_idx(M, ti, tj) * mat.sc];
        if (OP == BB_SOLVE && p.has_eps && ti == tj) a += (T)p.eps[ti];
    }
    if constexpr (OP == BB_SOLVE) {
        // LU with partial pivoting by ONE wavefront, without a workgroup barrier per step (round 3's first version
        // kept element (i, j) in thread 16 i + j and went through LDS and two barriers per step: a third of the
        // kernel's time at 4e6 vectors).  Lane l of wavefront 0 holds row l / 4, columns 4 (l % 4) .. + 3 in four
        // registers; a step is: argmax of column k over the rows >= k (shuffles), the two rows change places
        // (ds_bpermute), the pivot row reaches every row (ds_bpermute) and the multiplier its four lanes
        // (quad broadcast), one fused update per register.  Then the factors go to LDS once.
        lu[ti * MX + tj] = a;
        if (tid < MX) perm[tid] = tid;
        __syncthreads();
        if (tid < 64) {
            const int lane = tid, ri = lane >> 2, g = lane & 3;
            T r4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) r4[c] = lu[ri * MX + 4 * g + c];
            // (a run-time loop: unrolled 16 times the factorisation held 170 registers and halved the occupancy of
            // the streaming part below, which is where the time goes)
#pragma unroll 1
            for (int k = 0; k < M; ++k) {
                const int kg = k >> 2, kc = k & 3; // column k lives in register kc of the lanes with g == kg
                const T mine_k = kc == 0 ? r4[0] : kc == 1 ? r4[1] : kc == 2 ? r4[2] : r4[3];
                // pivot: first row >= k with the largest |a_rk|
                T best = (g == kg && ri >= k && ri < M) ? fabs_(mine_k) : T(-1);
                int brow = ri;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const T ob = __shfl_xor(best, off, 64);
                    const int orow = __shfl_xor(brow, off, 64);
                    const bool take = ob > best || (ob == best && orow < brow);
                    best = take ? ob : best;
                    brow = take ? orow : brow;
                }
                int pr = __builtin_amdgcn_readfirstlane(brow);
                pr = (pr < k || pr >= M) ? k : pr; // (a column of NaNs compares false everywhere: keep row k)
                // rows k and pr change places
                const int partner = ri == k ? pr : (ri == pr ? k : ri);
#pragma unroll
                for (int c = 0; c < 4; ++c) r4[c] = __shfl(r4[c], partner * 4 + g, 64);
                if (lane == 0 && pr != k) { const int t = perm[k]; perm[k] = perm[pr]; perm[pr] = t; }
                // the pivot row for my columns, the pivot, my row's entry in column k
                T prow[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) prow[c] = __shfl(r4[c], k * 4 + g, 64);
                const T colk = kc == 0 ? r4[0] : kc == 1 ? r4[1] : kc == 2 ? r4[2] : r4[3];
                const T piv = __shfl(colk, k * 4 + kg, 64);
                const T aik = __shfl(colk, ri * 4 + kg, 64);
                if (ri > k && ri < M) {
                    const T l = aik / piv;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int col = 4 * g + c;
                        if (col == k) r4[c] = l;
                        else if (col > k) r4[c] = r4[c] - l * prow[c];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) lu[ri * MX + 4 * g + c] = r4[c];
        }
        __syncthreads();
        if (tid < M) rdiag[tid] = T(1) / lu[tid * MX + tid];
    } else {
        lu[ti * MX + tj] = a;
    }
    __syncthreads();

    const T *pv0 = reinterpret_cast<const T *>(vec.ptr) + o * vec.so;
    const T *pi0 = inp.ptr ? reinterpret_cast<const T *>(inp.ptr) + o * inp.so : nullptr;
    T *po0 = reinterpret_cast<T *>(out.ptr) + o * out.so;
    const bool vstore = out.sc == 1 && out.si % (16 / (int)sizeof(T)) == 0 && (reinterpret_cast<uintptr_t>(po0) & 15) == 0;
    // contiguous vector records are fetched with 16-byte loads; the row permutation of the factorisation is then
    // applied through an LDS image of the workgroup's vectors ([component][lane]: conflict-free, and the permuted
    // component index is the same in every lane) -- fetching component perm[i] of every vector straight from
    // global memory costs M scattered 4-byte loads per vector and held the kernel at 2.1-2.6 TB/s
    __shared__ T stage[(OP == BB_SOLVE ? MX : 1) * 256];
    const bool vload = vec.sc == 1;
    // the vectors of a tile as they sit in memory (16-byte loads when the records are contiguous)
    auto fetch = [&](int64_t tile, T (&xx)[V][MX]) {
        const int64_t base = tile * (256 * V) + tid;
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const int64_t n = base + q * 256;
            const T *pv = pv0 + (n < n_inner ? n : n_inner - 1) * vec.si;
            if (vload) {
                using VGl = typename VecOf<T>::gtype;
                constexpr int NVl = VecOf<T>::N;
#pragma unroll
                for (int i = 0; i < MX; i += NVl)
                    if (i < M) {
                        if (i + NVl <= M) {
                            const VGl v = *reinterpret_cast<const VGl *>(pv + i);
#pragma unroll
                            for (int c = 0; c < NVl; ++c) xx[q][i + c] = v[c];
                        } else {
#pragma unroll
                            for (int c = 0; c < NVl; ++c)
                                if (i + c < M) xx[q][i + c] = pv[i + c];
                        }
                    }
            } else {
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < M) xx[q][i] = pv[i * vec.sc];
            }
        }
    };
    // software pipeline: the next tile's vectors are on their way while this tile is solved (two wavefronts per
    // SIMD do not hide the latency of HBM on their own)
    T xnext[V][MX];
    if ((int64_t)blockIdx.x * (256 * V) < n_inner) fetch(blockIdx.x, xnext);
    for (int64_t tile = blockIdx.x; tile * (256 * V) < n_inner; tile += gridDim.x) {
        // the factors are re-read from LDS in every tile (broadcast reads, cheap): hoisted out of this loop, the 256
        // entries would take the whole register file and leave one wavefront per SIMD to a streaming kernel
        asm volatile("" ::: "memory");
        const int64_t base = tile * (256 * V) + tid;
        T x[V][MX];
#pragma unroll
        for (int q = 0; q < V; ++q)
#pragma unroll
            for (int i = 0; i < MX; ++i) x[q][i] = xnext[q][i];
        if ((tile + gridDim.x) * (256 * V) < n_inner) fetch(tile + gridDim.x, xnext);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            if constexpr (OP == BB_SOLVE) { // x <- P x through the LDS image (each lane reads back its own column)
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < M) stage[i * 256 + tid] = x[q][i];
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < M) x[q][i] = stage[perm[i] * 256 + tid];
            }
        }
        if constexpr (OP == BB_SOLVE) {
            // L y = P v (unit lower), then U x = y
#pragma unroll
            for (int i = 1; i < MX; ++i)
                if (i < M) {
#pragma unroll
                    for (int j = 0; j < i; ++j) {
                        const T l = lu[i * MX + j];
#pragma unroll
                        for (int q = 0; q < V; ++q) x[q][i] = fma_(-l, x[q][j], x[q][i]);
                    }
                }
#pragma unroll
            for (int i = MX - 1; i >= 0; --i)
                if (i < M) {
#pragma unroll
                    for (int j = i + 1; j < MX; ++j)
                        if (j < M) {
                            const T u = lu[i * MX + j];
#pragma unroll
                            for (int q = 0; q < V; ++q) x[q][i] = fma_(-u, x[q][j], x[q][i]);
                        }
                    const T r = rdiag[i];
#pragma unroll
                    for (int q = 0; q < V; ++q) x[q][i] *= r;
                }
        } else {
            T y[V][MX];
#pragma unroll
            for (int i = 0; i < MX; ++i)
                if (i < M) {
#pragma unroll
                    for (int q = 0; q < V; ++q) y[q][i] = T(0);
#pragma unroll
                    for (int j = 0; j < MX; ++j)
                        if (j < M) {
                            const T aij = lu[i * MX + j];
#pragma unroll
                            for (int q = 0; q < V; ++q) y[q][i] = fma_(aij, x[q][j], y[q][i]);
                        }
                }
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const int64_t n = base + q * 256;
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < M) {
                        T r = y[q][i];
                        if (mode != 0) {
                            const T b = pi0[(n < n_inner ? n : n_inner - 1) * inp.si + i * inp.sc];
                            r = mode > 0 ? b + r : b - r;
                        }
                        x[q][i] = r;
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const int64_t n = base + q * 256;
            if (n >= n_inner) continue;
            T *po = po0 + n * out.si;
            if (vstore) {
                using VG = typename VecOf<T>::type;
                constexpr int NV = VecOf<T>::N;
#pragma unroll
                for (int i = 0; i < MX; i += NV)
                    if (i < M) {
                        if (i + NV <= M) {
                            VG v;
#pragma unroll
                            for (int c = 0; c < NV; ++c) v[c] = x[q][i + c];
                            *reinterpret_cast<VG *>(po + i) = v;
                        } else {
#pragma unroll
                            for (int c = 0; c < NV; ++c)
                                if (i + c < M) po[i + c] = x[q][i + c];
                        }
                    }
            } else {
#pragma unroll
                for (int i = 0; i < MX; ++i)
                    if (i < M) po[i * out.sc] = x[q][i];
            }
        }
    }
}

template <typename T, int OP>
static int sym_bcast_big(int M, int kind, int mode, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                         const nfm_operand *inp, const nfm_operand *out, const SolveParams &p, void *stream)
{
    constexpr int V = sizeof(T) == 4 ? 4 : 2;
    int64_t nblk = (ni + 256 * V - 1) / (256 * V);
    const int64_t cap = (512 + no - 1) / no; // the factorisation is paid once per workgroup: two per CU (150-170 registers:
                                             // two wavefronts per SIMD), all resident at once -- a second round of
                                             // workgroups would pay it again -- each streams many tiles
    if (nblk > cap) nblk = cap;
    if (no > 65535) return NFM_ESIZE;
    nfm_operand none = {nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL((sym_bcast_big_kernel<T, OP, V>), dim3((unsigned)nblk, (unsigned)no, 1), dim3(256, 1, 1), 0,
                       static_cast<hipStream_t>(stream), make_opnd(mat, MODE_STRIDED), make_opnd(vec, MODE_STRIDED),
                       make_opnd(inp ? inp : &none, MODE_STRIDED), make_opnd(out, MODE_STRIDED), ni, M, kind, mode, p);
    return launch_status();
}

template <typename T>
static int sym_solve_t(int M, int kind, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                       const nfm_operand *out, const SolveParams &p, void *stream, bool pivoted = false)
{
    // a matrix that is the same along the inner batch level (stride 0: one Hessian, many gradients)
    if (kind == NFM_MAT_SYM && M >= 2 && M <= 8 && mat->stride_inner == 0 && ni >= 1024) {
        switch (M) {
        case 2: return sym_solve_bcast<T, 2>(no, ni, mat, vec, out, p, stream);
        case 3: return sym_solve_bcast<T, 3>(no, ni, mat, vec, out, p, stream);
        case 4: return sym_solve_bcast<T, 4>(no, ni, mat, vec, out, p, stream);
        case 5: return sym_solve_bcast<T, 5>(no, ni, mat, vec, out, p, stream);
        case 6: return sym_solve_bcast<T, 6>(no, ni, mat, vec, out, p, stream);
        case 7: return sym_solve_bcast<T, 7>(no, ni, mat, vec, out, p, stream);
        default: return sym_solve_bcast<T, 8>(no, ni, mat, vec, out, p, stream);
        }
    }
    if (M > 8 && (kind == NFM_MAT_SYM || kind == NFM_MAT_FULL) && mat->stride_inner == 0 && ni >= 1024)
        return sym_bcast_big<T, BB_SOLVE>(M, kind, 0, no, ni, mat, vec, nullptr, out, p, stream);
    if (M > 8) {
        if (kind == NFM_MAT_SYM && no == 1) { // contiguous operands: registers; else LDS-resident
            if (!pivoted) { // positive definite first, pivoted elimination for the groups that need it (nfm_spd.hip)
                const int rc = Spd<T>::sym_solve(M, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
                if (rc != NFM_EFALLBACK) return rc;
            }
            if (rowwave_first<T>(M, RWW_SOLVE)) { // one matrix per 16 lanes (nfm_rowwave.hip)
                const int rc = RowWave<T>::sym_solve(M, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
                if (rc != NFM_EFALLBACK) return rc;
            }
            const int rc = Large<T>::sym_solve(M, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (kind == NFM_MAT_SYM && !pivoted) { // any strides, two batch levels: every lane addresses its own record (nfm_spd.hip)
            const int rc = Spd<T>::sym_solve_strided(M, no, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_solve<T>(M, kind, no, ni, mat, vec, out, p.has_eps ? p.eps : nullptr, stream);
    }
    switch (kind) {
    case NFM_MAT_SYM:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_SYM>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    case NFM_MAT_DIAG:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_DIAG>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    case NFM_MAT_SCAL:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_SCAL>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    case NFM_MAT_FULL:
        NFM_SWITCH_M8(M, return (rec_launch<T, SolveOp<T, M, NFM_MAT_FULL>>(mat, vec, nullptr, out, no, ni, p, stream)))
        break;
    default:
        return NFM_EINVAL;
    }
    return NFM_EINVAL;
}

template <typename T>
static int sym_matvec_t(int M, int kind, int mode, int64_t no, int64_t ni, const nfm_operand *mat,
                        const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out, void *stream)
{
    if (M > 8 && (kind == NFM_MAT_SYM || kind == NFM_MAT_FULL) && mat->stride_inner == 0 && ni >= 1024) {
        SolveParams sp{};
        return sym_bcast_big<T, BB_MATVEC>(M, kind, mode, no, ni, mat, vec, inp, out, sp, stream);
    }
    if (M > 8) {
        if (kind == NFM_MAT_SYM && no == 1) {
            { // float64 15, 16: the records through LDS images (nfm_spd.hip)
                const int rc = Spd<T>::sym_matvec(M, mode, ni, mat, vec, inp, out, stream);
                if (rc != NFM_EFALLBACK) return rc;
            }
            const int rc = Large<T>::sym_matvec(M, mode, ni, mat, vec, inp, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (kind == NFM_MAT_SYM) { // any strides, two batch levels: every lane addresses its own record (nfm_spd.hip)
            const int rc = Spd<T>::sym_matvec_strided(M, mode, no, ni, mat, vec, inp, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_matvec<T>(M, kind, mode, no, ni, mat, vec, inp, out, stream);
    }
    MatvecParams p{mode};
    switch (kind) {
    case NFM_MAT_SYM:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_SYM>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    case NFM_MAT_DIAG:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_DIAG>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    case NFM_MAT_SCAL:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_SCAL>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    case NFM_MAT_FULL:
        NFM_SWITCH_M8(M, return (rec_launch<T, MatvecOp<T, M, NFM_MAT_FULL>>(mat, vec, inp, out, no, ni, p, stream)))
        break;
    default:
        return NFM_EINVAL;
    }
    return NFM_EINVAL;
}

template <typename T>
static int sym_invert_t(int M, int flags, int64_t no, int64_t ni, const nfm_operand *mat,
                        const nfm_operand *out, void *stream)
{
    const int diag_only = flags & NFM_INVERT_DIAG;
    const bool pivoted = (flags & NFM_INVERT_PIVOTED) != 0;
    if (M > 8) {
        if (no == 1 && !pivoted) { // positive definite first (nfm_spd.hip)
            const int rc = Spd<T>::sym_invert(M, diag_only, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1 && rowwave_first<T>(M, diag_only ? RWW_INVDIAG_SYM : RWW_INV_SYM)) {
            const int rc = RowWave<T>::sym_invert(M, diag_only, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (!diag_only && no == 1) {
            const int rc = Large<T>::sym_invert(M, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (!pivoted) { // any strides, two batch levels (nfm_spd.hip)
            const int rc = Spd<T>::sym_invert_strided(M, diag_only, no, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_invert<T>(M, diag_only, no, ni, mat, out, stream);
    }
    NoParams p{0};
    if (diag_only) {
        NFM_SWITCH_M8(M, return (rec_launch<T, InvertOp<T, M, true>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    } else {
        NFM_SWITCH_M8(M, return (rec_launch<T, InvertOp<T, M, false>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    }
    return NFM_EINVAL;
}

template <typename T>
static int sym_det_t(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    if (M > 8) {
        if (no == 1) { // positive definite first (nfm_spd.hip)
            const int rc = Spd<T>::sym_det(M, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1 && rowwave_first<T>(M, RWW_DET_SYM)) {
            const int rc = RowWave<T>::sym_det(M, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        if (no == 1) {
            const int rc = Large<T>::sym_det(M, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        { // any strides, two batch levels (nfm_spd.hip)
            const int rc = Spd<T>::sym_det_strided(M, no, ni, mat, out, stream);
            if (rc != NFM_EFALLBACK) return rc;
        }
        return big_sym_det<T>(M, no, ni, mat, out, stream);
    }
    NoParams p{0};
    NFM_SWITCH_M8(M, return (rec_launch<T, DetOp<T, M>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int sym_to_full_t(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out,
                         void *stream)
{
    if (M > 8) return big_sym_to_full<T>(M, no, ni, mat, out, stream);
    NoParams p{0};
    NFM_SWITCH_M8(M, return (rec_launch<T, ToFullOp<T, M>>(mat, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int sym_outer_t(int M, int64_t no, int64_t ni, const nfm_operand *x, const nfm_operand *out, void *stream)
{
    if (M > 8) return big_sym_outer<T>(M, no, ni, x, out, stream);
    NoParams p{0};
    NFM_SWITCH_M8(M, return (rec_launch<T, OuterOp<T, M>>(x, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int sym_outer2_t(int M, int neg, int64_t no, int64_t ni, const nfm_operand *x, const nfm_operand *y,
                        const nfm_operand *out, void *stream)
{
    if (M > 8) return big_sym_outer2<T>(M, neg, no, ni, x, y, out, stream);
    Outer2Params p{neg};
    NFM_SWITCH_M8(M, return (rec_launch<T, Outer2Op<T, M>>(x, y, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T, int HK>
static int sym_matmul_t(int K, int D, int64_t no, int64_t ni, const nfm_operand *jac, const nfm_operand *hess,
                        const nfm_operand *out, void *stream)
{
    NoParams p{0};
#define NFM_MM(Kv, Dv)                                                                              \
    if (K == Kv && D == Dv)                                                                         \
        return (rec_launch<T, MatmulOp<T, Kv, Dv, HK>>(jac, hess, nullptr, out, no, ni, p, stream));
    NFM_MM(1, 1) NFM_MM(1, 2) NFM_MM(1, 3) NFM_MM(1, 4)
    NFM_MM(2, 1) NFM_MM(2, 2) NFM_MM(2, 3) NFM_MM(2, 4)
    NFM_MM(3, 1) NFM_MM(3, 2) NFM_MM(3, 3) NFM_MM(3, 4)
    NFM_MM(4, 1) NFM_MM(4, 2) NFM_MM(4, 3) NFM_MM(4, 4)
#undef NFM_MM
    return big_sym_matmul<T>(K, D, HK, no, ni, jac, hess, out, stream);
}

} // namespace nfm

using namespace nfm;

extern "C" {

int nfm_sym_solve(int dtype, int M, int mat_kind, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                  const nfm_operand *vec, const nfm_operand *out, const double *eps, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool pivoted = (mat_kind & NFM_MAT_PIVOTED) != 0;
    mat_kind &= ~NFM_MAT_PIVOTED;
    if (mat_kind < 0 || mat_kind > 3) return NFM_EINVAL;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(vec, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    SolveParams p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? eps[i] : 0.0;
    return dtype == NFM_F32 ? sym_solve_t<float>(M, mat_kind, n_outer, n_inner, mat, vec, out, p, stream, pivoted)
                            : sym_solve_t<double>(M, mat_kind, n_outer, n_inner, mat, vec, out, p, stream, pivoted);
}

int nfm_sym_matvec(int dtype, int M, int mat_kind, int mode, int64_t n_outer, int64_t n_inner,
                   const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *inp, const nfm_operand *out,
                   void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    if (mat_kind < 0 || mat_kind > 3 || mode < -1 || mode > 1) return NFM_EINVAL;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(vec, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    if (mode != 0 && (rc = check_operand(inp, dtype, nonempty))) return rc;
    if (mode == 0) inp = nullptr;
    return dtype == NFM_F32 ? sym_matvec_t<float>(M, mat_kind, mode, n_outer, n_inner, mat, vec, inp, out, stream)
                            : sym_matvec_t<double>(M, mat_kind, mode, n_outer, n_inner, mat, vec, inp, out, stream);
}

int nfm_sym_invert(int dtype, int M, int diag_only, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                   const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_invert_t<float>(M, diag_only, n_outer, n_inner, mat, out, stream)
                            : sym_invert_t<double>(M, diag_only, n_outer, n_inner, mat, out, stream);
}

int nfm_sym_det(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *mat, const nfm_operand *out,
                void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_det_t<float>(M, n_outer, n_inner, mat, out, stream)
                            : sym_det_t<double>(M, n_outer, n_inner, mat, out, stream);
}

int nfm_sym_to_full(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *mat,
                    const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(mat, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_to_full_t<float>(M, n_outer, n_inner, mat, out, stream)
                            : sym_to_full_t<double>(M, n_outer, n_inner, mat, out, stream);
}

int nfm_sym_outer(int dtype, int M, int64_t n_outer, int64_t n_inner, const nfm_operand *x, const nfm_operand *out,
                  void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(x, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_outer_t<float>(M, n_outer, n_inner, x, out, stream)
                            : sym_outer_t<double>(M, n_outer, n_inner, x, out, stream);
}

int nfm_sym_outer2(int dtype, int M, int neg, int64_t n_outer, int64_t n_inner, const nfm_operand *x,
                   const nfm_operand *y, const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (M < 1 || M > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(x, dtype, nonempty))) return rc;
    if ((rc = check_operand(y, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? sym_outer2_t<float>(M, neg ? 1 : 0, n_outer, n_inner, x, y, out, stream)
                            : sym_outer2_t<double>(M, neg ? 1 : 0, n_outer, n_inner, x, y, out, stream);
}

int nfm_sym_matmul(int dtype, int K, int D, int hess_kind, int64_t n_outer, int64_t n_inner,
                   const nfm_operand *jac, const nfm_operand *hess, const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (K < 1 || K > NFM_MAX_DIM || D < 1 || D > NFM_MAX_DIM) return NFM_ESIZE;
    if (hess_kind != NFM_MAT_SYM && hess_kind != NFM_MAT_DIAG) return NFM_EINVAL;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(jac, dtype, nonempty))) return rc;
    if ((rc = check_operand(hess, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    if (dtype == NFM_F32)
        return hess_kind == NFM_MAT_SYM
                   ? sym_matmul_t<float, NFM_MAT_SYM>(K, D, n_outer, n_inner, jac, hess, out, stream)
                   : sym_matmul_t<float, NFM_MAT_DIAG>(K, D, n_outer, n_inner, jac, hess, out, stream);
    return hess_kind == NFM_MAT_SYM
               ? sym_matmul_t<double, NFM_MAT_SYM>(K, D, n_outer, n_inner, jac, hess, out, stream)
               : sym_matmul_t<double, NFM_MAT_DIAG>(K, D, n_outer, n_inner, jac, hess, out, stream);
}

} // extern "C"

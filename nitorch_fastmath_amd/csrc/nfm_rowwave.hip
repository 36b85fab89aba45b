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
#include <stdlib.h>
#include "nfm_common.hpp"
#include "nfm_smallmat.hpp"
#include "nfm_rowwave.hpp"

namespace nfm {
namespace roww {

constexpr int MPB = 16; // matrices per workgroup

// The measurement knobs of this file (scripts/bench_rowwave.py: force a form, force the row-wave kernels from an
// order up) are read only when NFM_DEBUG is set in the environment: the product's dispatch depends on ONE variable.
static const char *dbg_env(const char *name)
{
    static const bool on = getenv("NFM_DEBUG") != nullptr;
    return on ? getenv(name) : nullptr;
}

enum { RW_SOLVE_SYM = 0, RW_INV_SYM, RW_INVDIAG_SYM, RW_DET_SYM, RW_INV_GEN, RW_DET_GEN };

struct RowParams {
    int has_eps;
    double eps[NFM_MAX_DIM];
};

// row stride of the N x N LDS image in elements: a whole, odd number of 16-byte slots
template <typename T, int N>
struct RowStride {
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int slots = (N + V - 1) / V;
    static constexpr int value = ((slots & 1) ? slots : slots + 1) * V;
};

__device__ __forceinline__ int bperm(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float bcast(float v, int src_lane) { return __int_as_float(bperm(src_lane, __float_as_int(v))); }
__device__ __forceinline__ double bcast(double v, int src_lane)
{
    const int lo = bperm(src_lane, __double2loint(v)), hi = bperm(src_lane, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int ROR>
__device__ __forceinline__ int dpp_ror(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + ROR, 0xf, 0xf, false); // row_ror:ROR
}
// Pivot search key: an unsigned integer that orders |x| -- float32: the bits of |x|; float64: the
// high dword of |x| (sign cleared: 11 exponent + 20 mantissa bits, the whole exponent range at a
// resolution of 2^-20; rows whose |a_rk| agree to 1e-6 tie and the first one pivots, which is as
// good a partial pivot).  +1 so that 0 is left for rows that may not pivot.  A NaN has the
// largest key: it pivots, and the result is NaN as it would be anyway.
__device__ __forceinline__ unsigned pivot_key(float x) { return ((unsigned)__float_as_int(x) & 0x7fffffffu) + 1u; }
__device__ __forceinline__ unsigned pivot_key(double x) { return ((unsigned)__double2hiint(x) & 0x7fffffffu) + 1u; }
template <int ROR>
__device__ __forceinline__ unsigned rowmax_step(unsigned v)
{
    const unsigned o = (unsigned)dpp_ror<ROR>((int)v);
    return v > o ? v : o;
}
__device__ __forceinline__ float recip(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double recip(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
// 1 / pivot = v_rcp + Newton steps; a zero / infinite / NaN pivot keeps the raw v_rcp (inf / 0 / NaN, what a
// division gives; the Newton step would turn inf and 0 into NaN).  Branch-free -- a select on the class of the
// raw reciprocal -- so that the elimination stays ONE basic block and the scheduler can start the pivot search
// of step k+1 (column k+1 is updated first) under the row updates of step k.  (Round 2 took the IEEE division
// on a wavefront vote here: a branch per step.)
__device__ __forceinline__ float pivot_recip(float pv)
{
    const float r0 = __builtin_amdgcn_rcpf(pv);
    const float r = __builtin_fmaf(__builtin_fmaf(-pv, r0, 1.0f), r0, r0);
    return __builtin_amdgcn_classf(r0, 0x267) ? r0 : r; // NaN (0x3), -inf (0x4), zeros (0x60), +inf (0x200): keep the raw value
}
__device__ __forceinline__ double pivot_recip(double pv)
{
    const double r0 = __builtin_amdgcn_rcp(pv);
    double r = __builtin_fma(__builtin_fma(-pv, r0, 1.0), r0, r0);
    r = __builtin_fma(__builtin_fma(-pv, r, 1.0), r, r);
    return __builtin_amdgcn_class(r0, 0x267) ? r0 : r;
}
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// compact-sym index of (i, j), `sym.py:7-14`
__device__ __forceinline__ int cidx(int N, int i, int j)
{
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return i == j ? i : N + lo * N - (lo * (lo + 1)) / 2 + (hi - lo - 1);
}

// elements of the LDS image of a tile of MPB matrices (input records, reused for the output)
template <typename T, int N, int OP>
__host__ __device__ constexpr int img_elems()
{
    constexpr bool SYM = OP == RW_SOLVE_SYM || OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_DET_SYM;
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int raw = SYM ? MPB * (N * (N + 1) / 2) : MPB * N * RowStride<T, N>::value;
    return ((raw + V - 1) / V) * V;
}

// max over the G = 16 / R lanes of a matrix, in every one of them
template <int G>
__device__ __forceinline__ unsigned groupmax(unsigned v)
{
    if constexpr (G == 16) {
        v = rowmax_step<8>(v);
        v = rowmax_step<4>(v);
        v = rowmax_step<2>(v);
        return rowmax_step<1>(v);
    } else {
        auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
        v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false)); // quad_perm [1,0,3,2]
        v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false)); // quad_perm [2,3,0,1]
        if constexpr (G == 8) v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false)); // row_half_mirror
        return v;
    }
}

// R rows per lane, G = 16 / R lanes per matrix, 16 matrices per workgroup of 16 * G lanes.
// The pivot row is broadcast once per wavefront whatever R is (ds_bpermute runs at one
// wave-instruction per ~6.5 clocks per CU and is what bounds the R = 1 form), so R = 2 / 4
// spread that cost over 2x / 4x the matrices, for R x the fma work and registers per lane.
//
// LB: how the pivot row reaches the other lanes.  false: ds_bpermute, one per dword.  true: the
// pivot lane writes the row into a per-matrix LDS slot (ds_write_b128, one lane per matrix active)
// and every lane reads it back (ds_read_b128 of one address per matrix: a broadcast read, 16 bytes
// per lane per instruction) -- LDS operations of a wavefront execute in order, so no barrier.
template <typename T, int N, int OP, int R, bool LB>
__global__ __launch_bounds__(256 / R) void roww_kernel(const T *__restrict__ A, const T *__restrict__ B,
                                                       T *__restrict__ O, int64_t n, RowParams p)
{
    constexpr bool SYM = OP == RW_SOLVE_SYM || OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_DET_SYM;
    constexpr bool INV = OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_INV_GEN;
    constexpr bool DET = OP == RW_DET_SYM || OP == RW_DET_GEN;
    constexpr int G = 16 / R;             // lanes per matrix
    constexpr int NT = MPB * G;           // lanes per workgroup
    constexpr int K = N * (N + 1) / 2;
    constexpr int RIN = SYM ? K : N * N;                                    // input record
    constexpr int ROUT = OP == RW_SOLVE_SYM ? N : OP == RW_INV_SYM ? K : OP == RW_INVDIAG_SYM ? N : DET ? 1 : N * N;
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int RS = RowStride<T, N>::value;
    using Vec = T __attribute__((ext_vector_type(V)));
    constexpr int NS = ((N + 1 + V - 1) / V) * V; // pivot-row slot of a matrix: N values + the right-hand side
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *img = reinterpret_cast<T *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = tid / G, lr = tid % G;  // matrix of the tile, lane within the matrix
    T *slot = img + img_elems<T, N, OP>() + g * NS; // LB: this matrix's pivot-row slot
    const int gbase = lane & ~(G - 1);    // first lane of this matrix within the wavefront
    const unsigned gmask = (1u << G) - 1u;
    const int64_t m0 = (int64_t)blockIdx.x * MPB;
    const int nm = (int)((n - m0) < MPB ? (n - m0) : MPB);

    // ---- stream the tile's contiguous input records into LDS
    {
        const T *src = A + m0 * RIN;
        const int total = nm * RIN;
        // a tile starts a whole number of 16-matrix blocks into the operand: 16-byte aligned exactly
        // when the operand's base is (row slices x[i:] of float64 tensors may not be)
        const bool vec_ok = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        for (int e = tid * V; e < total; e += NT * V) {
            if (vec_ok && e + V <= total) {
                const Vec v = NFM_LDG(reinterpret_cast<const Vec *>(src + e));
                if constexpr (SYM) { // flat copy of the compact records
                    *reinterpret_cast<Vec *>(img + e) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) {
                        const int ee = e + q, m = ee / (N * N), rem = ee - m * (N * N), i = rem / N, j = rem - i * N;
                        img[(m * N + i) * RS + j] = v[q];
                    }
                }
            } else {
                for (int ee = e; ee < total && ee < e + V; ++ee) {
                    const T x = NFM_LDG(src + ee);
                    if constexpr (SYM) img[ee] = x;
                    else {
                        const int m = ee / (N * N), rem = ee - m * (N * N), i = rem / N, j = rem - i * N;
                        img[(m * N + i) * RS + j] = x;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- my rows: row id of slot t is lr + t * G
    T row[R][N];
    T rhs[R];
    bool used[R];
    int ppos[R];   // the step at which the row pivoted = the row of the result it ends up holding
    T mypv[R];     // inverse: the pivot of the row (rows are scaled once, at the end)
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int rid = lr + t * G;
        const bool live = rid < N && g < nm;
        if constexpr (SYM) {
#pragma unroll
            for (int j = 0; j < N; ++j) row[t][j] = live ? img[g * K + cidx(N, rid, j)] : T(0);
            if (OP == RW_SOLVE_SYM && p.has_eps) {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (rid == j) row[t][j] += (T)p.eps[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) row[t][j] = live ? img[(g * N + rid) * RS + j] : T(0);
        }
        rhs[t] = T(0);
        if constexpr (OP == RW_SOLVE_SYM) rhs[t] = live ? NFM_LDG(B + (m0 + g) * N + rid) : T(0);
        used[t] = !(rid < N); // rows beyond the order never pivot
        if (!(g < nm)) {      // idle matrices of a ragged last tile: the identity (nothing divides by zero)
#pragma unroll
            for (int j = 0; j < N; ++j) row[t][j] = (rid == j) ? T(1) : T(0);
        }
        ppos[t] = -1;
        mypv[t] = T(1);
    }
    int col_of[INV ? N : 1]; // inverse: row id of the pivot of every step (the column permutation)
    T det = T(1);
    int inversions = 0;

#pragma unroll
    for (int k = 0; k < N; ++k) {
        // -- pivot: first unused row with the largest |a_rk| (pivot_key above); slot first, then lane
        unsigned key = 0;
        int ts = 0; // my best slot
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const unsigned kt = used[t] ? 0u : pivot_key(row[t][k]);
            ts = kt > key ? t : ts;
            key = kt > key ? kt : key;
        }
        const unsigned mx = groupmax<G>(key);
        const unsigned grp = (unsigned)(__ballot(key == mx) >> gbase) & gmask; // never 0: some row is unused
        const int pl = __builtin_ctz(grp | (1u << G));
        const bool isl = lr == pl;      // my lane holds the pivot row, in slot ts
        const int psrc = gbase + pl;
        int prid = 0;                   // row id of the pivot row
        if constexpr (INV || DET) {
            prid = R == 1 ? pl : bperm(psrc, lr + ts * G);
            if constexpr (INV) col_of[k] = prid;
        }
        if constexpr (DET) { // parity of the row permutation: unused rows with a smaller row id
            unsigned un = 0;
#pragma unroll
            for (int t = 0; t < R; ++t) un |= ((unsigned)(__ballot(!used[t]) >> gbase) & gmask) << (t * G);
            inversions += __builtin_popcount(un & ((1u << prid) - 1u));
        }
        // the pivot row of my lane's candidate slot (only the pivot lane's values are consumed)
        T cand[N];
        T crhs = rhs[0];
#pragma unroll
        for (int j = 0; j < N; ++j) cand[j] = row[0][j];
#pragma unroll
        for (int t = 1; t < R; ++t) {
#pragma unroll
            for (int j = 0; j < N; ++j) cand[j] = (ts == t) ? row[t][j] : cand[j];
            crhs = (ts == t) ? rhs[t] : crhs;
        }
        if constexpr (LB) {
            if (isl) { // INV needs the whole row, the others columns k.. and the right-hand side
#pragma unroll
                for (int j = INV ? 0 : k; j < N; ++j) slot[j] = cand[j];
                if constexpr (OP == RW_SOLVE_SYM) slot[N] = crhs;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto pivot_row = [&](const T &mine, int j) -> T {
            if constexpr (LB) return slot[j];
            else return bcast(mine, psrc);
        };
        const T pv = pivot_row(cand[k], k);
        const T rp = pivot_recip(pv);
        // multipliers of my rows; 0 in the pivot row itself, so that the same fma leaves it unchanged
        // (a determinant with a zero pivot is 0 whatever follows: no elimination then)
        T f[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            f[t] = row[t][k] * rp;
            f[t] = (isl && ts == t) ? T(0) : f[t];
            if constexpr (DET) f[t] = (pv == T(0)) ? T(0) : f[t];
        }
        if constexpr (DET) det *= pv;
        if constexpr (INV) {
            // in-place Gauss-Jordan on [A | I] WITHOUT scaling the pivot row (rows are divided by their
            // pivots at the end): column k of A is spent, its slot takes the column of the right half
            // that becomes non-trivial in this step -- 1 in the pivot row, -f elsewhere
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j == k) continue;
                const T pj = pivot_row(cand[j], j);
#pragma unroll
                for (int t = 0; t < R; ++t) row[t][j] = fma_(-f[t], pj, row[t][j]);
            }
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const bool me = isl && ts == t;
                row[t][k] = me ? T(1) : -f[t];
                mypv[t] = me ? pv : mypv[t];
            }
        } else {
            // Gauss-Jordan on [A | b] (solve) or plain elimination (det): columns k+1.. only
#pragma unroll
            for (int j = k + 1; j < N; ++j) {
                const T pj = pivot_row(cand[j], j);
#pragma unroll
                for (int t = 0; t < R; ++t) row[t][j] = fma_(-f[t], pj, row[t][j]);
            }
            if constexpr (OP == RW_SOLVE_SYM) {
                const T pb = pivot_row(crhs, N);
#pragma unroll
                for (int t = 0; t < R; ++t) rhs[t] = fma_(-f[t], pb, rhs[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const bool me = isl && ts == t;
            used[t] = used[t] || me;
            ppos[t] = me ? k : ppos[t];
        }
        if constexpr (LB) { // the slot is rewritten by the next step only after every lane has read it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- results: slot t holds row ppos[t] of the result
    if constexpr (OP == RW_SOLVE_SYM) {
#pragma unroll
        for (int t = 0; t < R; ++t) {
            T piv = T(1); // the pivot is still at column ppos of the row
#pragma unroll
            for (int j = 0; j < N; ++j) piv = (ppos[t] == j) ? row[t][j] : piv;
            if (g < nm && ppos[t] >= 0) NFM_STG(rhs[t] / piv, O + (m0 + g) * N + ppos[t]);
        }
    } else if constexpr (DET) {
        const T d = (inversions & 1) ? -det : det;
        if (lr == 0 && g < nm) NFM_STG(d, O + (m0 + g));
    } else {
        // its l-th value is column col_of[l]
        __syncthreads(); // everyone is done reading the input image
#pragma unroll
        for (int t = 0; t < R; ++t) {
            if (!(g < nm && ppos[t] >= 0)) continue;
            const T rp = T(1) / mypv[t];
            const int pr = ppos[t];
            if constexpr (OP == RW_INV_GEN) {
#pragma unroll
                for (int l = 0; l < N; ++l) img[(g * N + pr) * RS + col_of[l]] = row[t][l] * rp;
            } else if constexpr (OP == RW_INV_SYM) {
#pragma unroll
                for (int l = 0; l < N; ++l)
                    if (col_of[l] >= pr) img[g * K + cidx(N, pr, col_of[l])] = row[t][l] * rp;
            } else { // diagonal only
#pragma unroll
                for (int l = 0; l < N; ++l)
                    if (col_of[l] == pr) img[g * N + pr] = row[t][l] * rp;
            }
        }
        __syncthreads();
        T *dst = O + m0 * ROUT;
        const int total = nm * ROUT;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
        for (int e = tid * V; e < total; e += NT * V) {
            T tmp[V];
            if constexpr (OP == RW_INV_GEN) {
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    const int ee = e + q, m = ee / (N * N), rem = ee - m * (N * N), i = rem / N, j = rem - i * N;
                    tmp[q] = (ee < total) ? img[(m * N + i) * RS + j] : T(0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < V; ++q) tmp[q] = (e + q < total) ? img[e + q] : T(0);
            }
            if (vec_ok && e + V <= total) {
                Vec v;
#pragma unroll
                for (int q = 0; q < V; ++q) v[q] = tmp[q];
                NFM_STG(v, reinterpret_cast<Vec *>(dst + e));
            } else {
#pragma unroll
                for (int q = 0; q < V; ++q)
                    if (e + q < total) NFM_STG(tmp[q], dst + e + q);
            }
        }
    }
}

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
static int launch_r(const void *a, const void *b, void *o, int64_t n, const RowParams &p, void *stream)
{
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int NS = ((N + 1 + V - 1) / V) * V;
    constexpr size_t lds = ((size_t)img_elems<T, N, OP>() + (LB ? MPB * NS : 0)) * sizeof(T);
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
static int launch(const void *a, const void *b, void *o, int64_t n, const RowParams &p, void *stream)
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
    RowParams p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? eps[i] : 0.0;
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
    RowParams p{};
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
    RowParams p{};
    NFM_RW_SWITCH(M, return (launch<T, N, RW_DET_SYM>(mat->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::batch_inv(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (!rec_contig(a, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)) || !rec_contig(out, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)))
        return NFM_EFALLBACK_RW;
    RowParams p{};
    NFM_RW_SWITCH(Nn, return (launch<T, N, RW_INV_GEN>(a->ptr, nullptr, out->ptr, ni, p, stream)))
    return NFM_EFALLBACK_RW;
}

template <typename T>
int RowWave<T>::batch_det(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (!rec_contig(a, (int64_t)Nn * Nn, Nn, Nn, sizeof(T)) || out == nullptr || out->ptr == nullptr ||
        out->stride_inner != 1)
        return NFM_EFALLBACK_RW;
    RowParams p{};
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

// nfm_rowwave_core.hpp -- the device side of the one-matrix-per-16-lanes kernels (nfm_rowwave.hip documents the
// algorithm): shared with nfm_spd.hip, whose wavefronts fall back to it.
#pragma once
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

// (eps in the dtype of the kernel: the kernel adds it straight from its scalar registers; converted from double on
// the device, the sixteen values became loop-invariant vector registers of the fallback loop of nfm_spd.hip)
template <typename T>
struct RowParams {
    int has_eps;
    T eps[NFM_MAX_DIM];
};

// element strides of the operands of a STRIDED tile (the fallback of the strided kernels of nfm_spd.hip; compact
// symmetric ops only): element c of matrix m at ptr[m * si + c * sc]
struct RowStrides {
    int64_t a_si, a_sc, b_si, b_sc, o_si, o_sc;
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
template <typename T, int N, int OP, int MPB_ = MPB>
__host__ __device__ constexpr int img_elems()
{
    constexpr bool SYM = OP == RW_SOLVE_SYM || OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_DET_SYM;
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int raw = SYM ? MPB_ * (N * (N + 1) / 2) : MPB_ * N * RowStride<T, N>::value;
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
// `roww_tile`: the tile of MPB_ matrices starting at matrix m0, worked on by the MPB_ * G lanes of the calling
// workgroup (all of them call it, `tid` = threadIdx.x); `smem`: img_elems + (LB ? MPB * NS : 0) elements of LDS.  The kernel below is
// this function on tile blockIdx.x; the positive-definite-first kernels of nfm_spd.hip call it (R = 4: one
// wavefront) for the wavefronts that met a matrix their unpivoted factorisation does not cover.
template <typename T, int N, int OP, int R, bool LB, int MPB_ = MPB, bool STR = false>
__device__ __forceinline__ void roww_tile(const T *__restrict__ A, const T *__restrict__ B, T *__restrict__ O, int64_t n,
                                          int64_t m0, const RowParams<T> &p, char *smem, const int tid,
                                          const RowStrides &st = RowStrides{})
{
    constexpr bool SYM = OP == RW_SOLVE_SYM || OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_DET_SYM;
    constexpr bool INV = OP == RW_INV_SYM || OP == RW_INVDIAG_SYM || OP == RW_INV_GEN;
    constexpr bool DET = OP == RW_DET_SYM || OP == RW_DET_GEN;
    constexpr int G = 16 / R;             // lanes per matrix
    constexpr int NT = MPB_ * G;          // lanes per workgroup
    constexpr int K = N * (N + 1) / 2;
    constexpr int RIN = SYM ? K : N * N;                                    // input record
    constexpr int ROUT = OP == RW_SOLVE_SYM ? N : OP == RW_INV_SYM ? K : OP == RW_INVDIAG_SYM ? N : DET ? 1 : N * N;
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int RS = RowStride<T, N>::value;
    using Vec = T __attribute__((ext_vector_type(V)));
    constexpr int NS = ((N + 1 + V - 1) / V) * V; // pivot-row slot of a matrix: N values + the right-hand side
    T *img = reinterpret_cast<T *>(smem);

    const int lane = tid & 63;
    const int g = tid / G, lr = tid % G;  // matrix of the tile, lane within the matrix
    T *slot = img + img_elems<T, N, OP, MPB_>() + g * NS; // LB: this matrix's pivot-row slot
    const int gbase = lane & ~(G - 1);    // first lane of this matrix within the wavefront
    const unsigned gmask = (1u << G) - 1u;
    const int nm = (int)((n - m0) < MPB_ ? (n - m0) : MPB_);

    // ---- stream the tile's contiguous input records into LDS (strided operands: element by element)
    if constexpr (STR) {
        static_assert(!STR || SYM, "strided tiles: compact symmetric ops only");
        const int total = nm * RIN;
        for (int e = tid; e < total; e += NT) {
            const int m = e / RIN, c = e - m * RIN;
            img[e] = A[(m0 + m) * st.a_si + c * st.a_sc];
        }
    } else {
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
                    if (rid == j) row[t][j] += p.eps[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) row[t][j] = live ? img[(g * N + rid) * RS + j] : T(0);
        }
        rhs[t] = T(0);
        if constexpr (OP == RW_SOLVE_SYM) {
            if constexpr (STR) rhs[t] = live ? B[(m0 + g) * st.b_si + rid * st.b_sc] : T(0);
            else rhs[t] = live ? NFM_LDG(B + (m0 + g) * N + rid) : T(0);
        }
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
            if (g < nm && ppos[t] >= 0) {
                if constexpr (STR) O[(m0 + g) * st.o_si + ppos[t] * st.o_sc] = rhs[t] / piv;
                else NFM_STG(rhs[t] / piv, O + (m0 + g) * N + ppos[t]);
            }
        }
    } else if constexpr (DET) {
        const T d = (inversions & 1) ? -det : det;
        if (lr == 0 && g < nm) {
            if constexpr (STR) O[(m0 + g) * st.o_si] = d;
            else NFM_STG(d, O + (m0 + g));
        }
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
        if constexpr (STR) {
            for (int e = tid; e < total; e += NT) {
                const int m = e / ROUT, c = e - m * ROUT;
                O[(m0 + m) * st.o_si + c * st.o_sc] = img[e];
            }
            return;
        }
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

template <typename T, int N, int OP, int R, bool LB>
__global__ __launch_bounds__(256 / R) void roww_kernel(const T *__restrict__ A, const T *__restrict__ B,
                                                       T *__restrict__ O, int64_t n, RowParams<T> p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    roww_tile<T, N, OP, R, LB>(A, B, O, n, (int64_t)blockIdx.x * MPB, p, smem, (int)threadIdx.x);
}

// dynamic LDS of a tile
template <typename T, int N, int OP, bool LB, int MPB_ = MPB>
constexpr size_t tile_lds_bytes()
{
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int NS = ((N + 1 + V - 1) / V) * V;
    return ((size_t)img_elems<T, N, OP, MPB_>() + (LB ? MPB_ * NS : 0)) * sizeof(T);
}

} // namespace roww
} // namespace nfm

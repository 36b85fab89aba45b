// nfm_batched_ops.hpp -- per-lane operations on general small matrices (the `Op` structs
// plugged into rec_kernel).  Shared by nfm_batched.hip (orders 1..8) and nfm_large.hip.
#pragma once
#include "nfm_record_kernel.hpp"
#include "nfm_smallmat.hpp"

namespace nfm {

struct InvParams {
    int perturb;
};

template <typename T, int N>
struct BatchInvOp {
    using RA = Rec<N, N>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<N, N>;
    using Params = InvParams;
    static constexpr int TILE = pick_tile(RA::C * (int)sizeof(T) + 16);
    // 8x8 fp64 (configuration C3): two wavefronts per workgroup, +2 % in same-box A/B runs (256: -20 %)
    static constexpr int kAosTile = (N == 8 && sizeof(T) == 8) ? 128 : TILE;
    static __device__ __forceinline__ void apply(const T (&a)[RA::Cs], const T (&)[1], const T (&)[1],
                                                 T (&r)[RO::Cs], const Params &p)
    {
        if constexpr (N <= 3) {
            inv_closed<T, N>(a, r, p.perturb != 0);
        } else {
            T f[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) f[i][j] = a[i * N + j];
            gj_inverse<T, N>(f);
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) r[i * N + j] = f[i][j];
        }
    }
};

// Inverse built column by column (LU once, N unit-vector solves), each column written
// straight into the lane's slot of the LDS output image: N^2 + 3N live values.  Used for
// the orders where the in-place Gauss-Jordan no longer fits the register file.
// SYM: compact symmetric in and out (sym_invert); else full N x N (batchinv).
template <typename T, int N, bool SYM>
struct InvStreamOp {
    using RA = Rec<(SYM ? 1 : N), (SYM ? sym_k(N) : N)>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = RA;
    using Params = InvParams;
    static constexpr bool kStream = true;
    static constexpr int TILE = 64;
    static __device__ __forceinline__ void apply(const T (&)[RA::Cs], const T (&)[1], const T (&)[1], T (&)[RO::Cs],
                                                 const Params &)
    {
    }
    static __device__ __forceinline__ void apply_stream(const T (&r)[RA::Cs], const T (&)[1], const T (&)[1],
                                                        T *own, const Params &)
    {
        T a[N][N];
        int rowid[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) a[i][j] = SYM ? r[sym_idx(N, i, j)] : r[i * N + j];
        lu_factor_rowid<T, N>(a, rowid);
        // a real loop: unrolled, the scheduler interleaves several columns' substitutions
        // and the live ranges of their x[] vectors push the kernel into scratch
#pragma unroll 1
        for (int c = 0; c < N; ++c) {
            T x[N];
            lu_solve_unit<T, N>(a, rowid, c, x);
            if constexpr (SYM) { // entry (c, j >= c) comes from column c, like the reference
#pragma unroll
                for (int j = c; j < N; ++j) own[sym_idx(N, c, j)] = x[j];
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) own[i * N + c] = x[i];
            }
        }
    }
};

struct NoParamsB {
    int unused;
};

template <typename T, int N>
struct BatchDetOp {
    using RA = Rec<N, N>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, 1>;
    using Params = NoParamsB;
    static constexpr int TILE = pick_tile(RA::C * (int)sizeof(T) + 16);
    static constexpr bool kNoTile = large_no_tile(sizeof(T) == 8, N, LN_BDET);
    static __device__ __forceinline__ void apply(const T (&a)[RA::Cs], const T (&)[1], const T (&)[1], T (&r)[1],
                                                 const Params &)
    {
        if constexpr (N <= 3) {
            r[0] = det_closed<T, N>(a);
        } else {
            T f[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) f[i][j] = a[i * N + j];
            r[0] = lu_det<T, N>(f);
        }
    }
};

// rows x cols matrix times vector; the reference's closed forms (matvec1/2/3,
// _impl/batched.py:133-151) are plain sums of products, evaluated left to right
template <typename T, int R, int C>
struct BatchMatvecOp {
    using RA = Rec<R, C>;
    using RB = Rec<1, C>;
    using RC = NoRec;
    using RO = Rec<1, R>;
    using Params = NoParamsB;
    static constexpr int TILE = pick_tile((RA::C + C + R) * (int)sizeof(T) + 48);
    static __device__ __forceinline__ void apply(const T (&a)[RA::Cs], const T (&v)[C], const T (&)[1], T (&y)[R],
                                                 const Params &)
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < R; ++i) {
            T s = a[i * C] * v[0];
#pragma unroll
            for (int j = 1; j < C; ++j) s = s + a[i * C + j] * v[j];
            y[i] = s;
        }
    }
};

} // namespace nfm

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

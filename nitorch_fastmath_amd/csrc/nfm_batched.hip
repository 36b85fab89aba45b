// nfm_batched.hip -- general small matrices: batchinv / batchdet / batchmatvec
// (reference `_impl/batched.py`).  One matrix per lane; N <= 3 use the reference's
// adjugate closed forms, 4 <= N <= 8 Gauss-Jordan / LU with partial pivoting in
// registers, N > 8 the LDS-resident kernels of nfm_big.hpp.
#include "nfm_record_kernel.hpp"
#include "nfm_smallmat.hpp"
#include "nfm_big.hpp"

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

#define NFM_CASE_N(Nv, ...)   \
    case Nv: {                \
        constexpr int N = Nv; \
        __VA_ARGS__;          \
    } break;
#define NFM_SWITCH_N8(Nexpr, ...)  \
    switch (Nexpr) {               \
        NFM_CASE_N(1, __VA_ARGS__) \
        NFM_CASE_N(2, __VA_ARGS__) \
        NFM_CASE_N(3, __VA_ARGS__) \
        NFM_CASE_N(4, __VA_ARGS__) \
        NFM_CASE_N(5, __VA_ARGS__) \
        NFM_CASE_N(6, __VA_ARGS__) \
        NFM_CASE_N(7, __VA_ARGS__) \
        NFM_CASE_N(8, __VA_ARGS__) \
    default:                       \
        return NFM_ESIZE;          \
    }

template <typename T>
static int batch_inv_t(int N, int flags, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *out,
                       void *stream)
{
    if (N > 8) return big_batch_inv<T>(N, no, ni, a, out, stream);
    InvParams p{(flags & NFM_FLAG_TS_PERTURB) ? 1 : 0};
    NFM_SWITCH_N8(N, return (rec_launch<T, BatchInvOp<T, N>>(a, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int batch_det_t(int N, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    if (N > 8) return big_batch_det<T>(N, no, ni, a, out, stream);
    NoParamsB p{0};
    NFM_SWITCH_N8(N, return (rec_launch<T, BatchDetOp<T, N>>(a, nullptr, nullptr, out, no, ni, p, stream)))
    return NFM_EINVAL;
}

template <typename T>
static int batch_matvec_t(int R, int C, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *v,
                          const nfm_operand *out, void *stream)
{
    NoParamsB p{0};
#define NFM_MV(Rv, Cv) \
    if (R == Rv && C == Cv) return (rec_launch<T, BatchMatvecOp<T, Rv, Cv>>(a, v, nullptr, out, no, ni, p, stream));
    NFM_MV(1, 1) NFM_MV(2, 2) NFM_MV(3, 3) NFM_MV(4, 4) NFM_MV(5, 5) NFM_MV(6, 6) NFM_MV(7, 7) NFM_MV(8, 8)
    NFM_MV(2, 3) NFM_MV(3, 2) NFM_MV(3, 4) NFM_MV(4, 3) NFM_MV(4, 5) NFM_MV(5, 4)
#undef NFM_MV
    return big_batch_matvec<T>(R, C, no, ni, a, v, out, stream);
}

} // namespace nfm

using namespace nfm;

extern "C" {

int nfm_batch_inv(int dtype, int N, int flags, int64_t n_outer, int64_t n_inner, const nfm_operand *a,
                  const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (N < 1 || N > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? batch_inv_t<float>(N, flags, n_outer, n_inner, a, out, stream)
                            : batch_inv_t<double>(N, flags, n_outer, n_inner, a, out, stream);
}

int nfm_batch_det(int dtype, int N, int64_t n_outer, int64_t n_inner, const nfm_operand *a, const nfm_operand *out,
                  void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (N < 1 || N > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? batch_det_t<float>(N, n_outer, n_inner, a, out, stream)
                            : batch_det_t<double>(N, n_outer, n_inner, a, out, stream);
}

int nfm_batch_matvec(int dtype, int rows, int cols, int64_t n_outer, int64_t n_inner, const nfm_operand *a,
                     const nfm_operand *v, const nfm_operand *out, void *stream)
{
    int rc = check_common(dtype, n_outer, n_inner);
    if (rc) return rc;
    if (rows < 1 || rows > NFM_MAX_DIM || cols < 1 || cols > NFM_MAX_DIM) return NFM_ESIZE;
    const bool nonempty = n_outer > 0 && n_inner > 0;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(v, dtype, nonempty))) return rc;
    if ((rc = check_operand(out, dtype, nonempty))) return rc;
    return dtype == NFM_F32 ? batch_matvec_t<float>(rows, cols, n_outer, n_inner, a, v, out, stream)
                            : batch_matvec_t<double>(rows, cols, n_outer, n_inner, a, v, out, stream);
}

} // extern "C"

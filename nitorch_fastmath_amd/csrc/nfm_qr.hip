// nfm_qr.hip -- Givens / Householder / Hessenberg / QR-algorithm entry points
// (reference `qr.py`, `_impl/qr.py`).  Orders 1..8 run in registers through rec_kernel (orders 9..16 too
// where the matrices fit a lane's 512 registers: nfm_qr_large.hip); the other orders 9..16 run the same
// source on matrices held in LDS (qr_lds_kernel below: [element][lane] images, no scratch memory).
// Multi-output routines write ONE packed output record per matrix (the facade hands out
// views): eig_sym [vals | vecs], hessenberg [H | reflectors], qr_hessenberg [Q | R], ...
#include "nfm_record_kernel.hpp"
#include "nfm_qr_core.hpp"

// This file is compiled twelve times (-DNFM_QR_PART=0..11) so that the heavy template instantiations
// build in parallel: parts 0..3 hold the entry points and the register kernels of orders 1..8 (one object
// per group of entry points), parts 4..11 the LDS-resident kernels of orders 9..16 (qr_lds_kernel), one
// object per (dtype, group of operations): 4 + 4 * f64 + g, g = 0: eig_sym values, 1: eig_sym vectors,
// 2: hessenberg / qr_hessenberg / householder, 3: hessenberg_sym / rq_hessenberg; parts 12..27 the REGISTER
// kernels of orders 9..16 (the same Ops as orders 1..8, contiguous operands only), one object per
// (dtype, order): 12 + 8 * f64 + (N - 9) -- the cases whose matrices fit the 512 registers of a lane
// (qr_large_fits below, from the compiler's resource reports); the others stay on the LDS kernels.
#ifndef NFM_QR_PART
#error "compile with -DNFM_QR_PART=0..27"
#endif

namespace nfm {

constexpr int upack_len(int N) { return N > 2 ? (N - 2) * (N - 1) : 0; }

struct QrParams {
    int n;        // run-time order (generic kernels)
    int upper;    // which triangle of a symmetric input holds the data
    int sym;      // rq_hessenberg: tridiagonal shortcut
    int basis;    // householder: component to reflect onto
    int max_iter;
    double tol;
};

// load an N x N record into a 2-D register matrix (optionally mirroring one triangle)
template <typename T, int N>
__device__ __forceinline__ void to_mat(const T (&r)[N * N], T (&a)[N][N], int mirror_upper)
{
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (mirror_upper < 0) a[i][j] = r[i * N + j];
            else if (mirror_upper) a[i][j] = (i <= j) ? r[i * N + j] : r[j * N + i];
            else a[i][j] = (i >= j) ? r[i * N + j] : r[j * N + i];
        }
}

// FAST: fast sweep arithmetic of nfm_qr_core.hpp (FastSweeps); false = reference-order IEEE
template <typename T, int N, bool WITH_U, bool FAST = false>
struct EigSymOp {
    using RA = Rec<N, N>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, N + (WITH_U ? N * N : 0)>;
    using Params = QrParams;
    static constexpr int TILE = pick_tile((RA::C + RO::C) * (int)sizeof(T) + 32);
    // orders >= 5 (and 4 x 4 float64) skip the LDS transpose (op_no_tile): a wavefront's image of their
    // records held the kernel at 2 waves per SIMD, and from there on eig_sym is bound by its dependent
    // arithmetic, not by the load rate (same-box A/B: 8x8 with vectors 1.5x / 1.7x faster in float32 /
    // float64, 6x6 1.1-1.4x, 4x4 float64 1.02-1.09x; 3x3 and 4x4 float32 are better off tiled)
    static constexpr bool kNoTile = N >= 5 || (N == 4 && sizeof(T) == 8);
    static __device__ __forceinline__ void apply(const T (&r)[RA::Cs], const T (&)[1], const T (&)[1],
                                                 T (&o)[RO::Cs], const Params &p)
    {
        T a[N][N], u[N][N], up[N][N], vals[N];
        to_mat<T, N>(r, a, p.upper);
        qr::eig_sym1<T, N, WITH_U, FAST>(a, u, up, vals, N, p.max_iter, p.tol);
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = vals[i];
        if constexpr (WITH_U) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) o[N + i * N + j] = u[i][j];
        }
    }
};

// SYM: hessenberg_sym (tridiagonalisation, output filled symmetric); else hessenberg
template <typename T, int N, bool SYM, bool WITH_U>
struct HessOp {
    using RA = Rec<N, N>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, N * N + (WITH_U ? upack_len(N) : 0)>;
    using Params = QrParams;
    static constexpr int TILE = pick_tile((RA::C + RO::C) * (int)sizeof(T) + 32);
    // orders 9..16: 0.6-2 KiB of records per lane -- an LDS image of a 64-lane tile would be most of a CU's LDS
    // and leave one wavefront per CU; the records are fetched and stored per lane instead (op_no_tile)
    static constexpr bool kNoTile = N >= 9;
    static __device__ __forceinline__ void apply(const T (&r)[RA::Cs], const T (&)[1], const T (&)[1],
                                                 T (&o)[RO::Cs], const Params &p)
    {
        T a[N][N], up[N][N];
        if constexpr (SYM) {
            to_mat<T, N>(r, a, p.upper);
            qr::hessenberg_sym1<T, N, WITH_U>(a, N, up);
        } else {
            to_mat<T, N>(r, a, -1);
            qr::hessenberg1<T, N, WITH_U>(a, N, up);
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) o[i * N + j] = a[i][j];
        if constexpr (WITH_U && N > 2) {
#pragma unroll
            for (int k = 0; k < N - 2; ++k)
#pragma unroll
                for (int c = 0; c < N - 1; ++c) o[N * N + k * (N - 1) + c] = (c < N - 1 - k) ? up[k][c] : T(0);
        }
    }
};

template <typename T, int N>
struct QrHessOp {
    using RA = Rec<N, N>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, 2 * N * N>;
    using Params = QrParams;
    static constexpr int TILE = pick_tile((RA::C + RO::C) * (int)sizeof(T) + 32);
    // 8 x 8 float64: 1.5 KiB of records per lane, i.e. a 96 KiB LDS image per 64-lane workgroup and one
    // wavefront per CU -- fetched and stored per lane instead (op_no_tile)
    static constexpr bool kNoTile = N >= 9 || (RA::C + RO::C) * (int)sizeof(T) >= 1400;
    static __device__ __forceinline__ void apply(const T (&r)[RA::Cs], const T (&)[1], const T (&)[1],
                                                 T (&o)[RO::Cs], const Params &)
    {
        T a[N][N], q[N][N];
        to_mat<T, N>(r, a, -1);
        qr::qr_hessenberg1<T, N>(a, q, N);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                o[i * N + j] = q[i][j];
                o[N * N + i * N + j] = a[i][j];
            }
    }
};

template <typename T, int N, bool WITH_U>
struct RqHessOp {
    using RA = Rec<N, N>;
    using RB = Rec<(WITH_U ? N : 0), (WITH_U ? N : 0)>;
    using RC = NoRec;
    using RO = Rec<1, N * N * (WITH_U ? 2 : 1)>;
    using Params = QrParams;
    static constexpr int TILE = pick_tile((RA::C + RB::C + RO::C) * (int)sizeof(T) + 48);
    static constexpr bool kNoTile = N >= 9; // as HessOp
    static __device__ __forceinline__ void apply(const T (&r)[RA::Cs], const T (&ru)[RB::Cs], const T (&)[1],
                                                 T (&o)[RO::Cs], const Params &p)
    {
        T a[N][N], u[N][N];
        to_mat<T, N>(r, a, -1);
        if constexpr (WITH_U) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) u[i][j] = ru[i * N + j];
        }
        qr::rq_step1<T, N, WITH_U>(a, u, N, N, p.sym != 0);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                o[i * N + j] = a[i][j];
                if constexpr (WITH_U) o[N * N + i * N + j] = u[i][j];
            }
    }
};

template <typename T, int N>
struct HouseholderOp {
    using RA = Rec<1, N>;
    using RB = NoRec;
    using RC = NoRec;
    using RO = Rec<1, N + 1>;
    using Params = QrParams;
    static constexpr int TILE = pick_tile((RA::C + RO::C) * (int)sizeof(T) + 32);
    static __device__ __forceinline__ void apply(const T (&x)[N], const T (&)[1], const T (&)[1], T (&o)[N + 1],
                                                 const Params &p)
    {
        T u[N];
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] = x[i];
        const T alpha = qr::householder1<T, N>(u, N, p.basis);
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = u[i];
        o[N] = alpha;
    }
};

template <typename T>
struct GivensOp {
    using RA = Rec<1, 1>;
    using RB = Rec<1, 1>;
    using RC = NoRec;
    using RO = Rec<1, 2>;
    using Params = QrParams;
    static constexpr int TILE = 256;
    static __device__ __forceinline__ void apply(const T (&x)[1], const T (&y)[1], const T (&)[1], T (&o)[2],
                                                 const Params &)
    {
        qr::givens1(x[0], y[0], o[0], o[1]);
    }
};

// ---------------------------------------------------------------- generic (run-time n)
enum { QG_EIG = 0, QG_EIG_U, QG_HESS, QG_HESS_U, QG_HESSSYM, QG_HESSSYM_U, QG_QR, QG_RQ, QG_RQ_U, QG_HH,
       QG_EIG_FAST, QG_EIG_U_FAST };

// matrices a lane keeps in LDS for OP (n x n elements each)
constexpr int qg_mats(int op)
{
    return op == QG_HH ? 0
           : (op == QG_EIG_U || op == QG_EIG_U_FAST || op == QG_HESS_U || op == QG_QR || op == QG_RQ_U) ? 2
                                                                                                      : 1;
}

// Orders 9..16 whose matrices do not fit the registers of a lane: one matrix per lane all the same, the
// matrices in LDS as [element][lane] images (qr::LdsMat: conflict-free for any element, literal offsets after
// unrolling), vectors and the band of the QR sweeps in registers.  LANES lanes per workgroup -- 64, or 32 /
// 16 when n^2 * matrices * LANES elements would not fit the 160 KiB of a CU (the launcher picks).  The
// reflectors of the symmetric tridiagonalisation live in the upper triangle of the matrix they came from
// (UpperRows), so eig_sym with vectors needs two images and hessenberg_sym with reflectors one.
template <typename T, int OP, int LANES>
__global__ __launch_bounds__(LANES) void qr_lds_kernel(Opnd a, Opnd b, T *__restrict__ out, int64_t out_rec,
                                                       int64_t n_inner, QrParams p)
{
    constexpr int MX = NFM_MAX_DIM;
    extern __shared__ __attribute__((aligned(16))) char qr_smem[];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * LANES + lane;
    const int64_t o = blockIdx.y;
    if (i >= n_inner) return; // a lane only touches its own column of the images: no barrier is needed anywhere
    const int n = p.n;
    const T *pa = reinterpret_cast<const T *>(a.ptr) + o * a.so + i * a.si;
    T *po = out + (o * n_inner + i) * out_rec;
    using M = qr::LdsMat<T, LANES>;
    T *img = reinterpret_cast<T *>(qr_smem);
    M m{img + lane, n}, u{img + n * n * LANES + lane, n};
    if constexpr (OP == QG_HH) {
        T x[MX];
#pragma unroll
        for (int r = 0; r < MX; ++r) x[r] = (r < n) ? pa[r * a.sc] : T(0);
        const T alpha = qr::householder1<T, 0>(x, n, p.basis);
#pragma unroll
        for (int r = 0; r < MX; ++r)
            if (r < n) po[r] = x[r];
        po[n] = alpha;
        return;
    } else {
    constexpr bool EIG = OP == QG_EIG || OP == QG_EIG_U || OP == QG_EIG_FAST || OP == QG_EIG_U_FAST;
    constexpr bool EIGU = OP == QG_EIG_U || OP == QG_EIG_U_FAST;
    const bool symin = EIG || OP == QG_HESSSYM || OP == QG_HESSSYM_U;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            int rr = r, cc = c;
            if (symin && ((p.upper && r > c) || (!p.upper && r < c))) { rr = c; cc = r; }
            m[r][c] = pa[rr * a.sr + cc * a.sc];
        }
    if constexpr (EIG) {
        qr::UpperRows<T, M> up{m};
        T vals[MX];
        qr::eig_sym1<T, 0, EIGU, OP == QG_EIG_FAST || OP == QG_EIG_U_FAST>(m, u, up, vals, n, p.max_iter, p.tol);
#pragma unroll
        for (int r = 0; r < MX; ++r)
            if (r < n) po[r] = vals[r];
        if (EIGU)
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c) po[n + r * n + c] = u[r][c];
    } else if constexpr (OP == QG_HESS || OP == QG_HESS_U) {
        qr::hessenberg1<T, 0, OP == QG_HESS_U>(m, n, u);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) po[r * n + c] = m[r][c];
        if (OP == QG_HESS_U)
            for (int k = 0; k < n - 2; ++k)
                for (int c = 0; c < n - 1; ++c) po[n * n + k * (n - 1) + c] = (c < n - 1 - k) ? u[k][c] : T(0);
    } else if constexpr (OP == QG_HESSSYM || OP == QG_HESSSYM_U) {
        // the tridiagonal result is written out symmetric from its lower half; the upper half of the image
        // holds the reflectors (FILL = false)
        qr::UpperRows<T, M> up{m};
        qr::hessenberg_sym1<T, 0, OP == QG_HESSSYM_U, false, false, false>(m, n, up);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) po[r * n + c] = r >= c ? m[r][c] : m[c][r];
        if (OP == QG_HESSSYM_U)
            for (int k = 0; k < n - 2; ++k)
                for (int c = 0; c < n - 1; ++c) po[n * n + k * (n - 1) + c] = (c < n - 1 - k) ? up[k][c] : T(0);
    } else if constexpr (OP == QG_QR) {
        qr::qr_hessenberg1<T, 0>(m, u, n);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                po[r * n + c] = u[r][c];
                po[n * n + r * n + c] = m[r][c];
            }
    } else { // QG_RQ, QG_RQ_U
        if (OP == QG_RQ_U) {
            const T *pb = reinterpret_cast<const T *>(b.ptr) + o * b.so + i * b.si;
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c) u[r][c] = pb[r * b.sr + c * b.sc];
        }
        qr::rq_step1<T, 0, OP == QG_RQ_U>(m, u, n, n, p.sym != 0);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                po[r * n + c] = m[r][c];
                if (OP == QG_RQ_U) po[n * n + r * n + c] = u[r][c];
            }
    }
    }
}

// in-place rotations / reflections of strided matrices (any order <= 16)
template <typename T>
__global__ __launch_bounds__(256) void givens_apply_kernel(Opnd a, Opnd c, Opnd s, int64_t n_inner, int n, int gi,
                                                           int gj, int side)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t o = blockIdx.y;
    if (i >= n_inner) return;
    T *pa = reinterpret_cast<T *>(a.ptr) + o * a.so + i * a.si;
    const T *pc = reinterpret_cast<const T *>(c.ptr) + o * c.so + i * c.si;
    const T *ps = reinterpret_cast<const T *>(s.ptr) + o * s.so + i * s.si;
    if (side == 0 || side == 2)
        for (int k = 0; k < n; ++k) {
            T a0 = pa[gi * a.sr + k * a.sc], a1 = pa[gj * a.sr + k * a.sc];
            qr::rot1(a0, a1, pc[k * c.sc], ps[k * s.sc]);
            pa[gi * a.sr + k * a.sc] = a0;
            pa[gj * a.sr + k * a.sc] = a1;
        }
    if (side == 1 || side == 2)
        for (int k = 0; k < n; ++k) {
            T a0 = pa[k * a.sr + gi * a.sc], a1 = pa[k * a.sr + gj * a.sc];
            qr::rot1(a0, a1, pc[k * c.sc], ps[k * s.sc]);
            pa[k * a.sr + gi * a.sc] = a0;
            pa[k * a.sr + gj * a.sc] = a1;
        }
}

template <typename T>
__global__ __launch_bounds__(256) void householder_apply_kernel(Opnd a, Opnd u, int64_t n_inner, int n, int m,
                                                                int side)
{
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t o = blockIdx.y;
    if (i >= n_inner) return;
    T *pa = reinterpret_cast<T *>(a.ptr) + o * a.so + i * a.si;
    const T *pu = reinterpret_cast<const T *>(u.ptr) + o * u.so + i * u.si;
    const int k0 = n - m;
    if (side == 0 || side == 2)
        for (int c = 0; c < n; ++c) {
            T d = T(0);
            for (int r = 0; r < m; ++r) d += pu[r * u.sc] * pa[(k0 + r) * a.sr + c * a.sc];
            for (int r = 0; r < m; ++r) pa[(k0 + r) * a.sr + c * a.sc] -= T(2) * (pu[r * u.sc] * d);
        }
    if (side == 1 || side == 2)
        for (int r = 0; r < n; ++r) {
            T d = T(0);
            for (int c = 0; c < m; ++c) d += pa[r * a.sr + (k0 + c) * a.sc] * pu[c * u.sc];
            for (int c = 0; c < m; ++c) pa[r * a.sr + (k0 + c) * a.sc] -= T(2) * (d * pu[c * u.sc]);
        }
}

// ------------------------------------------------------------------------- dispatch
#define NFM_QR_CASE(Nv, ...)  \
    case Nv: {                \
        constexpr int N = Nv; \
        __VA_ARGS__;          \
    } break;
#define NFM_QR_SWITCH8(Nexpr, ...)   \
    switch (Nexpr) {                 \
        NFM_QR_CASE(1, __VA_ARGS__)  \
        NFM_QR_CASE(2, __VA_ARGS__)  \
        NFM_QR_CASE(3, __VA_ARGS__)  \
        NFM_QR_CASE(4, __VA_ARGS__)  \
        NFM_QR_CASE(5, __VA_ARGS__)  \
        NFM_QR_CASE(6, __VA_ARGS__)  \
        NFM_QR_CASE(7, __VA_ARGS__)  \
        NFM_QR_CASE(8, __VA_ARGS__)  \
    default:                         \
        break;                       \
    }

template <typename T, int OP, int LANES>
static int qr_lds_launch_l(const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no, int64_t ni,
                           const QrParams &p, size_t lds, void *stream)
{
    static std::atomic<uint64_t> have{0};
    if (lds > 64 * 1024) {
        const int rc = lds_opt_in(have, reinterpret_cast<const void *>(&qr_lds_kernel<T, OP, LANES>), 160 * 1024);
        if (rc) return rc;
    }
    nfm_operand none = {nullptr, 0, 0, 0, 0};
    const int64_t nblk = (ni + LANES - 1) / LANES;
    if (nblk > 0x7fffffffLL || no > 65535) return NFM_ESIZE;
    dim3 grid((unsigned)nblk, (unsigned)no, 1);
    hipLaunchKernelGGL((qr_lds_kernel<T, OP, LANES>), grid, dim3(LANES), lds, static_cast<hipStream_t>(stream),
                       make_opnd(a, 0), make_opnd(b ? b : &none, 0), static_cast<T *>(out), out_rec, ni, p);
    return launch_status();
}

// 64 lanes per workgroup when their images fit the CU's LDS (at least two workgroups per CU while that is
// possible), else 32, else 16 (two 16 x 16 float64 images of 64 lanes would be 256 KiB)
template <typename T, int OP>
static int qr_lds_launch(const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no,
                         int64_t ni, const QrParams &p, void *stream)
{
    if (no == 0 || ni == 0) return NFM_OK;
    const size_t per_lane = (size_t)qg_mats(OP) * p.n * p.n * sizeof(T);
    constexpr size_t cap = 156 * 1024;
    if (per_lane * 64 <= cap) return qr_lds_launch_l<T, OP, 64>(a, b, out, out_rec, no, ni, p, per_lane * 64, stream);
    if (per_lane * 32 <= cap) return qr_lds_launch_l<T, OP, 32>(a, b, out, out_rec, no, ni, p, per_lane * 32, stream);
    return qr_lds_launch_l<T, OP, 16>(a, b, out, out_rec, no, ni, p, per_lane * 16, stream);
}

// the LDS kernels live in parts 4..11 (one object per dtype and group of operations)
constexpr int qg_group(int op)
{
    return (op == QG_EIG || op == QG_EIG_FAST) ? 0
           : (op == QG_EIG_U || op == QG_EIG_U_FAST) ? 1
           : (op == QG_HESS || op == QG_HESS_U || op == QG_QR || op == QG_HH) ? 2
                                                                             : 3;
}
#define NFM_QR_LDS_DECL(t, g)                                                                                          \
    int qr_lds_##t##_g##g(int op, const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no, \
                          int64_t ni, const QrParams &p, void *stream);
NFM_QR_LDS_DECL(f32, 0) NFM_QR_LDS_DECL(f32, 1) NFM_QR_LDS_DECL(f32, 2) NFM_QR_LDS_DECL(f32, 3)
NFM_QR_LDS_DECL(f64, 0) NFM_QR_LDS_DECL(f64, 1) NFM_QR_LDS_DECL(f64, 2) NFM_QR_LDS_DECL(f64, 3)
#undef NFM_QR_LDS_DECL

// ---------------------------------------------------------------- orders 9..16 in registers
// Which (dtype, order, operation) of the orders 9..16 runs as a register kernel: the cases hipcc compiles
// without a private segment (scripts/survey_qr_large.sh: -Rpass-analysis=kernel-resource-usage of every
// combination; registers beyond the 256 architectural ones are parked in the 256 accumulation registers of the
// lane, which costs a move each way but no memory).  Everything else runs from LDS (qr_lds_kernel).
// largest order without a private segment, per case (f32: everything else fits at every order)
constexpr int NFM_QRL_F32_HESSU_MAX = 15, NFM_QRL_F32_QR_MAX = 16, NFM_QRL_F32_EIGU_MAX = 10, NFM_QRL_F32_EIGUF_MAX = 11, NFM_QRL_F32_RQU_MAX = 11;
constexpr int NFM_QRL_F64_EIG_MAX = 15, NFM_QRL_F64_EIGF_MAX = 15, NFM_QRL_F64_HESS_MAX = 11, NFM_QRL_F64_HSYM_MAX = 15,
              NFM_QRL_F64_QR_MAX = 11, NFM_QRL_F64_RQ_MAX = 13, NFM_QRL_F64_EIGU_MAX = 10, NFM_QRL_F64_EIGUF_MAX = 9,
              NFM_QRL_F64_RQU_MAX = 8;
constexpr bool qr_large_fits(bool f64, int N, int op)
{
    if (N < 9 || N > 16) return false;
    if (op == QG_HH) return true;
    if (!f64) {
        switch (op) {
        case QG_EIG: case QG_EIG_FAST: case QG_HESS: case QG_HESSSYM: case QG_HESSSYM_U: return true;
        case QG_HESS_U: return N <= NFM_QRL_F32_HESSU_MAX;
        case QG_QR: case QG_RQ: return N <= NFM_QRL_F32_QR_MAX;
        case QG_EIG_U: return N <= NFM_QRL_F32_EIGU_MAX;
        case QG_EIG_U_FAST: return N <= NFM_QRL_F32_EIGUF_MAX;
        case QG_RQ_U: return N <= NFM_QRL_F32_RQU_MAX;
        default: return false;
        }
    }
    switch (op) {
    case QG_EIG: return N <= NFM_QRL_F64_EIG_MAX;
    case QG_EIG_FAST: return N <= NFM_QRL_F64_EIGF_MAX;
    case QG_HESS: case QG_HESS_U: return N <= NFM_QRL_F64_HESS_MAX;
    case QG_HESSSYM: case QG_HESSSYM_U: return N <= NFM_QRL_F64_HSYM_MAX;
    case QG_QR: return N <= NFM_QRL_F64_QR_MAX;
    case QG_RQ: return N <= NFM_QRL_F64_RQ_MAX;
    case QG_EIG_U: return N <= NFM_QRL_F64_EIGU_MAX;
    case QG_EIG_U_FAST: return N <= NFM_QRL_F64_EIGUF_MAX;
    case QG_RQ_U: return N <= NFM_QRL_F64_RQU_MAX;
    default: return false;
    }
}
#define NFM_QRL_DECL(t, n)                                                                                     \
    int qr_large_##t##_n##n(int op, const nfm_operand *a, const nfm_operand *b, const nfm_operand *o, int64_t no, \
                            int64_t ni, const QrParams &p, void *stream);
#define NFM_QRL_DECL8(t) \
    NFM_QRL_DECL(t, 9) NFM_QRL_DECL(t, 10) NFM_QRL_DECL(t, 11) NFM_QRL_DECL(t, 12) NFM_QRL_DECL(t, 13) \
    NFM_QRL_DECL(t, 14) NFM_QRL_DECL(t, 15) NFM_QRL_DECL(t, 16)
NFM_QRL_DECL8(f32) NFM_QRL_DECL8(f64)
#undef NFM_QRL_DECL8
#undef NFM_QRL_DECL

// NFM_EFALLBACK: not a register case (order, operation or layout) -- the caller takes the LDS kernel
template <typename T>
static int qr_large_call(int op, const nfm_operand *a, const nfm_operand *b, const nfm_operand *o, int64_t no, int64_t ni,
                         const QrParams &p, void *stream)
{
    if (!qr_large_fits(sizeof(T) == 8, p.n, op)) return NFM_EFALLBACK;
#define NFM_QRL_CALL(n)                                                                  \
    case n:                                                                              \
        return sizeof(T) == 4 ? qr_large_f32_n##n(op, a, b, o, no, ni, p, stream)         \
                              : qr_large_f64_n##n(op, a, b, o, no, ni, p, stream);
    switch (p.n) {
        NFM_QRL_CALL(9) NFM_QRL_CALL(10) NFM_QRL_CALL(11) NFM_QRL_CALL(12) NFM_QRL_CALL(13) NFM_QRL_CALL(14)
        NFM_QRL_CALL(15) NFM_QRL_CALL(16)
    default: return NFM_EFALLBACK;
    }
#undef NFM_QRL_CALL
}

#if NFM_QR_PART >= 12 && NFM_QR_PART <= 27
#define NFM_QRL_F64 ((NFM_QR_PART - 12) / 8)
#if (NFM_QR_PART - 12) % 8 == 0 // (a literal: it is pasted into the function's name)
#define NFM_QRL_N 9
#elif (NFM_QR_PART - 12) % 8 == 1
#define NFM_QRL_N 10
#elif (NFM_QR_PART - 12) % 8 == 2
#define NFM_QRL_N 11
#elif (NFM_QR_PART - 12) % 8 == 3
#define NFM_QRL_N 12
#elif (NFM_QR_PART - 12) % 8 == 4
#define NFM_QRL_N 13
#elif (NFM_QR_PART - 12) % 8 == 5
#define NFM_QRL_N 14
#elif (NFM_QR_PART - 12) % 8 == 6
#define NFM_QRL_N 15
#else
#define NFM_QRL_N 16
#endif
#if NFM_QRL_F64
using TQl = double;
#define NFM_QRL_NAME2(n) qr_large_f64_n##n
#else
using TQl = float;
#define NFM_QRL_NAME2(n) qr_large_f32_n##n
#endif
#define NFM_QRL_NAME1(n) NFM_QRL_NAME2(n)
#define NFM_QRL_NAME NFM_QRL_NAME1(NFM_QRL_N)
int NFM_QRL_NAME(int op, const nfm_operand *a, const nfm_operand *b, const nfm_operand *o, int64_t no, int64_t ni,
                 const QrParams &p, void *stream)
{
    constexpr int N = NFM_QRL_N;
    constexpr bool F = NFM_QRL_F64 != 0;
#define NFM_QRL_CASE(OPv, ...)                                                                           \
    case OPv:                                                                                            \
        if constexpr (qr_large_fits(F, N, OPv)) return (rec_launch<TQl, __VA_ARGS__, true>(a, b, nullptr, o, no, ni, p, stream)); \
        break;
    switch (op) {
        NFM_QRL_CASE(QG_EIG, EigSymOp<TQl, N, false, false>)
        NFM_QRL_CASE(QG_EIG_U, EigSymOp<TQl, N, true, false>)
        NFM_QRL_CASE(QG_EIG_FAST, EigSymOp<TQl, N, false, true>)
        NFM_QRL_CASE(QG_EIG_U_FAST, EigSymOp<TQl, N, true, true>)
        NFM_QRL_CASE(QG_HESS, HessOp<TQl, N, false, false>)
        NFM_QRL_CASE(QG_HESS_U, HessOp<TQl, N, false, true>)
        NFM_QRL_CASE(QG_HESSSYM, HessOp<TQl, N, true, false>)
        NFM_QRL_CASE(QG_HESSSYM_U, HessOp<TQl, N, true, true>)
        NFM_QRL_CASE(QG_QR, QrHessOp<TQl, N>)
        NFM_QRL_CASE(QG_RQ, RqHessOp<TQl, N, false>)
        NFM_QRL_CASE(QG_RQ_U, RqHessOp<TQl, N, true>)
        NFM_QRL_CASE(QG_HH, HouseholderOp<TQl, N>)
    default: break;
    }
#undef NFM_QRL_CASE
    return NFM_EFALLBACK;
}
#endif

template <typename T, int OP>
static int qr_generic_launch(const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no,
                             int64_t ni, const QrParams &p, void *stream)
{
    constexpr int g = qg_group(OP);
    {   // orders 9..16 that fit the registers of a lane, contiguous operands (NFM_EFALLBACK otherwise)
        const nfm_operand o = {out, ni * out_rec, out_rec, 0, 1};
        const int rc = qr_large_call<T>(OP, a, b, &o, no, ni, p, stream);
        if (rc != NFM_EFALLBACK) return rc;
    }
    if constexpr (sizeof(T) == 4) {
        if constexpr (g == 0) return qr_lds_f32_g0(OP, a, b, out, out_rec, no, ni, p, stream);
        else if constexpr (g == 1) return qr_lds_f32_g1(OP, a, b, out, out_rec, no, ni, p, stream);
        else if constexpr (g == 2) return qr_lds_f32_g2(OP, a, b, out, out_rec, no, ni, p, stream);
        else return qr_lds_f32_g3(OP, a, b, out, out_rec, no, ni, p, stream);
    } else {
        if constexpr (g == 0) return qr_lds_f64_g0(OP, a, b, out, out_rec, no, ni, p, stream);
        else if constexpr (g == 1) return qr_lds_f64_g1(OP, a, b, out, out_rec, no, ni, p, stream);
        else if constexpr (g == 2) return qr_lds_f64_g2(OP, a, b, out, out_rec, no, ni, p, stream);
        else return qr_lds_f64_g3(OP, a, b, out, out_rec, no, ni, p, stream);
    }
}

#if NFM_QR_PART >= 4 && NFM_QR_PART <= 11
#define NFM_QR_LDS_F64 ((NFM_QR_PART - 4) / 4)
#define NFM_QR_LDS_G ((NFM_QR_PART - 4) % 4)
#if NFM_QR_LDS_F64
using TLds = double;
#define NFM_QR_LDS_NAME(g) qr_lds_f64_g##g
#else
using TLds = float;
#define NFM_QR_LDS_NAME(g) qr_lds_f32_g##g
#endif
#define NFM_QR_LDS_CASE(OPv) \
    case OPv: return qr_lds_launch<TLds, OPv>(a, b, out, out_rec, no, ni, p, stream);
#if NFM_QR_LDS_G == 0
int NFM_QR_LDS_NAME(0)(int op, const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no, int64_t ni,
                       const QrParams &p, void *stream)
{
    switch (op) { NFM_QR_LDS_CASE(QG_EIG) NFM_QR_LDS_CASE(QG_EIG_FAST) default: return NFM_EINVAL; }
}
#elif NFM_QR_LDS_G == 1
int NFM_QR_LDS_NAME(1)(int op, const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no, int64_t ni,
                       const QrParams &p, void *stream)
{
    switch (op) { NFM_QR_LDS_CASE(QG_EIG_U) NFM_QR_LDS_CASE(QG_EIG_U_FAST) default: return NFM_EINVAL; }
}
#elif NFM_QR_LDS_G == 2
int NFM_QR_LDS_NAME(2)(int op, const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no, int64_t ni,
                       const QrParams &p, void *stream)
{
    switch (op) {
        NFM_QR_LDS_CASE(QG_HESS) NFM_QR_LDS_CASE(QG_HESS_U) NFM_QR_LDS_CASE(QG_QR) NFM_QR_LDS_CASE(QG_HH)
    default: return NFM_EINVAL;
    }
}
#else
int NFM_QR_LDS_NAME(3)(int op, const nfm_operand *a, const nfm_operand *b, void *out, int64_t out_rec, int64_t no, int64_t ni,
                       const QrParams &p, void *stream)
{
    switch (op) {
        NFM_QR_LDS_CASE(QG_HESSSYM) NFM_QR_LDS_CASE(QG_HESSSYM_U) NFM_QR_LDS_CASE(QG_RQ) NFM_QR_LDS_CASE(QG_RQ_U)
    default: return NFM_EINVAL;
    }
}
#endif
#endif

// the packed output record is a plain contiguous (n_outer * n_inner, rec) buffer
static nfm_operand packed_out(void *out, int64_t rec, int64_t ni)
{
    nfm_operand o = {out, ni * rec, rec, 0, 1};
    return o;
}

template <typename T>
static int eig_sym_t(int N, int with_u, int fast, int64_t no, int64_t ni, const nfm_operand *a, void *out,
                     const QrParams &p, void *stream)
{
    const int64_t rec = N + (with_u ? N * N : 0);
    nfm_operand o = packed_out(out, rec, ni);
    if constexpr (qr::FastSweeps<T>::on) {
        if (fast && with_u) {
            NFM_QR_SWITCH8(N, return (rec_launch<T, EigSymOp<T, N, true, true>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
            return qr_generic_launch<T, QG_EIG_U_FAST>(a, nullptr, out, rec, no, ni, p, stream);
        } else if (fast) {
            NFM_QR_SWITCH8(N, return (rec_launch<T, EigSymOp<T, N, false, true>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
            return qr_generic_launch<T, QG_EIG_FAST>(a, nullptr, out, rec, no, ni, p, stream);
        }
    }
    if (with_u) {
        NFM_QR_SWITCH8(N, return (rec_launch<T, EigSymOp<T, N, true>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
        return qr_generic_launch<T, QG_EIG_U>(a, nullptr, out, rec, no, ni, p, stream);
    }
    NFM_QR_SWITCH8(N, return (rec_launch<T, EigSymOp<T, N, false>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
    return qr_generic_launch<T, QG_EIG>(a, nullptr, out, rec, no, ni, p, stream);
}

template <typename T>
static int hess_t(int N, int sym, int with_u, int64_t no, int64_t ni, const nfm_operand *a, void *out,
                  const QrParams &p, void *stream)
{
    const int64_t rec = N * N + (with_u ? upack_len(N) : 0);
    nfm_operand o = packed_out(out, rec, ni);
    if (sym && with_u) {
        NFM_QR_SWITCH8(N, return (rec_launch<T, HessOp<T, N, true, true>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
        return qr_generic_launch<T, QG_HESSSYM_U>(a, nullptr, out, rec, no, ni, p, stream);
    } else if (sym) {
        NFM_QR_SWITCH8(N, return (rec_launch<T, HessOp<T, N, true, false>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
        return qr_generic_launch<T, QG_HESSSYM>(a, nullptr, out, rec, no, ni, p, stream);
    } else if (with_u) {
        NFM_QR_SWITCH8(N, return (rec_launch<T, HessOp<T, N, false, true>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
        return qr_generic_launch<T, QG_HESS_U>(a, nullptr, out, rec, no, ni, p, stream);
    }
    NFM_QR_SWITCH8(N, return (rec_launch<T, HessOp<T, N, false, false>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
    return qr_generic_launch<T, QG_HESS>(a, nullptr, out, rec, no, ni, p, stream);
}

template <typename T>
static int qr_hess_t(int N, int64_t no, int64_t ni, const nfm_operand *a, void *out, const QrParams &p, void *stream)
{
    const int64_t rec = 2 * N * N;
    nfm_operand o = packed_out(out, rec, ni);
    NFM_QR_SWITCH8(N, return (rec_launch<T, QrHessOp<T, N>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
    return qr_generic_launch<T, QG_QR>(a, nullptr, out, rec, no, ni, p, stream);
}

template <typename T>
static int rq_hess_t(int N, int64_t no, int64_t ni, const nfm_operand *a, const nfm_operand *u, void *out,
                     const QrParams &p, void *stream)
{
    const int64_t rec = N * N * (u ? 2 : 1);
    nfm_operand o = packed_out(out, rec, ni);
    if (u) {
        // 8 x 8 float64 with U: two 128-register matrices plus the packed output.  Its mixed-layout kernel form
        // (per-operand mode branches) does not fit the register file, so only the contiguous form is built and
        // other layouts take the LDS-resident kernel, which reads any strides.
        NFM_QR_SWITCH8(N, {
            if constexpr (N == 8 && sizeof(T) == 8) {
                const int rc = rec_launch<T, RqHessOp<T, N, true>, true>(a, u, nullptr, &o, no, ni, p, stream);
                if (rc != NFM_EFALLBACK) return rc;
            } else {
                return (rec_launch<T, RqHessOp<T, N, true>>(a, u, nullptr, &o, no, ni, p, stream));
            }
        })
        return qr_generic_launch<T, QG_RQ_U>(a, u, out, rec, no, ni, p, stream);
    }
    NFM_QR_SWITCH8(N, return (rec_launch<T, RqHessOp<T, N, false>>(a, nullptr, nullptr, &o, no, ni, p, stream)))
    return qr_generic_launch<T, QG_RQ>(a, nullptr, out, rec, no, ni, p, stream);
}

template <typename T>
static int householder_t(int N, int64_t no, int64_t ni, const nfm_operand *x, void *out, const QrParams &p,
                         void *stream)
{
    const int64_t rec = N + 1;
    nfm_operand o = packed_out(out, rec, ni);
    NFM_QR_SWITCH8(N, return (rec_launch<T, HouseholderOp<T, N>>(x, nullptr, nullptr, &o, no, ni, p, stream)))
    return qr_generic_launch<T, QG_HH>(x, nullptr, out, rec, no, ni, p, stream);
}

static QrParams mkparams(int n, int upper, int sym, int basis, int max_iter, double tol)
{
    QrParams p;
    p.n = n;
    p.upper = upper;
    p.sym = sym;
    p.basis = basis;
    p.max_iter = max_iter;
    p.tol = tol;
    return p;
}

} // namespace nfm

using namespace nfm;

#define QR_COMMON_CHECKS(N)                                   \
    int rc = check_common(dtype, n_outer, n_inner);           \
    if (rc) return rc;                                        \
    if ((N) < 1 || (N) > NFM_MAX_DIM) return NFM_ESIZE;       \
    const bool nonempty = n_outer > 0 && n_inner > 0;         \
    (void)nonempty;

extern "C" {

#if NFM_QR_PART == 3
int nfm_qr_givens(int dtype, int64_t n_outer, int64_t n_inner, const nfm_operand *x, const nfm_operand *y, void *out,
                  void *stream)
{
    QR_COMMON_CHECKS(1)
    if ((rc = check_operand(x, dtype, nonempty))) return rc;
    if ((rc = check_operand(y, dtype, nonempty))) return rc;
    if (nonempty && out == nullptr) return NFM_EINVAL;
    nfm_operand o = packed_out(out, 2, n_inner);
    QrParams p = mkparams(1, 0, 0, 0, 0, 0.0);
    return dtype == NFM_F32 ? rec_launch<float, GivensOp<float>>(x, y, nullptr, &o, n_outer, n_inner, p, stream)
                            : rec_launch<double, GivensOp<double>>(x, y, nullptr, &o, n_outer, n_inner, p, stream);
}
#endif

#if NFM_QR_PART == 3
int nfm_qr_givens_apply(int dtype, int N, int side, int i, int j, int64_t n_outer, int64_t n_inner,
                        const nfm_operand *a, const nfm_operand *c, const nfm_operand *s, void *stream)
{
    QR_COMMON_CHECKS(N)
    if (side < 0 || side > 2 || i < 0 || j < 0 || i >= N || j >= N) return NFM_EINVAL;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(c, dtype, nonempty))) return rc;
    if ((rc = check_operand(s, dtype, nonempty))) return rc;
    if (!nonempty) return NFM_OK;
    dim3 grid((unsigned)((n_inner + 255) / 256), (unsigned)n_outer, 1);
    if (dtype == NFM_F32)
        hipLaunchKernelGGL((givens_apply_kernel<float>), grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           make_opnd(a, 0), make_opnd(c, 0), make_opnd(s, 0), n_inner, N, i, j, side);
    else
        hipLaunchKernelGGL((givens_apply_kernel<double>), grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           make_opnd(a, 0), make_opnd(c, 0), make_opnd(s, 0), n_inner, N, i, j, side);
    return launch_status();
}
#endif

#if NFM_QR_PART == 3
int nfm_qr_householder(int dtype, int N, int basis, int64_t n_outer, int64_t n_inner, const nfm_operand *x,
                       void *out, void *stream)
{
    QR_COMMON_CHECKS(N)
    if (basis < 0 || basis >= N) return NFM_EINVAL;
    if ((rc = check_operand(x, dtype, nonempty))) return rc;
    if (nonempty && out == nullptr) return NFM_EINVAL;
    QrParams p = mkparams(N, 0, 0, basis, 0, 0.0);
    return dtype == NFM_F32 ? householder_t<float>(N, n_outer, n_inner, x, out, p, stream)
                            : householder_t<double>(N, n_outer, n_inner, x, out, p, stream);
}
#endif

#if NFM_QR_PART == 3
int nfm_qr_householder_apply(int dtype, int N, int m, int side, int64_t n_outer, int64_t n_inner,
                             const nfm_operand *a, const nfm_operand *u, void *stream)
{
    QR_COMMON_CHECKS(N)
    if (side < 0 || side > 2 || m < 1 || m > N) return NFM_EINVAL;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if ((rc = check_operand(u, dtype, nonempty))) return rc;
    if (!nonempty) return NFM_OK;
    dim3 grid((unsigned)((n_inner + 255) / 256), (unsigned)n_outer, 1);
    if (dtype == NFM_F32)
        hipLaunchKernelGGL((householder_apply_kernel<float>), grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           make_opnd(a, 0), make_opnd(u, 0), n_inner, N, m, side);
    else
        hipLaunchKernelGGL((householder_apply_kernel<double>), grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           make_opnd(a, 0), make_opnd(u, 0), n_inner, N, m, side);
    return launch_status();
}
#endif

#if NFM_QR_PART == 1
int nfm_qr_hessenberg(int dtype, int N, int sym, int upper, int with_u, int64_t n_outer, int64_t n_inner,
                      const nfm_operand *a, void *out, void *stream)
{
    QR_COMMON_CHECKS(N)
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if (nonempty && out == nullptr) return NFM_EINVAL;
    QrParams p = mkparams(N, upper ? 1 : 0, 0, 0, 0, 0.0);
    return dtype == NFM_F32 ? hess_t<float>(N, sym, with_u, n_outer, n_inner, a, out, p, stream)
                            : hess_t<double>(N, sym, with_u, n_outer, n_inner, a, out, p, stream);
}
#endif

#if NFM_QR_PART == 2
int nfm_qr_qr_hessenberg(int dtype, int N, int64_t n_outer, int64_t n_inner, const nfm_operand *h, void *out,
                         void *stream)
{
    QR_COMMON_CHECKS(N)
    if ((rc = check_operand(h, dtype, nonempty))) return rc;
    if (nonempty && out == nullptr) return NFM_EINVAL;
    QrParams p = mkparams(N, 0, 0, 0, 0, 0.0);
    return dtype == NFM_F32 ? qr_hess_t<float>(N, n_outer, n_inner, h, out, p, stream)
                            : qr_hess_t<double>(N, n_outer, n_inner, h, out, p, stream);
}
#endif

#if NFM_QR_PART == 2
int nfm_qr_rq_hessenberg(int dtype, int N, int sym, int64_t n_outer, int64_t n_inner, const nfm_operand *h,
                         const nfm_operand *u, void *out, void *stream)
{
    QR_COMMON_CHECKS(N)
    if ((rc = check_operand(h, dtype, nonempty))) return rc;
    if (u && (rc = check_operand(u, dtype, nonempty))) return rc;
    if (nonempty && out == nullptr) return NFM_EINVAL;
    QrParams p = mkparams(N, 0, sym ? 1 : 0, 0, 0, 0.0);
    return dtype == NFM_F32 ? rq_hess_t<float>(N, n_outer, n_inner, h, u, out, p, stream)
                            : rq_hess_t<double>(N, n_outer, n_inner, h, u, out, p, stream);
}
#endif

#if NFM_QR_PART == 0
int nfm_qr_eig_sym(int dtype, int N, int upper, int flags, int max_iter, double tol, int64_t n_outer,
                   int64_t n_inner, const nfm_operand *a, void *out, void *stream)
{
    QR_COMMON_CHECKS(N)
    if (max_iter < 0 || flags < 0 || flags > (NFM_EIG_VECTORS | NFM_EIG_FAST)) return NFM_EINVAL;
    const int with_u = (flags & NFM_EIG_VECTORS) != 0, fast = (flags & NFM_EIG_FAST) != 0;
    if ((rc = check_operand(a, dtype, nonempty))) return rc;
    if (nonempty && out == nullptr) return NFM_EINVAL;
    QrParams p = mkparams(N, upper ? 1 : 0, 1, 0, max_iter, tol);
    return dtype == NFM_F32 ? eig_sym_t<float>(N, with_u, fast, n_outer, n_inner, a, out, p, stream)
                            : eig_sym_t<double>(N, with_u, fast, n_outer, n_inner, a, out, p, stream);
}
#endif

} // extern "C"

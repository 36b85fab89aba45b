// nfm_spd.hip -- orders 9..16 of sym_solve / sym_invert / sym_det on contiguous compact records:
// POSITIVE DEFINITE FIRST.
//
// The matrices these functions are fed are Hessians: symmetric positive definite.  The reference sends every
// order > 4 through `torch.linalg.solve` of the densified matrix (`_impl/sym.py:392-396`: LU with partial pivoting),
// and so did this library: N^2 registers, 2 N^3 / 3 fma and -- registers cannot be indexed at run time -- as many
// selects again for the row exchanges (16x16 float32 solve: 6 650 instructions, 3 185 of them v_cndmask, one
// wavefront per SIMD; float64 13..16 do not fit a lane at all and went to the one-matrix-per-16-lanes kernels).
// A positive definite matrix needs none of that: A = U^T D U without pivoting is backward stable, works on the
// N (N + 1) / 2 stored values in place and costs N^3 / 6 fma (nfm_smallmat.hpp: ldl_factor).
//
//   * one matrix per lane, the compact record in registers; a wavefront is a workgroup and moves its 64 records
//     through an LDS image with whole-line accesses (float64 orders 13..16, whose image would be 46-70 KiB: every
//     lane fetches and stores its records itself with element-aligned 16-byte accesses, all loads in flight
//     before the first use);
//   * every pivot positive <=> the matrix is positive definite (Sylvester).  The wavefront votes: when every
//     matrix passed, the lanes finish (solve: two triangular sweeps; inverse: W = U^-1, then W D^-1 W^T, in
//     place; determinant: the product of the pivots) and store;
//   * the matrices that did not pass go to the pivoted elimination of nfm_rowwave in GROUPS of 16 (what one of its
//     tiles works on); every other group finishes and stores as if nothing had happened.  When the output aliases no
//     input, the bad groups are MARKED (a NaN in the first output element of the group's first matrix) and a second
//     launch (`redo_kernel`) runs the elimination on the marked groups at the row-wave kernels' own occupancy; an
//     in-place call has no place for a mark and its wavefronts redo their bad groups themselves.  The answer for
//     indefinite, singular and NaN input is what it was before this file existed.
//
// Results differ from the LU's in rounding only (both are backward stable on these matrices); parity is asserted
// through the tolerance of the tests, like every result beyond the closed forms.
#include "nfm_rowwave_core.hpp"
#include "nfm_sym_ops.hpp"
#include "nfm_spd.hpp"

#ifndef NFM_SPD_PART
#error "compile with -DNFM_SPD_PART=0..7"
#endif
#ifndef NFM_GEN_F64_MAX_INV
#define NFM_GEN_F64_MAX_INV 13 // largest float64 orders of the general no-exchange kernels (N^2 doubles per lane + staging)
#define NFM_GEN_F64_MAX_DET 13
#endif

namespace nfm {
namespace spd {

using roww::RowParams;

// rows per lane of the fallback (FR; 16 / FR lanes per matrix, 4 FR matrices per pass).  4 in both dtypes: with one
// row per lane (16 passes of 4 matrices) a float64 batch in which every matrix is indefinite ran 5x behind the
// pivoted kernels alone
#ifndef NFM_SPD_FR64
#define NFM_SPD_FR64 4
#endif
template <typename T>
constexpr int fallback_rows()
{
    return sizeof(T) == 8 ? NFM_SPD_FR64 : 4;
}

constexpr int roww_op(int op)
{
    return op == SP_SOLVE ? roww::RW_SOLVE_SYM : op == SP_INV ? roww::RW_INV_SYM
           : op == SP_INVDIAG ? roww::RW_INVDIAG_SYM : op == SP_DET ? roww::RW_DET_SYM
           : op == SP_GINV ? roww::RW_INV_GEN : roww::RW_DET_GEN;
}

// a lane's record <-> registers: element-aligned 16-byte accesses and a tail of single elements.  PLAIN accesses:
// the eight a lane makes to one 128-byte line come from eight instructions, the line has to stay in the cache
// between them (nontemporal loads fetched it eight times: nfm_reduce_median_lane.hip measured 1.65 against 3.5 TB/s)
template <typename T, int C>
__device__ __forceinline__ void load_record(const T *__restrict__ p, T (&r)[C])
{
    using VG = typename VecOf<T>::gtype;
    constexpr int V = VecOf<T>::N, F = C / V;
    VG w[F > 0 ? F : 1];
#pragma unroll
    for (int i = 0; i < F; ++i) w[i] = *reinterpret_cast<const VG *>(p + i * V);
#pragma unroll
    for (int i = F * V; i < C; ++i) r[i] = p[i];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int q = 0; q < V; ++q) r[i * V + q] = w[i][q];
}
template <typename T, int C>
__device__ __forceinline__ void store_record(T *__restrict__ p, const T (&r)[C])
{
    using VG = typename VecOf<T>::gtype;
    constexpr int V = VecOf<T>::N, F = C / V;
#pragma unroll
    for (int i = 0; i < F; ++i) {
        VG w;
#pragma unroll
        for (int q = 0; q < V; ++q) w[q] = r[i * V + q];
        *reinterpret_cast<VG *>(p + i * V) = w;
    }
#pragma unroll
    for (int i = F * V; i < C; ++i) p[i] = r[i];
}

// The float64 kernels of the top orders hold more than 256 values per lane: they are compiled for ONE wavefront per
// SIMD, which opens the whole 512-register file (without it the backend stops at 256 and spills to scratch).  The
// others carry no bound: any bound is also a licence -- at `amdgpu_waves_per_eu(2)` the scheduler stretched the
// fallback (70-120 registers on its own) to 256 and spilled.
template <typename T, int N>
constexpr int spd_max_waves()
{
    return (sizeof(T) == 8 && N >= 14) ? 1 : 8;
}

// ---------------------------------------------------------------------------------------------
// records <-> lanes through LDS images of 64 / S records (S = 1: the whole wavefront at once).  Whole-line 16-byte
// accesses on the global side; the ragged last wavefront of a batch takes rolled element loops (an unrolled,
// predicated ragged path -- TileIO's -- set the register count of these kernels: 292 instead of 183 at one point).
template <typename T, int C, int S>
struct SubOut {
    static constexpr int L = 64 / S; // records per image
    using G = TileIO<T, C, L>;       // geometry of the image: padded row stride, vector <-> byte offset
    using V = typename VecOf<T>::type;
    using VG = typename VecOf<T>::gtype;
    static constexpr int kVec = VecOf<T>::N;
    static constexpr size_t kLdsBytes = G::kLdsBytes;
    // g: first record of the wavefront's 64; avail: elements that may be written from g on; `bad`: bit k set = the
    // records 64 / NG * k .. of the wavefront are NOT to be written (their matrices go to the fallback, which must
    // find its input untouched when the call is in place); group boundaries are multiples of 16 bytes
    template <int NG>
    static __device__ __forceinline__ void put(char *smem, const T (&r)[C], T *__restrict__ g, int64_t avail, int tid,
                                               unsigned bad)
    {
        unsigned char *lds = reinterpret_cast<unsigned char *>(smem);
        constexpr int GE = (64 / NG) * C; // elements of a group
        static_assert(GE % kVec == 0, "a vector never straddles two groups");
#pragma unroll
        for (int s = 0; s < S; ++s) {
            __syncthreads();
            if (tid / L == s) {
                unsigned char *p = lds + (tid % L) * G::kRowStride;
                // (big float64 records go value by value: packing 136 doubles into aligned register quads for 16-byte
                // LDS writes costs copies the float64 inverses of orders 14..16 have no registers for)
                if constexpr (G::kWide && !(sizeof(T) == 8 && C >= 100)) {
#pragma unroll
                    for (int q = 0; q < G::kSlots; ++q) {
                        V v;
#pragma unroll
                        for (int k = 0; k < kVec; ++k) v[k] = r[q * kVec + k];
                        *reinterpret_cast<V *>(p + q * 16) = v;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) *reinterpret_cast<T *>(p + c * sizeof(T)) = r[c];
                }
            }
            __syncthreads();
            T *gs = g + (int64_t)s * L * C;
            const int64_t left = avail - (int64_t)s * L * C;
            if (left >= (int64_t)L * C && bad == 0) { // uniform: nearly always
#pragma unroll
                for (int q = tid; q < G::kNVec; q += 64) {
                    const V v = *reinterpret_cast<const V *>(lds + G::lds_off(q));
                    NFM_STG(static_cast<VG>(v), reinterpret_cast<VG *>(gs) + q);
                }
            } else if (left > 0) { // the ragged last wavefront, or one with matrices for the fallback: a rolled loop
#pragma unroll 1
                for (int q = tid; q < G::kNVec; q += 64) {
                    const int e0 = s * L * C + q * kVec; // element of the wavefront's block
                    if ((bad >> (e0 / GE)) & 1u) continue;
                    const V v = *reinterpret_cast<const V *>(lds + G::lds_off(q));
#pragma unroll
                    for (int k = 0; k < kVec; ++k)
                        if ((int64_t)q * kVec + k < left) gs[(int64_t)q * kVec + k] = v[k];
                }
            }
        }
    }
};

// the way in: the wavefront's 64 records in S images of 64 / S.  EVERY load of all S images is issued first (C values
// per lane in all -- the staging registers of image s are free once it is parked in LDS, the lanes of image s then
// pick up their records: peak C + C (S - 1) / S registers), one round trip to memory.  Whole lines instead of a lane
// walking its own record: at 16x16 float32 the records are 1 KiB apart, the 64 addresses of a lane-by-lane load
// fall into the same cache sets and channel, and the general kernels ran at 0.33 of the roofline that way.
template <typename T, int C, int S>
struct SubIn {
    static constexpr int L = 64 / S;
    using G = TileIO<T, C, L>;
    using V = typename VecOf<T>::type;
    using VG = typename VecOf<T>::gtype;
    static constexpr int kVec = VecOf<T>::N;
    static constexpr int IT = (G::kNVec + 63) / 64;
    static constexpr size_t kLdsBytes = G::kLdsBytes;
    // this lane's record out of the image
    static __device__ __forceinline__ void pick(const unsigned char *lds, T (&r)[C], int tid)
    {
        const unsigned char *p = lds + (tid % L) * G::kRowStride;
        if constexpr (G::kWide) {
#pragma unroll
            for (int q = 0; q < G::kSlots; ++q) {
                const V v = *reinterpret_cast<const V *>(p + q * 16);
#pragma unroll
                for (int k = 0; k < kVec; ++k) r[q * kVec + k] = v[k];
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) r[c] = *reinterpret_cast<const T *>(p + c * sizeof(T));
        }
    }
    // a full wavefront: issue (every load of the S images, streaming loads, no waits) ... land (image by image)
    struct Stage {
        V v[S][IT];
    };
    static __device__ __forceinline__ void issue(Stage &st, const T *__restrict__ g, int tid)
    {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int q = tid + it * 64;
                if (G::kNVec % 64 == 0 || q < G::kNVec)
                    st.v[s][it] = NFM_LDG(reinterpret_cast<const VG *>(g + (int64_t)s * L * C) + q);
            }
    }
    static __device__ __forceinline__ void land(char *smem, const Stage &st, T (&r)[C], int tid)
    {
        unsigned char *lds = reinterpret_cast<unsigned char *>(smem);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int q = tid + it * 64;
                if (G::kNVec % 64 == 0 || q < G::kNVec) *reinterpret_cast<V *>(lds + G::lds_off(q)) = st.v[s][it];
            }
            __syncthreads();
            if (tid / L == s) pick(lds, r, tid);
        }
    }
    // the ragged last wavefront of a batch: element by element on a rolled loop, straight into the image (no staging
    // registers: an unrolled, predicated version of this path set the register count of the kernel)
    static __device__ __forceinline__ void get_ragged(char *smem, T (&r)[C], const T *__restrict__ g, int64_t avail, int tid)
    {
        unsigned char *lds = reinterpret_cast<unsigned char *>(smem);
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            const T *gs = g + (int64_t)s * L * C;
            const int64_t left = avail - (int64_t)s * L * C;
            __syncthreads();
#pragma unroll 1
            for (int e = tid; e < L * C; e += 64) {
                const int rr = e / C, c = e - rr * C;
                *reinterpret_cast<T *>(lds + rr * G::kRowStride + c * (int)sizeof(T)) = e < left ? gs[e] : T(0);
            }
            __syncthreads();
            if (tid / L == s) pick(lds, r, tid);
        }
    }
    // SEQ: one image in flight at a time (S round trips to memory, peak C + C / S registers) -- for the records that
    // leave no room for C (S - 1) / S staging registers next to themselves (float64 general matrices at 13, 14)
    template <bool SEQ = false>
    static __device__ __forceinline__ void get(char *smem, T (&r)[C], const T *__restrict__ g, int64_t avail, int tid)
    {
        if (avail >= (int64_t)64 * C) { // uniform: every wavefront but the last
            if constexpr (SEQ) {
                unsigned char *lds = reinterpret_cast<unsigned char *>(smem);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    V st[IT];
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const int q = tid + it * 64;
                        if (G::kNVec % 64 == 0 || q < G::kNVec)
                            st[it] = NFM_LDG(reinterpret_cast<const VG *>(g + (int64_t)s * L * C) + q);
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const int q = tid + it * 64;
                        if (G::kNVec % 64 == 0 || q < G::kNVec) *reinterpret_cast<V *>(lds + G::lds_off(q)) = st[it];
                    }
                    __syncthreads();
                    if (tid / L == s) pick(lds, r, tid);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            Stage st;
            issue(st, g, tid);
            __builtin_amdgcn_sched_barrier(0);
            land(smem, st, r, tid);
        } else {
            get_ragged(smem, r, g, avail, tid);
        }
    }
};

// how the records travel: through an LDS image of the wavefront's 64 records (whole-line accesses), or lane by lane
// with element-aligned 16-byte accesses.  The image of float64 orders 13..16 is 46-70 KiB a wavefront -- two or
// three wavefronts a CU -- so those go lane by lane (measured there: 0.60-0.65 of the roofline); everywhere else the
// image wins (float32 12x12 inverse: 0.35 lane by lane -- the eight 16-byte stores a lane makes to a 128-byte line
// are eight transactions at the L2 -- against 0.42 for the kernel this file replaces).
template <typename T, int N>
constexpr bool spd_tiled()
{
    return sizeof(T) == 4 || N <= 12;
}
template <typename T, int N, int OP>
constexpr size_t spd_lds_bytes()
{
    constexpr int ROUT = OP == SP_SOLVE ? N : OP == SP_INV ? sym_k(N) : OP == SP_INVDIAG ? N : 1;
    size_t b = roww::tile_lds_bytes<T, N, roww_op(OP), false, 4 * fallback_rows<T>()>();
    if (!spd_tiled<T, N>() && OP == SP_INV) {
        const size_t t = TileIO<T, sym_k(N), 32>::kLdsBytes; // SubOut<T, K, 2>
        b = b > t ? b : t;
    }
    if (spd_tiled<T, N>()) {
        const size_t tm = TileIO<T, sym_k(N), 64>::kLdsBytes, tr = TileIO<T, ROUT, 64>::kLdsBytes;
        const size_t tv = TileIO<T, N, 64>::kLdsBytes;
        b = b > tm ? b : tm;
        b = b > tr ? b : tr;
        b = b > tv ? b : tv;
    }
    return b;
}

template <typename T, int N, int OP>
__global__ __attribute__((amdgpu_waves_per_eu(1, spd_max_waves<T, N>()))) __launch_bounds__(64) void spd_kernel(const T *__restrict__ A, const T *__restrict__ B, T *__restrict__ O,
                                                 int64_t n, RowParams<T> p, int mark)
{
    constexpr int K = sym_k(N);
    constexpr int ROUT = OP == SP_SOLVE ? N : OP == SP_INV ? K : OP == SP_INVDIAG ? N : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t tile0 = (int64_t)blockIdx.x * 64;
    const int64_t i = tile0 + threadIdx.x;
    const bool live = i < n;
    const int64_t ii = live ? i : n - 1; // lanes past the end redo the last matrix and store nothing
    T m[K];
    T v[OP == SP_SOLVE ? N : 1];
    constexpr bool TILED = spd_tiled<T, N>();
    if constexpr (TILED) {
        // the wavefront streams its 64 records with whole-line 16-byte accesses through an LDS image and every lane
        // picks up its own (the vectors follow through the same bytes of LDS once the matrices are in registers)
        using IM = SubIn<T, K, 1>;
        using IV = SubIn<T, N, 1>;
        if (n - tile0 >= 64) { // uniform: every wavefront but the last
            typename IM::Stage sm;
            typename IV::Stage sv;
            IM::issue(sm, A + tile0 * K, (int)threadIdx.x);
            if constexpr (OP == SP_SOLVE) IV::issue(sv, B + tile0 * N, (int)threadIdx.x);
            IM::land(smem, sm, m, (int)threadIdx.x);
            if constexpr (OP == SP_SOLVE) IV::land(smem, sv, v, (int)threadIdx.x);
        } else {
            IM::get_ragged(smem, m, A + tile0 * K, (n - tile0) * K, (int)threadIdx.x);
            if constexpr (OP == SP_SOLVE) IV::get_ragged(smem, v, B + tile0 * N, (n - tile0) * N, (int)threadIdx.x);
        }
        if (!live) { // lanes past the end of the batch: the identity (positive definite: they do not trigger the fallback)
#pragma unroll
            for (int c = 0; c < K; ++c) m[c] = c < N ? T(1) : T(0);
        }
    } else {
        load_record<T, K>(A + ii * K, m);
        if constexpr (OP == SP_SOLVE) load_record<T, N>(B + ii * N, v);
        __builtin_amdgcn_sched_barrier(0); // every load issued before the first use
    }
    if constexpr (OP == SP_SOLVE) {
        if (p.has_eps) { // smoothing term on the diagonal (_impl/sym.py:356-357)
#pragma unroll
            for (int d = 0; d < N; ++d) m[d] += p.eps[d];
        }
    }
    T det;
    const bool ok = ldl_factor<T, N>(m, det);
    // the vote.  Matrices go to the fallback in groups of FM = what one of its passes works on; every other lane of
    // the wavefront finishes and stores as if nothing had happened (the groups of the bad matrices store nothing: an
    // in-place call must leave their input for the fallback)
    constexpr int FR = fallback_rows<T>(), FM = 4 * FR, NG = 64 / FM;
    const unsigned long long badl = __ballot(!ok);
    unsigned bad = 0;
    if (__builtin_expect(badl != 0, 0)) {
#pragma unroll
        for (int k = 0; k < NG; ++k) bad |= ((badl >> (k * FM)) & ((1ull << FM) - 1ull)) ? (1u << k) : 0u;
    }
    const bool mine = live && !((bad >> (threadIdx.x / FM)) & 1u); // this lane's result is stored by this path
    if (bad != (1u << NG) - 1u) { // (every group bad: nothing to finish or store here)
        // the result record of the lane: through the LDS image again (whole-line stores), or straight from the lane
        auto put = [&](auto &rec) {
            if constexpr (TILED) {
                SubOut<T, ROUT, 1>::template put<NG>(smem, rec, O + tile0 * ROUT, (n - tile0) * ROUT, (int)threadIdx.x, bad);
            } else if constexpr (OP == SP_INV) {
                // (the lane-by-lane kernels' inverse leaves through two LDS images: whole-line stores, no packing)
                SubOut<T, ROUT, 2>::template put<NG>(smem, rec, O + tile0 * ROUT, (n - tile0) * ROUT, (int)threadIdx.x, bad);
            } else {
                if (mine) store_record<T, ROUT>(O + i * ROUT, rec);
            }
        };
        if constexpr (OP == SP_SOLVE) {
            T x[N];
            ldl_solve<T, N>(m, v, x);
            put(x);
        } else if constexpr (OP == SP_DET) {
            if (mine) O[i] = det;
        } else if constexpr (OP == SP_INV) {
            ldl_inverse<T, N>(m);
            put(m);
        } else {
            T dg[N];
            ldl_inverse_diag<T, N>(m, dg);
            put(dg);
        }
    }
    if (__builtin_expect(bad == 0, 1)) return;
    if (mark) { // uniform: the output does not alias an input -- the bad groups are left to `redo_kernel`, which runs
                // right behind this one at the row-wave kernels' own occupancy; the MARK of a group is a NaN in the first
                // output element of its first matrix (every good group has written that element by now)
        if (threadIdx.x % FM == 0 && ((bad >> (threadIdx.x / FM)) & 1u) && live) O[i * ROUT] = (T)__builtin_nanf("");
        return;
    }
    // not positive definite somewhere in this wavefront: the pivoted elimination of nfm_rowwave on the groups that
    // hold such a matrix, FR rows per lane = FM matrices per pass (a pass reads its records before it writes: in-place
    // calls are safe).  float32: 4 rows per lane, up to 4 passes; float64: one row per lane, up to 16 passes -- 4 rows
    // of 16 doubles are 128 registers before anything else.
#pragma unroll 1
    for (int pass = 0; pass < NG; ++pass) {
        const int64_t m0 = tile0 + FM * pass;
        // (the lane id is made opaque per pass: everything a pass derives from it -- 64 LDS addresses and more --
        // would otherwise be hoisted out of this loop and live, or spilled, across its body)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (((bad >> pass) & 1u) && m0 < n) roww::roww_tile<T, N, roww_op(OP), FR, false, FM>(A, B, O, n, m0, p, smem, tid);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// the same kernel for ANY element strides of the compact operands and a second batch level (blockIdx.y):
// channel-first fields -- component c of matrix i at ptr[c * plane + i]: what the callers of these functions hold,
// and for a lane-per-matrix kernel the ideal layout, consecutive lanes read consecutive addresses --, padded or
// interleaved records, one vector for every matrix (stride 0).  Every lane addresses its own record element by
// element; the fallback gathers its group's records into the LDS image of nfm_rowwave and scatters the results
// (`roww_tile<..., STR = true>`).  Before this kernel these layouts went to the LDS-resident kernels of nfm_big.hpp
// (0.2-0.9 TB/s at orders 12 and 16) or through the facade's packing copy.
struct SOp {
    const void *ptr;
    int64_t so, si, sc; // element strides: outer batch level, inner batch level, component
};

template <typename T, int N, int OP>
__global__ __attribute__((amdgpu_waves_per_eu(1, spd_max_waves<T, N>()))) __launch_bounds__(64) void spd_strided_kernel(
    SOp a, SOp b, SOp o, int64_t n, RowParams<T> p)
{
    constexpr int K = sym_k(N);
    constexpr int ROUT = OP == SP_SOLVE ? N : OP == SP_INV ? K : OP == SP_INVDIAG ? N : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const T *A = static_cast<const T *>(a.ptr) + (int64_t)blockIdx.y * a.so;
    const T *B = OP == SP_SOLVE ? static_cast<const T *>(b.ptr) + (int64_t)blockIdx.y * b.so : nullptr;
    T *O = const_cast<T *>(static_cast<const T *>(o.ptr)) + (int64_t)blockIdx.y * o.so;
    const int64_t tile0 = (int64_t)blockIdx.x * 64;
    const int64_t i = tile0 + threadIdx.x;
    const bool live = i < n;
    const int64_t ii = live ? i : n - 1; // lanes past the end redo the last matrix and store nothing
    T m[K];
    T v[OP == SP_SOLVE ? N : 1];
    {
        const T *pa = A + ii * a.si;
#pragma unroll
        for (int c = 0; c < K; ++c) m[c] = pa[c * a.sc];
        if constexpr (OP == SP_SOLVE) {
            const T *pb = B + ii * b.si;
#pragma unroll
            for (int c = 0; c < N; ++c) v[c] = pb[c * b.sc];
        }
    }
    __builtin_amdgcn_sched_barrier(0); // every load issued before the first use
    if constexpr (OP == SP_SOLVE) {
        if (p.has_eps) {
#pragma unroll
            for (int d = 0; d < N; ++d) m[d] += p.eps[d];
        }
    }
    T det;
    const bool ok = ldl_factor<T, N>(m, det);
    constexpr int FR = fallback_rows<T>(), FM = 4 * FR, NG = 64 / FM;
    const unsigned long long badl = __ballot(!ok);
    unsigned bad = 0;
    if (__builtin_expect(badl != 0, 0)) {
#pragma unroll
        for (int k = 0; k < NG; ++k) bad |= ((badl >> (k * FM)) & ((1ull << FM) - 1ull)) ? (1u << k) : 0u;
    }
    const bool mine = live && !((bad >> (threadIdx.x / FM)) & 1u);
    if (bad != (1u << NG) - 1u) { // (every group bad: nothing to finish or store here)
        T *po = O + i * o.si;
        auto put = [&](auto &rec) {
            if (mine) {
#pragma unroll
                for (int c = 0; c < ROUT; ++c) po[c * o.sc] = rec[c];
            }
        };
        if constexpr (OP == SP_SOLVE) {
            T x[N];
            ldl_solve<T, N>(m, v, x);
            put(x);
        } else if constexpr (OP == SP_DET) {
            if (mine) po[0] = det;
        } else if constexpr (OP == SP_INV) {
            ldl_inverse<T, N>(m);
            put(m);
        } else {
            T dg[N];
            ldl_inverse_diag<T, N>(m, dg);
            put(dg);
        }
    }
    if (__builtin_expect(bad == 0, 1)) return;
    const roww::RowStrides st{a.si, a.sc, b.si, b.sc, o.si, o.sc};
#pragma unroll 1
    for (int pass = 0; pass < NG; ++pass) {
        const int64_t m0 = tile0 + FM * pass;
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid)); // (see spd_kernel)
        if (((bad >> pass) & 1u) && m0 < n)
            roww::roww_tile<T, N, roww_op(OP), FR, false, FM, true>(A, B, O, n, m0, p, smem, tid, st);
        __syncthreads();
    }
}

template <typename T, int N, int OP>
static int launch_strided(const SOp &a, const SOp &b, const SOp &o, int64_t no, int64_t n, const RowParams<T> &p, void *stream)
{
    constexpr size_t lds = roww::tile_lds_bytes<T, N, roww_op(OP), false, 4 * fallback_rows<T>()>();
    static_assert(lds <= 64 * 1024, "the fallback's tile must fit the default dynamic LDS limit");
    if (n == 0 || no == 0) return NFM_OK;
    const int64_t nblk = (n + 63) / 64;
    if (nblk > 0x7fffffffLL || no > 65535) return NFM_EFALLBACK_RW;
    hipLaunchKernelGGL((spd_strided_kernel<T, N, OP>), dim3((unsigned)nblk, (unsigned)no), dim3(64), lds,
                       static_cast<hipStream_t>(stream), a, b, o, n, p);
    return launch_status();
}

// y = [inp +/-] A v on compact records at any strides and two batch levels: what a channel-first field asks of
// `sym_matvec` / `sym_addmatvec` / `sym_submatvec` at orders 9..16 (before this kernel: two packing copies and the
// contiguous kernel).  The arithmetic is MatvecOp's (nfm_sym_ops.hpp: products and sums rounded separately, in the
// reference's order -- the result is bit-identical to the CPU restatement whatever the layout).
template <typename T, int N>
__global__ __launch_bounds__(64) void matvec_strided_kernel(SOp a, SOp b, SOp c, SOp o, int64_t n, int mode)
{
    constexpr int K = sym_k(N);
    using Op = MatvecOp<T, N, NFM_MAT_SYM>;
    const T *A = static_cast<const T *>(a.ptr) + (int64_t)blockIdx.y * a.so;
    const T *B = static_cast<const T *>(b.ptr) + (int64_t)blockIdx.y * b.so;
    const T *C = mode != 0 ? static_cast<const T *>(c.ptr) + (int64_t)blockIdx.y * c.so : nullptr;
    T *O = const_cast<T *>(static_cast<const T *>(o.ptr)) + (int64_t)blockIdx.y * o.so;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    T m[K], v[N], w[N], y[N];
    {
        const T *pa = A + i * a.si;
#pragma unroll
        for (int k = 0; k < K; ++k) m[k] = pa[k * a.sc];
        const T *pb = B + i * b.si;
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = pb[k * b.sc];
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] = T(0);
        if (mode != 0) {
            const T *pc = C + i * c.si;
#pragma unroll
            for (int k = 0; k < N; ++k) w[k] = pc[k * c.sc];
        }
    }
    __builtin_amdgcn_sched_barrier(0); // every load issued before the first use
    typename Op::Params prm{mode};
    Op::apply(m, v, w, y, prm);
    T *po = O + i * o.si;
#pragma unroll
    for (int k = 0; k < N; ++k) po[k * o.sc] = y[k];
}

// the same on contiguous operands with the records through LDS images (whole-line accesses): float64 orders 15, 16,
// whose lane-by-lane kernel (nfm_large.hip, 1088-byte records) ran at 0.49 of the roofline at 16x16 (0.68 here)
template <typename T, int N>
constexpr int matvec_subs()
{
    return TileIO<T, sym_k(N), 64>::kLdsBytes <= 40 * 1024 ? 1 : 2;
}
template <typename T, int N, bool INP> // INP: mode != 0 (a third operand: 2 x 32 more registers at 16x16 float64)
__global__ __attribute__((amdgpu_waves_per_eu(1, spd_max_waves<T, N>()))) __launch_bounds__(64) void matvec_tiled_kernel(
    const T *__restrict__ A, const T *__restrict__ B, const T *__restrict__ C, T *__restrict__ O, int64_t n, int mode)
{
    constexpr int K = sym_k(N), S = matvec_subs<T, N>();
    using Op = MatvecOp<T, N, NFM_MAT_SYM>;
    using IM = SubIn<T, K, S>;
    using IV = SubIn<T, N, 1>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t tile0 = (int64_t)blockIdx.x * 64;
    const int tid = (int)threadIdx.x;
    T m[K], v[N], w[INP ? N : 1], y[N];
    if (n - tile0 >= 64) { // uniform: every wavefront but the last
        typename IM::Stage sm;
        typename IV::Stage sv, sw;
        IM::issue(sm, A + tile0 * K, tid);
        IV::issue(sv, B + tile0 * N, tid);
        if constexpr (INP) IV::issue(sw, C + tile0 * N, tid);
        IM::land(smem, sm, m, tid);
        IV::land(smem, sv, v, tid);
        if constexpr (INP) IV::land(smem, sw, w, tid);
    } else {
        IM::get_ragged(smem, m, A + tile0 * K, (n - tile0) * K, tid);
        IV::get_ragged(smem, v, B + tile0 * N, (n - tile0) * N, tid);
        if constexpr (INP) IV::get_ragged(smem, w, C + tile0 * N, (n - tile0) * N, tid);
    }
    typename Op::Params prm{INP ? mode : 0};
    if constexpr (INP) {
        Op::apply(m, v, w, y, prm);
    } else {
        T none[N];
#pragma unroll
        for (int k = 0; k < N; ++k) none[k] = T(0);
        Op::apply(m, v, none, y, prm);
    }
    SubOut<T, N, 1>::template put<4>(smem, y, O + tile0 * N, (n - tile0) * N, tid, 0u);
}

template <typename T, int N>
static int launch_matvec_tiled(const void *a, const void *b, const void *c, void *o, int64_t n, int mode, void *stream)
{
    constexpr size_t lds = SubIn<T, sym_k(N), matvec_subs<T, N>()>::kLdsBytes;
    static_assert(lds <= 64 * 1024 && lds >= SubIn<T, N, 1>::kLdsBytes, "the images must fit the default dynamic LDS limit");
    if (n == 0) return NFM_OK;
    const int64_t nblk = (n + 63) / 64;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    if (mode != 0) {
        // (16x16 float64 with a third operand is 8 registers past the file: that one case keeps its kernel of nfm_large.hip)
        if constexpr (sizeof(T) == 8 && N == 16) return NFM_EFALLBACK_RW;
        else
            hipLaunchKernelGGL((matvec_tiled_kernel<T, N, true>), dim3((unsigned)nblk), dim3(64), lds,
                               static_cast<hipStream_t>(stream), static_cast<const T *>(a), static_cast<const T *>(b),
                               static_cast<const T *>(c), static_cast<T *>(o), n, mode);
    } else {
        hipLaunchKernelGGL((matvec_tiled_kernel<T, N, false>), dim3((unsigned)nblk), dim3(64), lds,
                           static_cast<hipStream_t>(stream), static_cast<const T *>(a), static_cast<const T *>(b),
                           static_cast<const T *>(c), static_cast<T *>(o), n, mode);
    }
    return launch_status();
}

template <typename T, int N>
static int launch_matvec_strided(const SOp &a, const SOp &b, const SOp &c, const SOp &o, int64_t no, int64_t n, int mode,
                                 void *stream)
{
    if (n == 0 || no == 0) return NFM_OK;
    const int64_t nblk = (n + 63) / 64;
    if (nblk > 0x7fffffffLL || no > 65535) return NFM_EFALLBACK_RW;
    hipLaunchKernelGGL((matvec_strided_kernel<T, N>), dim3((unsigned)nblk, (unsigned)no), dim3(64), 0,
                       static_cast<hipStream_t>(stream), a, b, c, o, n, mode);
    return launch_status();
}

template <typename T, int N>
static int call_strided(int op, const SOp &a, const SOp &b, const SOp &o, int64_t no, int64_t n, const RowParams<T> &p,
                        void *stream)
{
    switch (op) {
    case SP_SOLVE: return launch_strided<T, N, SP_SOLVE>(a, b, o, no, n, p, stream);
    case SP_INV:
        return launch_strided<T, N, SP_INV>(a, b, o, no, n, p, stream);
    case SP_INVDIAG: return launch_strided<T, N, SP_INVDIAG>(a, b, o, no, n, p, stream);
    case SP_DET: return launch_strided<T, N, SP_DET>(a, b, o, no, n, p, stream);
    default: return NFM_EINVAL;
    }
}

// ---------------------------------------------------------------------------------------------
// batchinv / batchdet of GENERAL matrices at orders 9..16: DIAGONAL PIVOTS FIRST (nfm_smallmat.hpp:
// gj_inverse_nopivot / lu_det_nopivot).  Same frame as above: one matrix per lane on N^2 registers, the elimination
// without row exchanges accepts the diagonal while it is within a factor 8 of the column maximum, the wavefront votes,
// and a wavefront with one matrix that needed an exchange redoes its 64 matrices with the pivoted row-wave kernels.
// A record is N^2 values (1 KiB at 16x16 float32): the image of 64 of them does not fit next to three other
// wavefronts, so the records come and go through LDS images of 64 / S of them at a time (SubIn / SubOut: whole-line
// accesses both ways).
// images per wavefront for the output of the general inverse: the smallest S whose image stays under 40 KiB
template <typename T, int N>
constexpr int gen_out_subs()
{
    return TileIO<T, N * N, 64>::kLdsBytes <= 40 * 1024 ? 1 : TileIO<T, N * N, 32>::kLdsBytes <= 40 * 1024 ? 2 : 4;
}
// float64 from order 12 up holds more than 256 values per lane: one wavefront per SIMD (see spd_max_waves)
template <typename T, int N>
constexpr int gen_max_waves()
{
    return (sizeof(T) == 8 && N >= 10) ? 1 : 8;
}
template <typename T, int N, int OP>
constexpr size_t gen_lds_bytes()
{
    size_t b = roww::tile_lds_bytes<T, N, roww_op(OP), false, 4 * fallback_rows<T>()>();
    const size_t t = SubOut<T, N * N, gen_out_subs<T, N>()>::kLdsBytes; // (the way in uses the same geometry)
    b = b > t ? b : t;
    return b;
}

template <typename T, int N, int OP>
__global__ __attribute__((amdgpu_waves_per_eu(1, gen_max_waves<T, N>()))) __launch_bounds__(64) void gen_kernel(
    const T *__restrict__ A, T *__restrict__ O, int64_t n, RowParams<T> p, int mark)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t tile0 = (int64_t)blockIdx.x * 64;
    const int64_t i = tile0 + threadIdx.x;
    const bool live = i < n;
    T f[N * N];
    SubIn<T, N * N, gen_out_subs<T, N>()>::template get<(sizeof(T) == 8 && N >= 13)>(smem, f, A + tile0 * (N * N),
                                                                                      (n - tile0) * (N * N), (int)threadIdx.x);
    if (!live) { // lanes past the end of the batch: the identity (they do not trigger the fallback)
#pragma unroll
        for (int c = 0; c < N * N; ++c) f[c] = (c % (N + 1) == 0) ? T(1) : T(0);
    }
    T a[N][N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) a[r][c] = f[r * N + c];
    constexpr int FR = fallback_rows<T>(), FM = 4 * FR, NG = 64 / FM;
    bool ok;
    T det = T(0);
    if constexpr (OP == SP_GDET) det = lu_det_nopivot<T, N>(a, ok);
    else gj_inverse_nopivot<T, N>(a, ok);
    // the vote: see spd_kernel
    const unsigned long long badl = __ballot(!ok);
    unsigned bad = 0;
    if (__builtin_expect(badl != 0, 0)) {
#pragma unroll
        for (int k = 0; k < NG; ++k) bad |= ((badl >> (k * FM)) & ((1ull << FM) - 1ull)) ? (1u << k) : 0u;
    }
    if constexpr (OP == SP_GDET) {
        if (live && !((bad >> (threadIdx.x / FM)) & 1u)) O[i] = det;
    } else {
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int c = 0; c < N; ++c) f[r * N + c] = a[r][c];
        SubOut<T, N * N, gen_out_subs<T, N>()>::template put<NG>(smem, f, O + tile0 * (N * N), (n - tile0) * (N * N),
                                                                 (int)threadIdx.x, bad);
    }
    if (__builtin_expect(bad == 0, 1)) return;
    if (mark) { // (see spd_kernel)
        constexpr int ROUT = OP == SP_GDET ? 1 : N * N;
        if (threadIdx.x % FM == 0 && ((bad >> (threadIdx.x / FM)) & 1u) && live) O[i * ROUT] = (T)__builtin_nanf("");
        return;
    }
    // a row exchange was needed somewhere in this wavefront: the pivoted elimination on the groups that hold such a matrix
#pragma unroll 1
    for (int pass = 0; pass < NG; ++pass) {
        const int64_t m0 = tile0 + FM * pass;
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid)); // (see spd_kernel)
        if (((bad >> pass) & 1u) && m0 < n) roww::roww_tile<T, N, roww_op(OP), FR, false, FM>(A, nullptr, O, n, m0, p, smem, tid);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// the second launch: the groups the first one marked, at the row-wave kernels' own register count (4-8 wavefronts per
// SIMD instead of the 1-2 of the kernels above, whose in-kernel fallback serves calls whose output aliases an input
// -- there is no place for a mark then).  A workgroup of 128 lanes (2 rows per lane, 16 matrices per tile) reads the
// marks of CH groups -- 16 scattered words; with nothing marked that is all the kernel does: n / 256 workgroups,
// 64 bytes per group of 16 matrices, 5-10 us at 2e6 matrices -- and runs `roww_tile` on the marked ones.
template <typename T, int N, int OP>
__global__ __launch_bounds__(128) void redo_kernel(const T *__restrict__ A, const T *__restrict__ B, T *__restrict__ O,
                                                   int64_t n, RowParams<T> p)
{
    constexpr int CH = 16; // groups per workgroup (64: a batch of indefinite matrices ran 64 tiles in a row per workgroup, too few in flight)
    constexpr int ROUT = OP == SP_SOLVE ? N : OP == SP_INV ? sym_k(N) : OP == SP_INVDIAG ? N
                         : OP == SP_GINV ? N * N : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned long long marks;
    const int64_t g0 = (int64_t)blockIdx.x * CH;
    const int64_t ngroups = (n + 15) / 16;
    if (threadIdx.x < 64) {
        bool marked = false;
        if (threadIdx.x < CH && g0 + threadIdx.x < ngroups) {
            const T m = O[(g0 + threadIdx.x) * 16 * ROUT];
            marked = m != m;
        }
        const unsigned long long b = __ballot(marked);
        if (threadIdx.x == 0) marks = b;
    }
    __syncthreads();
    const unsigned long long todo = marks;
    if (todo == 0) return;
#pragma unroll 1
    for (int k = 0; k < CH; ++k) {
        if (!((todo >> k) & 1ull)) continue; // uniform
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid)); // (see spd_kernel)
        roww::roww_tile<T, N, roww_op(OP), 2, false, 16>(A, B, O, n, (g0 + k) * 16, p, smem, tid);
        __syncthreads();
    }
}

template <typename T, int N, int OP>
static int launch_redo(const void *a, const void *b, void *o, int64_t n, const RowParams<T> &p, void *stream)
{
    constexpr size_t lds = roww::tile_lds_bytes<T, N, roww_op(OP), false, 16>();
    static_assert(lds <= 64 * 1024, "the tile must fit the default dynamic LDS limit");
    const int64_t nblk = ((n + 15) / 16 + 15) / 16;
    hipLaunchKernelGGL((redo_kernel<T, N, OP>), dim3((unsigned)nblk), dim3(128), lds, static_cast<hipStream_t>(stream),
                       static_cast<const T *>(a), static_cast<const T *>(b), static_cast<T *>(o), n, p);
    return launch_status();
}

// below this batch the second launch is not worth its 5-20 us: the wavefronts redo their bad groups themselves (a
// batch of 6e5 general 16x16 matrices with 2 % of its groups marked: 0.46 of the roofline with the second launch,
// 0.55 without)
constexpr int64_t kRedoMinBatch = 1 << 20;

// does the output overlap an input?  (byte ranges of contiguous operands)
static bool ranges_overlap(const void *x, size_t xb, const void *y, size_t yb)
{
    if (x == nullptr || y == nullptr) return false;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(x), b0 = reinterpret_cast<uintptr_t>(y);
    return a0 < b0 + yb && b0 < a0 + xb;
}

template <typename T, int N, int OP>
static int launch_gen(const void *a, void *o, int64_t n, void *stream)
{
    constexpr size_t lds = gen_lds_bytes<T, N, OP>();
    static_assert(lds <= 64 * 1024, "the images must fit the default dynamic LDS limit");
    if (n == 0) return NFM_OK;
    const int64_t nblk = (n + 63) / 64;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    RowParams<T> p{};
    constexpr int64_t ROUT = OP == SP_GDET ? 1 : N * N;
    const int mark = n >= kRedoMinBatch && !ranges_overlap(o, (size_t)n * ROUT * sizeof(T), a, (size_t)n * N * N * sizeof(T));
    hipLaunchKernelGGL((gen_kernel<T, N, OP>), dim3((unsigned)nblk), dim3(64), lds, static_cast<hipStream_t>(stream),
                       static_cast<const T *>(a), static_cast<T *>(o), n, p, mark);
    const int rc = launch_status();
    if (rc != NFM_OK || !mark) return rc;
    return launch_redo<T, N, OP>(a, nullptr, o, n, p, stream);
}

// the orders whose N^2 record (+ temporaries) the backend holds in a lane without scratch (scripts/survey_spd.sh)
template <typename T, int N, int OP>
constexpr bool gen_fits()
{
    return sizeof(T) == 4 || N <= (OP == SP_GINV ? NFM_GEN_F64_MAX_INV : NFM_GEN_F64_MAX_DET);
}

template <typename T, int N, int OP>
static int launch(const void *a, const void *b, void *o, int64_t n, const RowParams<T> &p, void *stream)
{
    constexpr size_t lds = spd_lds_bytes<T, N, OP>();
    static_assert(lds <= 64 * 1024, "the fallback's tile must fit the default dynamic LDS limit");
    if (n == 0) return NFM_OK;
    const int64_t nblk = (n + 63) / 64;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    constexpr int64_t K = sym_k(N), ROUT = OP == SP_SOLVE ? N : OP == SP_INV ? K : OP == SP_INVDIAG ? N : 1;
    const size_t ob = (size_t)n * ROUT * sizeof(T);
    const int mark = n >= kRedoMinBatch && !ranges_overlap(o, ob, a, (size_t)n * K * sizeof(T)) &&
                     !(OP == SP_SOLVE && ranges_overlap(o, ob, b, (size_t)n * N * sizeof(T)));
    hipLaunchKernelGGL((spd_kernel<T, N, OP>), dim3((unsigned)nblk), dim3(64), lds, static_cast<hipStream_t>(stream),
                       static_cast<const T *>(a), static_cast<const T *>(b), static_cast<T *>(o), n, p, mark);
    const int rc = launch_status();
    if (rc != NFM_OK || !mark) return rc;
    return launch_redo<T, N, OP>(a, b, o, n, p, stream);
}

template <typename T, int N>
static int call(int op, int64_t n, const void *a, const void *b, void *o, const RowParams<T> &p, void *stream)
{
    switch (op) {
    case SP_SOLVE: return launch<T, N, SP_SOLVE>(a, b, o, n, p, stream);
    case SP_INV: return launch<T, N, SP_INV>(a, b, o, n, p, stream);
    case SP_INVDIAG: return launch<T, N, SP_INVDIAG>(a, b, o, n, p, stream);
    case SP_DET: return launch<T, N, SP_DET>(a, b, o, n, p, stream);
    case SP_GINV:
        if constexpr (gen_fits<T, N, SP_GINV>()) return launch_gen<T, N, SP_GINV>(a, o, n, stream);
        else return NFM_EFALLBACK_RW;
    case SP_GDET:
        if constexpr (gen_fits<T, N, SP_GDET>()) return launch_gen<T, N, SP_GDET>(a, o, n, stream);
        else return NFM_EFALLBACK_RW;
    default: return NFM_EINVAL;
    }
}

} // namespace spd

#define NFM_SPD_F64 (NFM_SPD_PART / 4)
#if NFM_SPD_PART % 4 == 0 // (a literal: it is pasted into the function's name)
#define NFM_SPD_Q 0
#elif NFM_SPD_PART % 4 == 1
#define NFM_SPD_Q 1
#elif NFM_SPD_PART % 4 == 2
#define NFM_SPD_Q 2
#else
#define NFM_SPD_Q 3
#endif
#if NFM_SPD_F64
using TS = double;
#define NFM_SPD_NAME2(q) spd_call_f64_q##q
#else
using TS = float;
#define NFM_SPD_NAME2(q) spd_call_f32_q##q
#endif
#define NFM_SPD_NAME1(q) NFM_SPD_NAME2(q)

int NFM_SPD_NAME1(NFM_SPD_Q)(int op, int M, int64_t n, const void *a, const void *b, void *o, const double *eps, void *stream)
{
    spd::RowParams<TS> p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? (TS)eps[i] : TS(0);
    if (M == 9 + 2 * NFM_SPD_Q) return spd::call<TS, 9 + 2 * NFM_SPD_Q>(op, n, a, b, o, p, stream);
    if (M == 10 + 2 * NFM_SPD_Q) return spd::call<TS, 10 + 2 * NFM_SPD_Q>(op, n, a, b, o, p, stream);
    return NFM_EFALLBACK_RW;
}

#if NFM_SPD_F64
#define NFM_SPD_SNAME2(q) spd_call_strided_f64_q##q
#else
#define NFM_SPD_SNAME2(q) spd_call_strided_f32_q##q
#endif
#define NFM_SPD_SNAME1(q) NFM_SPD_SNAME2(q)
int NFM_SPD_SNAME1(NFM_SPD_Q)(int op, int M, int64_t no, int64_t n, const nfm_operand *a, const nfm_operand *b,
                              const nfm_operand *o, const double *eps, void *stream)
{
    spd::RowParams<TS> p;
    p.has_eps = eps != nullptr;
    for (int i = 0; i < NFM_MAX_DIM; ++i) p.eps[i] = (eps && i < M) ? (TS)eps[i] : TS(0);
    auto sop = [](const nfm_operand *x) {
        return x ? spd::SOp{x->ptr, x->stride_outer, x->stride_inner, x->stride_col} : spd::SOp{nullptr, 0, 0, 0};
    };
    const spd::SOp sa = sop(a), sb = sop(b), so = sop(o);
    if (M == 9 + 2 * NFM_SPD_Q) return spd::call_strided<TS, 9 + 2 * NFM_SPD_Q>(op, sa, sb, so, no, n, p, stream);
    if (M == 10 + 2 * NFM_SPD_Q) return spd::call_strided<TS, 10 + 2 * NFM_SPD_Q>(op, sa, sb, so, no, n, p, stream);
    return NFM_EFALLBACK_RW;
}

#if NFM_SPD_F64
#define NFM_SPD_MNAME2(q) spd_matvec_strided_f64_q##q
#else
#define NFM_SPD_MNAME2(q) spd_matvec_strided_f32_q##q
#endif
#define NFM_SPD_MNAME1(q) NFM_SPD_MNAME2(q)
int NFM_SPD_MNAME1(NFM_SPD_Q)(int M, int mode, int64_t no, int64_t n, const nfm_operand *a, const nfm_operand *b,
                              const nfm_operand *c, const nfm_operand *o, void *stream)
{
    auto sop = [](const nfm_operand *x) {
        return x ? spd::SOp{x->ptr, x->stride_outer, x->stride_inner, x->stride_col} : spd::SOp{nullptr, 0, 0, 0};
    };
    if (M == 9 + 2 * NFM_SPD_Q)
        return spd::launch_matvec_strided<TS, 9 + 2 * NFM_SPD_Q>(sop(a), sop(b), sop(c), sop(o), no, n, mode, stream);
    if (M == 10 + 2 * NFM_SPD_Q)
        return spd::launch_matvec_strided<TS, 10 + 2 * NFM_SPD_Q>(sop(a), sop(b), sop(c), sop(o), no, n, mode, stream);
    return NFM_EFALLBACK_RW;
}

#if NFM_SPD_F64
#define NFM_SPD_TNAME2(q) spd_matvec_tiled_f64_q##q
#else
#define NFM_SPD_TNAME2(q) spd_matvec_tiled_f32_q##q
#endif
#define NFM_SPD_TNAME1(q) NFM_SPD_TNAME2(q)
int NFM_SPD_TNAME1(NFM_SPD_Q)(int M, int mode, int64_t n, const void *a, const void *b, const void *c, void *o, void *stream)
{
#if NFM_SPD_F64 && NFM_SPD_Q >= 3 // float64 15, 16 only: every other case keeps its kernel of nfm_large.hip (0.72-0.76)
    if (M == 9 + 2 * NFM_SPD_Q) return spd::launch_matvec_tiled<TS, 9 + 2 * NFM_SPD_Q>(a, b, c, o, n, mode, stream);
    if (M == 10 + 2 * NFM_SPD_Q) return spd::launch_matvec_tiled<TS, 10 + 2 * NFM_SPD_Q>(a, b, c, o, n, mode, stream);
#endif
    return NFM_EFALLBACK_RW;
}

#if NFM_SPD_Q == 0
// the front end lives in the q0 object of each dtype
static bool spd_contig(const nfm_operand *o, int64_t rec, size_t elem)
{
    if (o == nullptr || o->ptr == nullptr) return false;
    if (reinterpret_cast<uintptr_t>(o->ptr) % elem != 0) return false;
    if (o->stride_inner != rec) return false;
    if (rec > 1 && o->stride_col != 1) return false;
    return true;
}

#if NFM_SPD_F64
#define NFM_SPD_CALL(q) spd_call_f64_q##q
#else
#define NFM_SPD_CALL(q) spd_call_f32_q##q
#endif
static int spd_dispatch(int op, int M, int64_t n, const void *a, const void *b, void *o, const double *eps, void *stream)
{
    // measurement knob (only under NFM_DEBUG, like the row-wave ones): NFM_SPD_OFF=1 sends everything to the pivoted
    // kernels, NFM_SPD_OFF=2 only the general matrices -- the A/B runs of scripts/bench_spd_ab.py
    static const int off = [] { const char *e = roww::dbg_env("NFM_SPD_OFF"); return e ? atoi(e) : 0; }();
    if (off == 1 || (off == 2 && (op == SP_GINV || op == SP_GDET))) return NFM_EFALLBACK_RW;
    switch ((M - 9) >> 1) {
    case 0: return NFM_SPD_CALL(0)(op, M, n, a, b, o, eps, stream);
    case 1: return NFM_SPD_CALL(1)(op, M, n, a, b, o, eps, stream);
    case 2: return NFM_SPD_CALL(2)(op, M, n, a, b, o, eps, stream);
    case 3: return NFM_SPD_CALL(3)(op, M, n, a, b, o, eps, stream);
    default: return NFM_EFALLBACK_RW;
    }
}

template <>
int Spd<TS>::sym_solve(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *out,
                       const double *eps, void *stream)
{
    const int K = M * (M + 1) / 2;
    if (M < 9 || M > 16 || !spd_contig(mat, K, sizeof(TS)) || !spd_contig(vec, M, sizeof(TS)) ||
        !spd_contig(out, M, sizeof(TS)))
        return NFM_EFALLBACK_RW;
    return spd_dispatch(SP_SOLVE, M, ni, mat->ptr, vec->ptr, out->ptr, eps, stream);
}

template <>
int Spd<TS>::sym_invert(int M, int diag_only, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    const int K = M * (M + 1) / 2;
    if (M < 9 || M > 16 || !spd_contig(mat, K, sizeof(TS)) || !spd_contig(out, diag_only ? M : K, sizeof(TS)))
        return NFM_EFALLBACK_RW;
    return spd_dispatch(diag_only ? SP_INVDIAG : SP_INV, M, ni, mat->ptr, nullptr, out->ptr, nullptr, stream);
}

#if NFM_SPD_F64
#define NFM_SPD_SCALL(q) spd_call_strided_f64_q##q
#else
#define NFM_SPD_SCALL(q) spd_call_strided_f32_q##q
#endif
static int spd_dispatch_strided(int op, int M, int64_t no, int64_t n, const nfm_operand *a, const nfm_operand *b,
                                const nfm_operand *o, const double *eps, void *stream)
{
    static const int off = [] { const char *e = roww::dbg_env("NFM_SPD_OFF"); return e ? atoi(e) : 0; }();
    if (off == 1 || off == 3) return NFM_EFALLBACK_RW; // (3: only the strided kernels off)
    if (M < 9 || M > 16 || a == nullptr || a->ptr == nullptr || o == nullptr || o->ptr == nullptr) return NFM_EFALLBACK_RW;
    switch ((M - 9) >> 1) {
    case 0: return NFM_SPD_SCALL(0)(op, M, no, n, a, b, o, eps, stream);
    case 1: return NFM_SPD_SCALL(1)(op, M, no, n, a, b, o, eps, stream);
    case 2: return NFM_SPD_SCALL(2)(op, M, no, n, a, b, o, eps, stream);
    case 3: return NFM_SPD_SCALL(3)(op, M, no, n, a, b, o, eps, stream);
    default: return NFM_EFALLBACK_RW;
    }
}

#if NFM_SPD_F64
#define NFM_SPD_MCALL(q) spd_matvec_strided_f64_q##q
#else
#define NFM_SPD_MCALL(q) spd_matvec_strided_f32_q##q
#endif
#if NFM_SPD_F64
#define NFM_SPD_TCALL(q) spd_matvec_tiled_f64_q##q
#else
#define NFM_SPD_TCALL(q) spd_matvec_tiled_f32_q##q
#endif
template <>
int Spd<TS>::sym_matvec(int M, int mode, int64_t ni, const nfm_operand *mat, const nfm_operand *vec, const nfm_operand *inp,
                        const nfm_operand *out, void *stream)
{
    static const int off = [] { const char *e = roww::dbg_env("NFM_SPD_OFF"); return e ? atoi(e) : 0; }();
    const int K = M * (M + 1) / 2;
    // (13 and 14 measured level with their kernels of nfm_large.hip: 0.68-0.70 here, 0.72 there)
    if (off == 1 || sizeof(TS) != 8 || M < 15 || M > 16 || !spd_contig(mat, K, sizeof(TS)) ||
        !spd_contig(vec, M, sizeof(TS)) || !spd_contig(out, M, sizeof(TS)) || (mode != 0 && !spd_contig(inp, M, sizeof(TS))))
        return NFM_EFALLBACK_RW;
    const void *c = mode != 0 ? inp->ptr : nullptr;
    switch ((M - 9) >> 1) {
    case 3: return NFM_SPD_TCALL(3)(M, mode, ni, mat->ptr, vec->ptr, c, out->ptr, stream);
    default: return NFM_EFALLBACK_RW;
    }
}

template <>
int Spd<TS>::sym_matvec_strided(int M, int mode, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                                const nfm_operand *inp, const nfm_operand *out, void *stream)
{
    static const int off = [] { const char *e = roww::dbg_env("NFM_SPD_OFF"); return e ? atoi(e) : 0; }();
    if (off == 1 || off == 3) return NFM_EFALLBACK_RW;
    if (M < 9 || M > 16 || mat == nullptr || mat->ptr == nullptr || vec == nullptr || vec->ptr == nullptr ||
        out == nullptr || out->ptr == nullptr || (mode != 0 && (inp == nullptr || inp->ptr == nullptr)))
        return NFM_EFALLBACK_RW;
    switch ((M - 9) >> 1) {
    case 0: return NFM_SPD_MCALL(0)(M, mode, no, ni, mat, vec, inp, out, stream);
    case 1: return NFM_SPD_MCALL(1)(M, mode, no, ni, mat, vec, inp, out, stream);
    case 2: return NFM_SPD_MCALL(2)(M, mode, no, ni, mat, vec, inp, out, stream);
    case 3: return NFM_SPD_MCALL(3)(M, mode, no, ni, mat, vec, inp, out, stream);
    default: return NFM_EFALLBACK_RW;
    }
}

template <>
int Spd<TS>::sym_solve_strided(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *vec,
                               const nfm_operand *out, const double *eps, void *stream)
{
    if (vec == nullptr || vec->ptr == nullptr) return NFM_EFALLBACK_RW;
    return spd_dispatch_strided(SP_SOLVE, M, no, ni, mat, vec, out, eps, stream);
}

template <>
int Spd<TS>::sym_invert_strided(int M, int diag_only, int64_t no, int64_t ni, const nfm_operand *mat,
                                const nfm_operand *out, void *stream)
{
    return spd_dispatch_strided(diag_only ? SP_INVDIAG : SP_INV, M, no, ni, mat, nullptr, out, nullptr, stream);
}

template <>
int Spd<TS>::sym_det_strided(int M, int64_t no, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    return spd_dispatch_strided(SP_DET, M, no, ni, mat, nullptr, out, nullptr, stream);
}

template <>
int Spd<TS>::batch_inv(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    const int64_t rec = (int64_t)Nn * Nn;
    auto full = [&](const nfm_operand *o) { return spd_contig(o, rec, sizeof(TS)) && o->stride_row == Nn; };
    if (Nn < 9 || Nn > 16 || !full(a) || !full(out)) return NFM_EFALLBACK_RW;
    return spd_dispatch(SP_GINV, Nn, ni, a->ptr, nullptr, out->ptr, nullptr, stream);
}

template <>
int Spd<TS>::batch_det(int Nn, int64_t ni, const nfm_operand *a, const nfm_operand *out, void *stream)
{
    const int64_t rec = (int64_t)Nn * Nn;
    if (Nn < 9 || Nn > 16 || !spd_contig(a, rec, sizeof(TS)) || a->stride_row != Nn || out == nullptr ||
        out->ptr == nullptr || out->stride_inner != 1)
        return NFM_EFALLBACK_RW;
    return spd_dispatch(SP_GDET, Nn, ni, a->ptr, nullptr, out->ptr, nullptr, stream);
}

template <>
int Spd<TS>::sym_det(int M, int64_t ni, const nfm_operand *mat, const nfm_operand *out, void *stream)
{
    const int K = M * (M + 1) / 2;
    if (M < 9 || M > 16 || !spd_contig(mat, K, sizeof(TS)) || out == nullptr || out->ptr == nullptr ||
        out->stride_inner != 1)
        return NFM_EFALLBACK_RW;
    return spd_dispatch(SP_DET, M, ni, mat->ptr, nullptr, out->ptr, nullptr, stream);
}
#endif

} // namespace nfm

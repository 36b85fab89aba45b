// nfm_record_kernel.hpp -- the one kernel skeleton every lane-per-matrix op uses.
//
//   rec_kernel<T, Op>: up to three input records (A, B, C) and one output record
//   per batch element; Op::apply() is the in-register arithmetic.  Each operand
//   independently takes the LDS-transposed path (contiguous batch-major storage)
//   or the direct per-lane path (SoA / broadcast / strided); the choice is a
//   wave-uniform kernel argument, so there is no divergence.
//
// Timeline of a workgroup (TILE lanes = TILE batch elements):
//   1. issue every 16-byte global load of the tile for all tiled inputs (no waits
//      in between -> all of the tile's bytes are in flight together);
//   2. direct loads for the non-tiled inputs;
//   3. park the staged vectors in LDS, one barrier, every lane picks up its records;
//   4. Op::apply in registers;
//   5. the output goes back through LDS (barrier) and out with 16-byte stores, or
//      directly when it is not batch-major contiguous.
#pragma once
#include "nfm_common.hpp"

namespace nfm {

// shape of one operand record: R x Cc elements (R == 1 for vectors / compact storage)
template <int R_, int C_>
struct Rec {
    static constexpr int R = R_;
    static constexpr int Cc = C_;
    static constexpr int C = R_ * C_;
    static constexpr bool used = C > 0;
    static constexpr int Cs = C > 0 ? C : 1; // array extent (no zero-size arrays)
};
using NoRec = Rec<0, 0>;

// how an operand is moved (wave-uniform kernel argument Opnd::tiled)
enum { MODE_STRIDED = 0, MODE_TILED = 1, MODE_VEC = 2 };

template <typename T, class R, int TILE>
struct RecIO {
    using IO = TileIO<T, R::Cs, TILE>;
    // Records of exactly 4, 8 or 16 bytes that sit back to back in memory need no
    // transpose: one b32/b64/b128 access per lane is already a fully coalesced wave
    // access (measured on the 4x4 solve: the vec/out records moved this way instead of
    // through LDS take the kernel from 5.3 to 6.2 TB/s).
    static constexpr int RB = R::Cs * (int)sizeof(T);
    static constexpr bool can_vec = R::used && (R::C == 1 || R::C == 2 || R::C == 4) && RB <= 16;
    static constexpr bool can_tile = R::used && R::C > 1 && !can_vec;
    static constexpr int lds = can_tile ? IO::kLdsBytes : 0;
    static constexpr int pref = can_vec ? MODE_VEC : (can_tile ? MODE_TILED : MODE_STRIDED);
};

template <typename T, int C>
struct PackOf {
    typedef T type __attribute__((ext_vector_type(C)));
};
template <typename T>
struct PackOf<T, 1> {
    typedef T type;
};

template <typename T, class R>
__device__ __forceinline__ void rec_vec_load(const Opnd &op, int64_t o, int64_t i, bool valid, T (&r)[R::Cs])
{
    using P = typename PackOf<T, R::Cs>::type;
    const P *p = reinterpret_cast<const P *>(reinterpret_cast<const T *>(op.ptr) + o * op.so + i * R::Cs);
    if (valid) {
        const P v = __builtin_nontemporal_load(p);
        if constexpr (R::Cs == 1) r[0] = v;
        else {
#pragma unroll
            for (int c = 0; c < R::Cs; ++c) r[c] = v[c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < R::Cs; ++c) r[c] = T(1);
    }
}

template <typename T, class R>
__device__ __forceinline__ void rec_vec_store(const Opnd &op, int64_t o, int64_t i, bool valid,
                                              const T (&r)[R::Cs])
{
    using P = typename PackOf<T, R::Cs>::type;
    P *p = reinterpret_cast<P *>(reinterpret_cast<T *>(op.ptr) + o * op.so + i * R::Cs);
    if (valid) {
        P v;
        if constexpr (R::Cs == 1) v = r[0];
        else {
#pragma unroll
            for (int c = 0; c < R::Cs; ++c) v[c] = r[c];
        }
        __builtin_nontemporal_store(v, p);
    }
}

template <typename T, class Op>
struct RecLayout {
    static constexpr int TILE = Op::TILE;
    using A = RecIO<T, typename Op::RA, TILE>;
    using B = RecIO<T, typename Op::RB, TILE>;
    using C = RecIO<T, typename Op::RC, TILE>;
    using O = RecIO<T, typename Op::RO, TILE>;
    static constexpr int offA = 0;
    static constexpr int offB = offA + A::lds;
    static constexpr int offC = offB + B::lds;
    static constexpr int in_end = offC + C::lds;
    // The output image may reuse an input image of identical geometry: every lane
    // then overwrites only the record it has itself already read.
    static constexpr int COUT = Op::RO::C;
    static constexpr bool aliasA = A::can_tile && Op::RA::C == COUT;
    static constexpr bool aliasB = !aliasA && B::can_tile && Op::RB::C == COUT;
    static constexpr bool aliasC = !aliasA && !aliasB && C::can_tile && Op::RC::C == COUT;
    static constexpr int offO = aliasA ? offA : (aliasB ? offB : (aliasC ? offC : in_end));
    static constexpr int total = (aliasA || aliasB || aliasC) ? in_end : in_end + O::lds;
};

template <typename T, class R>
__device__ __forceinline__ void rec_direct_load(const Opnd &op, int64_t o, int64_t i, bool valid, T (&r)[R::Cs])
{
    const T *p = reinterpret_cast<const T *>(op.ptr) + o * op.so + i * op.si;
#pragma unroll
    for (int a = 0; a < R::R; ++a)
#pragma unroll
        for (int b = 0; b < R::Cc; ++b) r[a * R::Cc + b] = valid ? p[a * op.sr + b * op.sc] : T(1);
}

template <typename T, class R>
__device__ __forceinline__ void rec_direct_store(const Opnd &op, int64_t o, int64_t i, bool valid,
                                                 const T (&r)[R::Cs])
{
    T *p = reinterpret_cast<T *>(op.ptr) + o * op.so + i * op.si;
    if (valid) {
#pragma unroll
        for (int a = 0; a < R::R; ++a)
#pragma unroll
            for (int b = 0; b < R::Cc; ++b) p[a * op.sr + b * op.sc] = r[a * R::Cc + b];
    }
}

// FAST = every operand is a contiguous batch-major block (the default torch layout):
// the movement mode of each operand is then a compile-time constant (packed access for
// 4/8/16-byte records, LDS transpose for the rest), which removes every mode branch and
// a third of the VGPRs (92 -> 60 on the 4x4 solve, 5 -> 8 waves per SIMD).  FAST = false
// keeps the wave-uniform run-time modes for SoA / broadcast / strided operands.
template <typename T, class Op, bool FAST>
__global__ __launch_bounds__(Op::TILE) void rec_kernel(Opnd a, Opnd b, Opnd c, Opnd out, int64_t n_inner,
                                                       typename Op::Params prm)
{
    using L = RecLayout<T, Op>;
    using RA = typename Op::RA;
    using RB = typename Op::RB;
    using RC = typename Op::RC;
    using RO = typename Op::RO;
    constexpr int TILE = Op::TILE;
    extern __shared__ __align__(16) unsigned char smem[];

    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int64_t i = tile0 + threadIdx.x;
    const int64_t o = blockIdx.y;
    const bool valid = i < n_inner;
    const int64_t left = n_inner - tile0; // batch elements from the tile start to the end

    T ra[RA::Cs], rb[RB::Cs], rc[RC::Cs], ro[RO::Cs];
    typename L::A::IO::Stage sa;
    typename L::B::IO::Stage sb;
    typename L::C::IO::Stage sc;

    const int ma = FAST ? L::A::pref : a.tiled, mb = FAST ? L::B::pref : b.tiled;
    const int mc = FAST ? L::C::pref : c.tiled, mo = FAST ? L::O::pref : out.tiled;
    const bool ta = L::A::can_tile && ma == MODE_TILED;
    const bool tb = L::B::can_tile && mb == MODE_TILED;
    const bool tc = L::C::can_tile && mc == MODE_TILED;
    const bool to = L::O::can_tile && mo == MODE_TILED;
    const bool va = L::A::can_vec && ma == MODE_VEC;
    const bool vb = L::B::can_vec && mb == MODE_VEC;
    const bool vc = L::C::can_vec && mc == MODE_VEC;
    const bool vo = L::O::can_vec && mo == MODE_VEC;
    const bool use_c = RC::used && c.ptr != nullptr;

    // 1. all tiled global loads in flight
    if constexpr (L::A::can_tile)
        if (ta) L::A::IO::issue(reinterpret_cast<const T *>(a.ptr) + tile0 * RA::C, left * RA::C, sa);
    if constexpr (L::B::can_tile)
        if (tb) L::B::IO::issue(reinterpret_cast<const T *>(b.ptr) + tile0 * RB::C, left * RB::C, sb);
    if constexpr (L::C::can_tile)
        if (tc && use_c) L::C::IO::issue(reinterpret_cast<const T *>(c.ptr) + tile0 * RC::C, left * RC::C, sc);
    // 2. per-lane loads: one packed access for back-to-back 4/8/16-byte records,
    //    element-wise strided access for everything else
    if constexpr (RA::used) {
        if (va) {
            if constexpr (L::A::can_vec) rec_vec_load<T, RA>(a, o, i, valid, ra);
        } else if (!ta) rec_direct_load<T, RA>(a, o, i, valid, ra);
    }
    if constexpr (RB::used) {
        if (vb) {
            if constexpr (L::B::can_vec) rec_vec_load<T, RB>(b, o, i, valid, rb);
        } else if (!tb) rec_direct_load<T, RB>(b, o, i, valid, rb);
    }
    if constexpr (RC::used) {
        if (vc) {
            if constexpr (L::C::can_vec) rec_vec_load<T, RC>(c, o, i, valid && use_c, rc);
        } else if (!tc) rec_direct_load<T, RC>(c, o, i, valid && use_c, rc);
    }
    // 3. LDS transpose
    if constexpr (L::A::can_tile)
        if (ta) L::A::IO::commit(smem + L::offA, sa);
    if constexpr (L::B::can_tile)
        if (tb) L::B::IO::commit(smem + L::offB, sb);
    if constexpr (L::C::can_tile)
        if (tc && use_c) L::C::IO::commit(smem + L::offC, sc);
    if (ta || tb || tc) __syncthreads();
    if constexpr (L::A::can_tile)
        if (ta) L::A::IO::read_own(smem + L::offA, ra);
    if constexpr (L::B::can_tile)
        if (tb) L::B::IO::read_own(smem + L::offB, rb);
    if constexpr (L::C::can_tile)
        if (tc && use_c) L::C::IO::read_own(smem + L::offC, rc);

    // 4. arithmetic
    Op::apply(ra, rb, rc, ro, prm);

    // 5. output
    if constexpr (L::O::can_tile) {
        if (to) {
            L::O::IO::write_own(smem + L::offO, ro);
            __syncthreads();
            L::O::IO::flush(reinterpret_cast<T *>(out.ptr) + tile0 * RO::C, left * RO::C, smem + L::offO);
            return;
        }
    }
    if constexpr (L::O::can_vec) {
        if (vo) {
            rec_vec_store<T, RO>(out, o, i, valid, ro);
            return;
        }
    }
    rec_direct_store<T, RO>(out, o, i, valid, ro);
}

// Host launcher.  Operands that qualify are tiled; `force_direct` (testing/benchmark
// knob) sends everything down the per-lane path.
template <typename T, class Op>
int rec_launch(const nfm_operand *a, const nfm_operand *b, const nfm_operand *c, const nfm_operand *out,
               int64_t n_outer, int64_t n_inner, const typename Op::Params &prm, void *stream)
{
    using L = RecLayout<T, Op>;
    using RA = typename Op::RA;
    using RB = typename Op::RB;
    using RC = typename Op::RC;
    using RO = typename Op::RO;
    if (n_outer == 0 || n_inner == 0) return NFM_OK;
    nfm_operand none = {nullptr, 0, 0, 0, 0};
    if (a == nullptr) a = &none;
    if (b == nullptr) b = &none;
    if (c == nullptr) c = &none;
    const bool ta = L::A::can_tile && tile_ok(a, RA::C, RA::R, RA::Cc, n_outer, n_inner, sizeof(T));
    const bool tb = L::B::can_tile && tile_ok(b, RB::C, RB::R, RB::Cc, n_outer, n_inner, sizeof(T));
    const bool tc = L::C::can_tile && tile_ok(c, RC::C, RC::R, RC::Cc, n_outer, n_inner, sizeof(T));
    const bool to = L::O::can_tile && tile_ok(out, RO::C, RO::R, RO::Cc, n_outer, n_inner, sizeof(T));
    const int ma = ta ? MODE_TILED : (L::A::can_vec && vec_ok(a, RA::C, RA::R, RA::Cc, n_outer, sizeof(T)) ? MODE_VEC : MODE_STRIDED);
    const int mb = tb ? MODE_TILED : (L::B::can_vec && vec_ok(b, RB::C, RB::R, RB::Cc, n_outer, sizeof(T)) ? MODE_VEC : MODE_STRIDED);
    const int mc = tc ? MODE_TILED : (L::C::can_vec && vec_ok(c, RC::C, RC::R, RC::Cc, n_outer, sizeof(T)) ? MODE_VEC : MODE_STRIDED);
    const int mo = to ? MODE_TILED : (L::O::can_vec && vec_ok(out, RO::C, RO::R, RO::Cc, n_outer, sizeof(T)) ? MODE_VEC : MODE_STRIDED);
    const bool any = ta || tb || tc || to;
    // FAST path: every used operand in its preferred mode (an absent C operand is fine)
    const bool c_absent = c->ptr == nullptr;
    const bool fast = n_outer == 1 && (!RA::used || ma == L::A::pref) && (!RB::used || mb == L::B::pref) &&
                      (!RC::used || c_absent || mc == L::C::pref) && mo == L::O::pref &&
                      L::A::pref != MODE_STRIDED && L::O::pref != MODE_STRIDED &&
                      (!RB::used || L::B::pref != MODE_STRIDED) && (!RC::used || L::C::pref != MODE_STRIDED);
    const int64_t nblk = (n_inner + Op::TILE - 1) / Op::TILE;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    dim3 grid((unsigned)nblk, (unsigned)n_outer, 1), block(Op::TILE, 1, 1);
    const size_t lds = any ? (size_t)L::total : 0;
    static bool attr_done = false; // > 64 KiB dynamic LDS needs an opt-in, once per kernel
    if (L::total > 64 * 1024 && !attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&rec_kernel<T, Op, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, L::total);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&rec_kernel<T, Op, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, L::total);
        attr_done = true;
    }
    if (fast)
        hipLaunchKernelGGL((rec_kernel<T, Op, true>), grid, block, (size_t)L::total,
                           static_cast<hipStream_t>(stream), make_opnd(a, ma), make_opnd(b, mb), make_opnd(c, mc),
                           make_opnd(out, mo), n_inner, prm);
    else
        hipLaunchKernelGGL((rec_kernel<T, Op, false>), grid, block, lds, static_cast<hipStream_t>(stream),
                           make_opnd(a, ma), make_opnd(b, mb), make_opnd(c, mc), make_opnd(out, mo), n_inner, prm);
    return launch_status();
}

} // namespace nfm

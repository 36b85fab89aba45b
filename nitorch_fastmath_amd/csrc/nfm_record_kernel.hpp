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

template <typename T, class R, int TILE>
struct RecIO {
    using IO = TileIO<T, R::Cs, TILE>;
    // a one-element record is already contiguous across lanes: never tiled
    static constexpr bool can_tile = R::used && R::C > 1;
    static constexpr int lds = can_tile ? IO::kLdsBytes : 0;
};

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

template <typename T, class Op>
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

    const bool ta = L::A::can_tile && a.tiled;
    const bool tb = L::B::can_tile && b.tiled;
    const bool tc = L::C::can_tile && c.tiled;
    const bool to = L::O::can_tile && out.tiled;
    const bool use_c = RC::used && c.ptr != nullptr;

    // 1. all tiled global loads in flight
    if constexpr (L::A::can_tile)
        if (ta) L::A::IO::issue(reinterpret_cast<const T *>(a.ptr) + tile0 * RA::C, left * RA::C, sa);
    if constexpr (L::B::can_tile)
        if (tb) L::B::IO::issue(reinterpret_cast<const T *>(b.ptr) + tile0 * RB::C, left * RB::C, sb);
    if constexpr (L::C::can_tile)
        if (tc && use_c) L::C::IO::issue(reinterpret_cast<const T *>(c.ptr) + tile0 * RC::C, left * RC::C, sc);
    // 2. direct loads
    if constexpr (RA::used)
        if (!ta) rec_direct_load<T, RA>(a, o, i, valid, ra);
    if constexpr (RB::used)
        if (!tb) rec_direct_load<T, RB>(b, o, i, valid, rb);
    if constexpr (RC::used)
        if (!tc) rec_direct_load<T, RC>(c, o, i, valid && use_c, rc);
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
    const bool any = ta || tb || tc || to;
    const int64_t nblk = (n_inner + Op::TILE - 1) / Op::TILE;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    dim3 grid((unsigned)nblk, (unsigned)n_outer, 1), block(Op::TILE, 1, 1);
    const size_t lds = any ? (size_t)L::total : 0;
    static bool attr_done = false; // > 64 KiB dynamic LDS needs an opt-in, once per kernel
    if (L::total > 64 * 1024 && !attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&rec_kernel<T, Op>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, L::total);
        attr_done = true;
    }
    hipLaunchKernelGGL((rec_kernel<T, Op>), grid, block, lds, static_cast<hipStream_t>(stream),
                       make_opnd(a, ta), make_opnd(b, tb), make_opnd(c, tc), make_opnd(out, to), n_inner, prm);
    return launch_status();
}

} // namespace nfm

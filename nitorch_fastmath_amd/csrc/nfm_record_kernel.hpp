// nfm_record_kernel.hpp -- the one kernel skeleton every lane-per-matrix op uses.
//
//   rec_kernel<T, Op, KIND>: up to three input records (A, B, C) and one output record
//   per batch element; Op::apply() is the in-register arithmetic.  Each operand
//   independently takes one of five movement modes (LDS-transposed AoS tile, packed
//   per-lane access, LDS component-major tile, strided per-lane access); the choice is a
//   wave-uniform kernel argument (or a compile-time constant in the FAST kernel), so
//   there is no divergence.
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
#include <type_traits>
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
// MODE_PACKED: records that are contiguous INSIDE but sit at any batch stride (every k-th record
// of a field, rows of a cropped field, a padded record, a broadcast operand): the lane fetches
// its record with ceil(C / 4) element-aligned 16-byte accesses instead of C scalar ones.
// MODE_PACKED2: the same for INPUT records whose elements are two apart (the real parts of a complex
// field, one of two interleaved fields): the lane fetches the covering span and keeps every other
// element.  MODE_PACKEDT: full matrices stored transposed (a.mT): packed accesses, the transpose is a
// renaming of registers.
enum { MODE_STRIDED = 0, MODE_TILED = 1, MODE_VEC = 2, MODE_SOA = 3, MODE_PACKED = 4, MODE_PACKED2 = 5,
       MODE_PACKEDT = 6 };

template <typename T, class R, int TILE>
struct RecIO {
    using IO = TileIO<T, R::Cs, TILE>;
    using SO = SoaIO<T, (R::used ? R::R : 1), (R::used ? R::Cc : 1), TILE>;
    static constexpr bool can_soa = R::used && R::C > 1;
    static constexpr int soa_lds = can_soa ? SO::kLdsBytes : 0;
    // Records of exactly 4, 8 or 16 bytes that sit back to back in memory need no
    // transpose: one b32/b64/b128 access per lane is already a fully coalesced wave
    // access (measured on the 4x4 solve: the vec/out records moved this way instead of
    // through LDS take the kernel from 5.3 to 6.2 TB/s).
    static constexpr int RB = R::Cs * (int)sizeof(T);
    static constexpr bool can_vec = R::used && (R::C == 1 || R::C == 2 || R::C == 4) && RB <= 16;
    static constexpr bool can_tile = R::used && R::C > 1 && !can_vec;
    static constexpr int lds = can_tile ? IO::kLdsBytes : 0;
    // region size that fits either LDS image of this operand (AoS-transposed or SoA)
    static constexpr int lds_any = lds > soa_lds ? lds : soa_lds;
    static constexpr int pref = can_vec ? MODE_VEC : (can_tile ? MODE_TILED : MODE_STRIDED);
    // preferred mode when the caller holds component-major (channel-first) fields: every
    // multi-component operand through the SoA image, scalars packed
    static constexpr int pref_soa = can_soa ? MODE_SOA : (R::used && R::C == 1 ? MODE_VEC : MODE_STRIDED);
};

// element-aligned on purpose: a packed 8 / 16-byte global access only needs dword alignment on
// gfx950 (same instruction), so records at any element offset keep the one-access-per-lane path
template <typename T, int C>
struct PackOf {
    typedef T type __attribute__((ext_vector_type(C), aligned(sizeof(T))));
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
        const P v = NFM_LDG(p);
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
        NFM_STG(v, p);
    }
}

template <typename T, class Op, int TILE_ = Op::TILE>
struct RecLayout {
    static constexpr int TILE = TILE_;
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
    // layout of the run-time-mode (non-FAST) kernel: every region fits either image and the
    // output has a region of its own (an SoA output image never aliases an AoS input image)
    static constexpr int gA = 0;
    static constexpr int gB = gA + A::lds_any;
    static constexpr int gC = gB + B::lds_any;
    static constexpr int gO = gC + C::lds_any;
    static constexpr int gtotal = gO + O::lds_any;
};

// One record of C back-to-back elements at p, moved with the widest element-aligned accesses:
// whole 16-byte vectors, then an 8-byte pair (float), then a single element.  Plain (cached)
// accesses on purpose: when the batch stride leaves gaps between records, the 2-3 accesses of a
// lane -- and the neighbouring lanes' -- land in the same 128-byte lines one after the other.
template <typename T, int C>
__device__ __forceinline__ void packed_load(const T *p, T (&r)[C])
{
    using VG = typename VecOf<T>::gtype;
    constexpr int kVec = VecOf<T>::N;
#pragma unroll
    for (int s = 0; s < C / kVec; ++s) {
        const VG v = *reinterpret_cast<const VG *>(p + s * kVec);
#pragma unroll
        for (int k = 0; k < kVec; ++k) r[s * kVec + k] = v[k];
    }
    constexpr int done = (C / kVec) * kVec;
    if constexpr (sizeof(T) == 4 && C - done >= 2) {
        using P2 = typename PackOf<T, 2>::type;
        const P2 v = *reinterpret_cast<const P2 *>(p + done);
        r[done] = v[0];
        r[done + 1] = v[1];
        if constexpr (C - done == 3) r[done + 2] = p[done + 2];
    } else if constexpr (C - done == 1) {
        r[done] = p[done];
    }
}

template <typename T, int C>
__device__ __forceinline__ void packed_store(T *p, const T (&r)[C])
{
    using VG = typename VecOf<T>::gtype;
    constexpr int kVec = VecOf<T>::N;
#pragma unroll
    for (int s = 0; s < C / kVec; ++s) {
        VG v;
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] = r[s * kVec + k];
        *reinterpret_cast<VG *>(p + s * kVec) = v;
    }
    constexpr int done = (C / kVec) * kVec;
    if constexpr (sizeof(T) == 4 && C - done >= 2) {
        using P2 = typename PackOf<T, 2>::type;
        P2 v;
        v[0] = r[done];
        v[1] = r[done + 1];
        *reinterpret_cast<P2 *>(p + done) = v;
        if constexpr (C - done == 3) p[done + 2] = r[done + 2];
    } else if constexpr (C - done == 1) {
        p[done] = r[done];
    }
}

template <typename T, class R>
__device__ __forceinline__ void rec_direct_load(const Opnd &op, int mode, int64_t o, int64_t i, bool valid,
                                                T (&r)[R::Cs])
{
    const T *p = reinterpret_cast<const T *>(op.ptr) + o * op.so + i * op.si;
    if (!valid) { // lanes past the end of the batch compute on ones
#pragma unroll
        for (int c = 0; c < R::Cs; ++c) r[c] = T(1);
        return;
    }
    if (mode == MODE_PACKED) { // wave-uniform
        packed_load<T, R::Cs>(p, r);
        return;
    }
    if constexpr (packed2_fits(R::Cs, sizeof(T))) {
        if (mode == MODE_PACKED2) { // elements two apart: fetch the covering span, keep every other one
            T span[2 * R::Cs - 1];
            packed_load<T, 2 * R::Cs - 1>(p, span);
#pragma unroll
            for (int c = 0; c < R::Cs; ++c) r[c] = span[2 * c];
            return;
        }
    }
    if constexpr (R::R > 1 && R::Cc > 1) {
        if (mode == MODE_PACKEDT) { // stored transposed
            T tr[R::Cs];
            packed_load<T, R::Cs>(p, tr);
#pragma unroll
            for (int a = 0; a < R::R; ++a)
#pragma unroll
                for (int b = 0; b < R::Cc; ++b) r[a * R::Cc + b] = tr[b * R::R + a];
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < R::R; ++a)
#pragma unroll
        for (int b = 0; b < R::Cc; ++b) r[a * R::Cc + b] = p[a * op.sr + b * op.sc];
}

template <typename T, class R>
__device__ __forceinline__ void rec_direct_store(const Opnd &op, int mode, int64_t o, int64_t i, bool valid,
                                                 const T (&r)[R::Cs])
{
    T *p = reinterpret_cast<T *>(op.ptr) + o * op.so + i * op.si;
    if (valid) {
        if (mode == MODE_PACKED) { // wave-uniform
            packed_store<T, R::Cs>(p, r);
            return;
        }
        if constexpr (R::R > 1 && R::Cc > 1) {
            if (mode == MODE_PACKEDT) {
                T tr[R::Cs];
#pragma unroll
                for (int a = 0; a < R::R; ++a)
#pragma unroll
                    for (int b = 0; b < R::Cc; ++b) tr[b * R::R + a] = r[a * R::Cc + b];
                packed_store<T, R::Cs>(p, tr);
                return;
            }
        }
#pragma unroll
        for (int a = 0; a < R::R; ++a)
#pragma unroll
            for (int b = 0; b < R::Cc; ++b) p[a * op.sr + b * op.sc] = r[a * R::Cc + b];
    }
}

// Ops that define `static constexpr bool kStream = true` provide apply_stream() (see step 4)
template <class Op, class = void>
struct op_streams : std::false_type {};
template <class Op>
struct op_streams<Op, std::void_t<decltype(Op::kStream)>> : std::bool_constant<Op::kStream> {};

// Ops that define `static constexpr bool kNoTile = true` never stage their records through LDS: their
// LDS image (a wavefront's worth of large records) would bound the occupancy of a kernel that is bound by
// its arithmetic, not by the load rate -- each lane fetches its record with packed 16-byte accesses.
template <class Op, class = void>
struct op_no_tile : std::false_type {};
template <class Op>
struct op_no_tile<Op, std::void_t<decltype(Op::kNoTile)>> : std::bool_constant<Op::kNoTile> {};

// Orders 9..16, one matrix per lane: which kernels fetch their records per lane instead of through the LDS
// transpose (op_no_tile, nfm_record_kernel.hpp).  Without the LDS image some of them gain a wavefront per
// SIMD (12x12 float32 solve: 257 -> 254 VGPRs, 1 -> 2 waves: 1.6x), others lose the coalescing for nothing;
// measured case by case on one box with two builds of the library (scripts/gpu_ab_large.sh,
// profiles/r02/large_no_tile_ab.md) -- the cases where the no-tile build won by more than 5 %.
enum { LN_SOLVE = 0, LN_DET = 1, LN_BDET = 2 };
__host__ __device__ constexpr bool large_no_tile(bool f64, int N, int what)
{
    if (N < 9) return false;
    if (!f64) {
        if (what == LN_SOLVE) return N >= 10 && N <= 14;
        if (what == LN_DET) return N <= 15;
        return N >= 12 && N <= 15; // LN_BDET
    }
    if (what == LN_SOLVE) return N <= 10;
    if (what == LN_DET) return N >= 10 && N <= 13;
    return N >= 10 && N <= 12; // LN_BDET
}

// mode of an operand in the compile-time (KIND_AOS) kernel: its preferred one, or for a no-tile Op the
// packed per-lane access wherever the preferred one is the LDS tile
template <class Op, class IO>
struct aos_mode {
    static constexpr int value = (op_no_tile<Op>::value && IO::pref == MODE_TILED) ? (int)MODE_PACKED : IO::pref;
};

// KIND_AOS = every operand is a contiguous batch-major block (the default torch layout):
// the movement mode of each operand is then a compile-time constant (packed access for
// 4/8/16-byte records, LDS transpose for the rest), which removes every mode branch and
// a third of the VGPRs (92 -> 60 on the 4x4 solve, 5 -> 8 waves per SIMD).
// KIND_SOA = the same for component-major (channel-first) fields: every multi-component
// operand through the SoA image.  KIND_ANY keeps the wave-uniform run-time modes for mixed,
// broadcast and strided operands.
// (A fourth compile-time kind -- every operand packed per lane at run-time strides, no LDS -- was
// built and measured in round 2: level with KIND_ANY on strided / padded records, ahead only for a
// broadcast 6x6 matrix; not kept.  profiles/r02/layouts_packed_kernel_ab.md)
enum { KIND_ANY = 0, KIND_AOS = 1, KIND_SOA = 2, KIND_SOAW = 3 }; // SOAW: SoA with 512-lane tiles

// lanes per workgroup of a kernel variant: Op::TILE, unless the Op names another size for its
// AoS variant (`kAosTile`: the 4x4 fp32 solve runs 512-lane tiles there, measured +2-3 % in
// same-box A/B runs; the other small-record Ops measured flat or slightly worse at 512).
template <class Op, class = void>
struct op_aos_tile {
    static constexpr int value = Op::TILE;
};
template <class Op>
struct op_aos_tile<Op, std::void_t<decltype(Op::kAosTile)>> {
    static constexpr int value = Op::kAosTile;
};
template <typename T, class Op, int KIND>
struct KindTile {
    static constexpr int value = KIND == KIND_AOS ? op_aos_tile<Op>::value : Op::TILE;
};
// Component runs that do not start on 16-byte boundaries (odd voxel counts) pay per tile for the
// two straddling vectors of every component: 512-lane tiles halve that (4x4 solve: +7 %), while
// aligned runs are better off at 256 (-4 % at 512).  Only Ops whose SoA image fits 80 KiB (6x6: +8 %).
template <typename T, class Op>
struct KindTile<T, Op, KIND_SOAW> {
#ifndef NFM_SOAW_LIMIT
#define NFM_SOAW_LIMIT (80 * 1024) // two 512-lane workgroups per CU
#endif
    static constexpr int value = (Op::TILE == 256 && RecLayout<T, Op, 512>::gtotal <= NFM_SOAW_LIMIT) ? 512 : Op::TILE;
};

// NFM_REC_KERNEL_ATTR: extra kernel attributes of a translation unit (the orders 9..16 of the QR family ask for
// the whole register file: `amdgpu_waves_per_eu(1, 1)`, without which the backend stops at 256 registers and spills)
#ifndef NFM_REC_KERNEL_ATTR
#define NFM_REC_KERNEL_ATTR
#endif
template <typename T, class Op, int KIND>
__global__ NFM_REC_KERNEL_ATTR __launch_bounds__((KindTile<T, Op, KIND>::value)) void rec_kernel(Opnd a, Opnd b, Opnd c, Opnd out,
                                                                          int64_t n_inner,
                                                                          typename Op::Params prm)
{
    constexpr bool FAST = KIND == KIND_AOS;
    constexpr bool SFAST = KIND == KIND_SOA || KIND == KIND_SOAW;
    constexpr int TILE = KindTile<T, Op, KIND>::value;
    using L = RecLayout<T, Op, TILE>;
    using RA = typename Op::RA;
    using RB = typename Op::RB;
    using RC = typename Op::RC;
    using RO = typename Op::RO;
    using IA = RecIO<T, RA, TILE>;
    using IB = RecIO<T, RB, TILE>;
    using IC = RecIO<T, RC, TILE>;
    using IO_ = RecIO<T, RO, TILE>;
    extern __shared__ __align__(16) unsigned char smem[];

    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int64_t i = tile0 + threadIdx.x;
    const int64_t o = blockIdx.y;
    const bool valid = i < n_inner;
    const int64_t left = n_inner - tile0; // batch elements from the tile start to the end

    T ra[RA::Cs], rb[RB::Cs], rc[RC::Cs], ro[RO::Cs];

    // LDS regions and movement modes: compile-time constants in the FAST kernel
    constexpr int oA = FAST ? L::offA : L::gA, oB = FAST ? L::offB : L::gB;
    constexpr int oC = FAST ? L::offC : L::gC, oO = FAST ? L::offO : L::gO;
    const int ma = FAST ? aos_mode<Op, IA>::value : (SFAST ? IA::pref_soa : a.tiled);
    const int mb = FAST ? aos_mode<Op, IB>::value : (SFAST ? IB::pref_soa : b.tiled);
    const int mc = FAST ? aos_mode<Op, IC>::value : (SFAST ? IC::pref_soa : c.tiled);
    const int mo = FAST ? aos_mode<Op, IO_>::value : (SFAST ? IO_::pref_soa : out.tiled);
    const bool use_c = RC::used && c.ptr != nullptr;
    const bool tA = IA::can_tile && ma == MODE_TILED, tB = IB::can_tile && mb == MODE_TILED;
    const bool tC = IC::can_tile && mc == MODE_TILED && use_c, tO = IO_::can_tile && mo == MODE_TILED;
    const bool vA = IA::can_vec && ma == MODE_VEC, vB = IB::can_vec && mb == MODE_VEC;
    const bool vC = IC::can_vec && mc == MODE_VEC, vO = IO_::can_vec && mo == MODE_VEC;
    const bool sA = !FAST && IA::can_soa && ma == MODE_SOA, sB = !FAST && IB::can_soa && mb == MODE_SOA;
    const bool sC = !FAST && IC::can_soa && mc == MODE_SOA && use_c, sO = !FAST && IO_::can_soa && mo == MODE_SOA;

    // the AoS-tile and SoA-tile staging registers of an operand are never live together
    union StageA { typename IA::IO::Stage t; typename IA::SO::Stage s; } uA;
    union StageB { typename IB::IO::Stage t; typename IB::SO::Stage s; } uB;
    union StageC { typename IC::IO::Stage t; typename IC::SO::Stage s; } uC;
    auto &stA = uA.t; auto &sqA = uA.s;
    auto &stB = uB.t; auto &sqB = uB.s;
    auto &stC = uC.t; auto &sqC = uC.s;

    // 1. every 16-byte global load of the tile in flight (no waits in between)
    if constexpr (IA::can_tile)
        if (tA) IA::IO::issue(reinterpret_cast<const T *>(a.ptr) + tile0 * RA::C, left * RA::C, stA);
    if constexpr (IB::can_tile)
        if (tB) IB::IO::issue(reinterpret_cast<const T *>(b.ptr) + tile0 * RB::C, left * RB::C, stB);
    if constexpr (IC::can_tile)
        if (tC) IC::IO::issue(reinterpret_cast<const T *>(c.ptr) + tile0 * RC::C, left * RC::C, stC);
    if constexpr (!FAST && IA::can_soa)
        if (sA) IA::SO::issue(reinterpret_cast<const T *>(a.ptr) + o * a.so + tile0, a.sr, a.sc, left, sqA);
    if constexpr (!FAST && IB::can_soa)
        if (sB) IB::SO::issue(reinterpret_cast<const T *>(b.ptr) + o * b.so + tile0, b.sr, b.sc, left, sqB);
    if constexpr (!FAST && IC::can_soa)
        if (sC) IC::SO::issue(reinterpret_cast<const T *>(c.ptr) + o * c.so + tile0, c.sr, c.sc, left, sqC);

    // 2. per-lane loads: one packed access for back-to-back 4/8/16-byte records,
    //    element-wise strided access for everything else
    if constexpr (RA::used) {
        if (vA) {
            if constexpr (IA::can_vec) rec_vec_load<T, RA>(a, o, i, valid, ra);
        } else if (!tA && !sA) rec_direct_load<T, RA>(a, ma, o, i, valid, ra);
    }
    if constexpr (RB::used) {
        if (vB) {
            if constexpr (IB::can_vec) rec_vec_load<T, RB>(b, o, i, valid, rb);
        } else if (!tB && !sB) rec_direct_load<T, RB>(b, mb, o, i, valid, rb);
    }
    if constexpr (RC::used) {
        if (vC) {
            if constexpr (IC::can_vec) rec_vec_load<T, RC>(c, o, i, valid && use_c, rc);
        } else if (!tC && !sC) rec_direct_load<T, RC>(c, mc, o, i, valid && use_c, rc);
    }

    // 3. LDS transpose: park the staged vectors, one barrier, every lane picks up its records
    if constexpr (IA::can_tile)
        if (tA) IA::IO::commit(smem + oA, stA);
    if constexpr (IB::can_tile)
        if (tB) IB::IO::commit(smem + oB, stB);
    if constexpr (IC::can_tile)
        if (tC) IC::IO::commit(smem + oC, stC);
    if constexpr (!FAST && IA::can_soa)
        if (sA) IA::SO::commit(smem + oA, sqA);
    if constexpr (!FAST && IB::can_soa)
        if (sB) IB::SO::commit(smem + oB, sqB);
    if constexpr (!FAST && IC::can_soa)
        if (sC) IC::SO::commit(smem + oC, sqC);
    if (tA || tB || tC || sA || sB || sC) __syncthreads();
    if constexpr (IA::can_tile)
        if (tA) IA::IO::read_own(smem + oA, ra);
    if constexpr (IB::can_tile)
        if (tB) IB::IO::read_own(smem + oB, rb);
    if constexpr (IC::can_tile)
        if (tC) IC::IO::read_own(smem + oC, rc);
    if constexpr (!FAST && IA::can_soa)
        if (sA) IA::SO::read_own(smem + oA, ra, reinterpret_cast<const T *>(a.ptr) + o * a.so + tile0, a.sr, a.sc);
    if constexpr (!FAST && IB::can_soa)
        if (sB) IB::SO::read_own(smem + oB, rb, reinterpret_cast<const T *>(b.ptr) + o * b.so + tile0, b.sr, b.sc);
    if constexpr (!FAST && IC::can_soa)
        if (sC) IC::SO::read_own(smem + oC, rc, reinterpret_cast<const T *>(c.ptr) + o * c.so + tile0, c.sr, c.sc);

    // 4. arithmetic.  Streaming Ops (FAST kernel, tiled output) write their output record
    //    element by element into this lane's slot of the LDS image instead of returning
    //    it in registers (an inverse built column by column never holds two matrices).
    if constexpr (op_streams<Op>::value && FAST && IO_::can_tile) {
        T *own = reinterpret_cast<T *>(smem + oO + threadIdx.x * IO_::IO::kRowStride);
        Op::apply_stream(ra, rb, rc, own, prm);
        __syncthreads();
        IO_::IO::flush(reinterpret_cast<T *>(out.ptr) + tile0 * RO::C, left * RO::C, smem + oO);
        return;
    } else {
        Op::apply(ra, rb, rc, ro, prm);
    }

    // 5. output
    if constexpr (IO_::can_tile) {
        if (tO) {
            IO_::IO::write_own(smem + oO, ro);
            __syncthreads();
            IO_::IO::flush(reinterpret_cast<T *>(out.ptr) + tile0 * RO::C, left * RO::C, smem + oO);
            return;
        }
    }
    if constexpr (!FAST && IO_::can_soa) {
        if (sO) {
            IO_::SO::write_own(smem + oO, ro, reinterpret_cast<const T *>(out.ptr) + o * out.so + tile0, out.sr, out.sc);
            __syncthreads();
            IO_::SO::flush(reinterpret_cast<T *>(out.ptr) + o * out.so + tile0, out.sr, out.sc, left, smem + oO);
            return;
        }
    }
    if constexpr (IO_::can_vec) {
        if (vO) {
            rec_vec_store<T, RO>(out, o, i, valid, ro);
            return;
        }
    }
    rec_direct_store<T, RO>(out, mo, o, i, valid, ro);
}

// Host launcher: picks the movement mode of every operand and the kernel variant.
// FAST_ONLY: instantiate only the compile-time-mode kernel and answer NFM_EFALLBACK when the
// operands do not qualify (used for orders 9..16, whose run-time-mode twin is not worth its
// compile time: the caller falls back to the LDS-resident kernels).
constexpr int NFM_EFALLBACK = -100;

template <typename T, class Op, bool FAST_ONLY = false>
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
    const bool c_absent = c->ptr == nullptr;
    struct Modes {
        int ma, mb, mc, mo;
        bool any, fast;
    };
    // movement mode of every operand for a batch of n records starting at the given pointers
    auto classify = [&](const nfm_operand *pa, const nfm_operand *pb, const nfm_operand *pc,
                        const nfm_operand *po, int64_t n) {
        constexpr bool tiles = !op_no_tile<Op>::value;
        const bool ta = tiles && L::A::can_tile && tile_ok(pa, RA::C, RA::R, RA::Cc, n_outer, n, sizeof(T));
        const bool tb = tiles && L::B::can_tile && tile_ok(pb, RB::C, RB::R, RB::Cc, n_outer, n, sizeof(T));
        const bool tc = tiles && L::C::can_tile && tile_ok(pc, RC::C, RC::R, RC::Cc, n_outer, n, sizeof(T));
        const bool to = tiles && L::O::can_tile && tile_ok(po, RO::C, RO::R, RO::Cc, n_outer, n, sizeof(T));
        auto mode = [&](bool tiled, bool can_vec, bool can_soa, const nfm_operand *op, int C, int R, int Cc,
                        bool input) {
            if (tiled) return (int)MODE_TILED;
            if (can_vec && vec_ok(op, C, R, Cc, n_outer, sizeof(T))) return (int)MODE_VEC;
            if (can_soa && soa_ok(op, C, R, sizeof(T))) return (int)MODE_SOA;
            if (packed_ok(op, C, R, Cc, sizeof(T))) return (int)MODE_PACKED;
            if (input && packed2_fits(C, sizeof(T)) && packed2_ok(op, C, R, Cc, sizeof(T))) return (int)MODE_PACKED2;
            if (packedt_ok(op, R, Cc, sizeof(T))) return (int)MODE_PACKEDT;
            return (int)MODE_STRIDED;
        };
        Modes m;
        m.ma = mode(ta, L::A::can_vec, L::A::can_soa, pa, RA::C, RA::R, RA::Cc, true);
        m.mb = mode(tb, L::B::can_vec, L::B::can_soa, pb, RB::C, RB::R, RB::Cc, true);
        m.mc = mode(tc, L::C::can_vec, L::C::can_soa, pc, RC::C, RC::R, RC::Cc, true);
        m.mo = mode(to, L::O::can_vec, L::O::can_soa, po, RO::C, RO::R, RO::Cc, false);
        m.any = ta || tb || tc || to || m.ma == MODE_SOA || m.mb == MODE_SOA || m.mc == MODE_SOA ||
                m.mo == MODE_SOA;
        // FAST path: every used operand in its preferred mode (an absent C operand is fine)
        // (for a no-tile Op a packed operand must also be what the tile would have taken: contiguous
        // back-to-back records -- the compile-time kernel computes addresses from the record size)
        auto want = [&](int pref, int nt_mode, const nfm_operand *op, int C, int R, int Cc, int got) {
            if (nt_mode == pref) return got == pref;
            return got == nt_mode && tile_ok(op, C, R, Cc, n_outer, n, sizeof(T));
        };
        m.fast = n_outer == 1 &&
                 (!RA::used || want(L::A::pref, aos_mode<Op, typename L::A>::value, pa, RA::C, RA::R, RA::Cc, m.ma)) &&
                 (!RB::used || want(L::B::pref, aos_mode<Op, typename L::B>::value, pb, RB::C, RB::R, RB::Cc, m.mb)) &&
                 (!RC::used || c_absent ||
                  want(L::C::pref, aos_mode<Op, typename L::C>::value, pc, RC::C, RC::R, RC::Cc, m.mc)) &&
                 want(L::O::pref, aos_mode<Op, typename L::O>::value, po, RO::C, RO::R, RO::Cc, m.mo) &&
                 L::A::pref != MODE_STRIDED && L::O::pref != MODE_STRIDED &&
                 (!RB::used || L::B::pref != MODE_STRIDED) && (!RC::used || L::C::pref != MODE_STRIDED);
        return m;
    };
    Modes md = classify(a, b, c, out, n_inner);
    const int ma = md.ma, mb = md.mb, mc = md.mc, mo = md.mo;
    const bool any = md.any, fast = md.fast;
    constexpr int TILE_F = KindTile<T, Op, KIND_AOS>::value; // lanes per workgroup of the AoS variant
    using LF = RecLayout<T, Op, TILE_F>;
    const int tile = fast ? TILE_F : Op::TILE;
    const int64_t nblk = (n_inner + tile - 1) / tile;
    if (nblk > 0x7fffffffLL) return NFM_ESIZE;
    dim3 grid((unsigned)nblk, (unsigned)n_outer, 1), block(tile, 1, 1);
    const size_t lds = any ? (size_t)L::gtotal : 0;
    // > 64 KiB of dynamic LDS: opt-in per kernel and per device (lds_opt_in, nfm_common.hpp)
    if (fast) {
        if (!op_no_tile<Op>::value && LF::total > 64 * 1024) {
            static std::atomic<uint64_t> have_aos{0};
            const int rc = lds_opt_in(have_aos, reinterpret_cast<const void *>(&rec_kernel<T, Op, KIND_AOS>), LF::total);
            if (rc != NFM_OK) return rc;
        }
    } else if (L::gtotal > 64 * 1024) {
        if constexpr (!FAST_ONLY) {
            static std::atomic<uint64_t> have_any{0}, have_soa{0};
            int rc = lds_opt_in(have_any, reinterpret_cast<const void *>(&rec_kernel<T, Op, KIND_ANY>), L::gtotal);
            if (rc == NFM_OK)
                rc = lds_opt_in(have_soa, reinterpret_cast<const void *>(&rec_kernel<T, Op, KIND_SOA>), L::gtotal);
            if (rc != NFM_OK) return rc;
        }
    }
    if (fast) {
        hipLaunchKernelGGL((rec_kernel<T, Op, KIND_AOS>), grid, block, op_no_tile<Op>::value ? (size_t)0 : (size_t)LF::total,
                           static_cast<hipStream_t>(stream), make_opnd(a, ma), make_opnd(b, mb), make_opnd(c, mc),
                           make_opnd(out, mo), n_inner, prm);
    } else {
        if constexpr (FAST_ONLY) return NFM_EFALLBACK;
        else {
            // channel-first fields: every used operand in its SoA-preferred mode
            const bool sfast = (!RA::used || ma == L::A::pref_soa) && (!RB::used || mb == L::B::pref_soa) &&
                               (!RC::used || c_absent || mc == L::C::pref_soa) && mo == L::O::pref_soa &&
                               L::A::pref_soa != MODE_STRIDED && L::O::pref_soa != MODE_STRIDED &&
                               (!RB::used || L::B::pref_soa != MODE_STRIDED) &&
                               (!RC::used || L::C::pref_soa != MODE_STRIDED);
            constexpr int TILE_W = KindTile<T, Op, KIND_SOAW>::value;
            bool wide = false;
            if constexpr (TILE_W != Op::TILE) {
                // does any component run of any SoA operand start off a 16-byte boundary?
                auto off16 = [&](const nfm_operand *op, int mode, int rows) {
                    if (mode != MODE_SOA || op->ptr == nullptr) return false;
                    const int64_t v = 16 / (int64_t)sizeof(T);
                    return reinterpret_cast<uintptr_t>(op->ptr) % 16 != 0 || op->stride_col % v != 0 ||
                           op->stride_outer % v != 0 || (rows > 1 && op->stride_row % v != 0);
                };
                wide = sfast && (off16(a, ma, RA::R) || off16(b, mb, RB::R) || (!c_absent && off16(c, mc, RC::R)) ||
                                 off16(out, mo, RO::R));
            }
            if (sfast && wide) {
                if constexpr (TILE_W != Op::TILE) {
                    using LW = RecLayout<T, Op, TILE_W>;
                    if (LW::gtotal > 64 * 1024) {
                        static std::atomic<uint64_t> have_soaw{0};
                        const int rc = lds_opt_in(
                            have_soaw, reinterpret_cast<const void *>(&rec_kernel<T, Op, KIND_SOAW>), LW::gtotal);
                        if (rc != NFM_OK) return rc;
                    }
                    const int64_t nb = (n_inner + TILE_W - 1) / TILE_W;
                    hipLaunchKernelGGL((rec_kernel<T, Op, KIND_SOAW>), dim3((unsigned)nb, (unsigned)n_outer, 1),
                                       dim3(TILE_W, 1, 1), (size_t)LW::gtotal, static_cast<hipStream_t>(stream),
                                       make_opnd(a, ma), make_opnd(b, mb), make_opnd(c, mc), make_opnd(out, mo),
                                       n_inner, prm);
                }
            } else if (sfast)
                hipLaunchKernelGGL((rec_kernel<T, Op, KIND_SOA>), grid, block, lds,
                                   static_cast<hipStream_t>(stream), make_opnd(a, ma), make_opnd(b, mb),
                                   make_opnd(c, mc), make_opnd(out, mo), n_inner, prm);
            else
                hipLaunchKernelGGL((rec_kernel<T, Op, KIND_ANY>), grid, block, lds,
                                   static_cast<hipStream_t>(stream), make_opnd(a, ma), make_opnd(b, mb),
                                   make_opnd(c, mc), make_opnd(out, mo), n_inner, prm);
        }
    }
    return launch_status();
}

} // namespace nfm

// nfm_common.hpp -- shared device/host machinery for the gfx950 kernels.
//
// Execution model used by every small-matrix kernel in this library:
//   * one matrix (one batch element) per LANE, held entirely in VGPRs;
//   * a workgroup of TILE lanes owns TILE consecutive batch elements;
//   * operands that are contiguous batch-major ("AoS": the default torch layout,
//     e.g. mat (n, K)) are moved HBM <-> registers through an LDS transpose:
//     the workgroup streams its TILE*C contiguous elements with 16-byte-per-lane
//     loads (1 KiB per wave instruction, the coalescing sweet spot), parks them in
//     LDS, and each lane then reads its own C-element record with conflict-free
//     wide LDS reads.  Stores run the same path backwards;
//   * records of exactly 4/8/16 bytes need no transpose (one packed access per lane);
//   * component-major ("SoA", channel-first) operands stream each component's run with
//     16-byte loads into an LDS image [component][TILE] (SoaIO);
//   * records contiguous inside at any batch stride (strided, padded, broadcast): 16-byte accesses
//     per lane (MODE_PACKED); anything else is read/written element by
//     element by each lane (rec_direct_load/store in nfm_record_kernel.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "../../include/nfm_hip.h"

namespace nfm {

constexpr int kWave = 64;

// streaming global accesses: every byte is touched once, so loads and stores carry the
// nontemporal hint (measured: see DESIGN.md section 4); the macros exist for that experiment
#if defined(NFM_PLAIN_LOADS)
#define NFM_LDG(p) (*(p))
#else
#define NFM_LDG(p) __builtin_nontemporal_load(p)
#endif
#if defined(NFM_PLAIN_STORES)
#define NFM_STG(v, p) (*(p) = (v))
#else
#define NFM_STG(v, p) __builtin_nontemporal_store(v, p)
#endif

// Device-side view of nfm_operand (+ whether it takes the LDS-transposed path).
struct Opnd {
    char *ptr;
    int64_t so, si, sr, sc;
    int tiled;
};

template <typename T>
struct VecOf;
// `gtype` is the same 16-byte vector with ELEMENT alignment: what global memory is accessed
// through.  gfx950 executes global_load/store_dwordx4 at any dword-aligned address (the compiler
// emits the same instruction for both types), so a contiguous operand whose base sits at an odd
// element offset inside a 16-byte line -- x[1:], a field carved out of a larger buffer -- streams
// through the same wide accesses as an aligned one; only LDS keeps the 16-byte alignment.
template <>
struct VecOf<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    typedef float gtype __attribute__((ext_vector_type(4), aligned(4)));
    static constexpr int N = 4;
};
template <>
struct VecOf<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    typedef double gtype __attribute__((ext_vector_type(2), aligned(8)));
    static constexpr int N = 2;
};

__host__ __device__ constexpr int sym_k(int M) { return M * (M + 1) / 2; }
// index of (i, j), i < j, in compact storage: diagonal first, then upper rows
__host__ __device__ constexpr int sym_idx(int M, int i, int j)
{
    return i == j ? i : (i < j ? M + i * (M - 1) - i * (i - 1) / 2 + (j - i - 1)
                               : M + j * (M - 1) - j * (j - 1) / 2 + (i - j - 1));
}

// ---------------------------------------------------------------------------
// TileIO<T, C, TILE>: LDS-transposed movement of TILE records of C elements.
//
// LDS image: record r starts at byte r * kRowStride.  Bank-conflict rules
// (MI355X_MICROARCH.md, LDS): a lane reads its record with the widest access W
// dividing the record size RB; consecutive lanes are RB (or kRowStride) apart.
//   RB % 16 == 0 : ds_read_b128, conflict-free iff the stride in 16-B slots is odd
//                  -> pad one slot when RB/16 is even;
//   RB % 8  == 0 : ds_read_b64, stride in 8-B units is odd -> conflict-free;
//   else         : ds_read_b32, stride in dwords is odd -> conflict-free.
// ---------------------------------------------------------------------------
template <typename T, int C, int TILE>
struct TileIO {
    using V = typename VecOf<T>::type;
    using VG = typename VecOf<T>::gtype; // element-aligned: global side only
    static constexpr int kVec = VecOf<T>::N;
    static constexpr int RB = C * (int)sizeof(T);
    static constexpr bool kWide = (RB % 16) == 0;
    static constexpr int kSlots = RB / 16;
    static constexpr int kPadSlots = (kWide && (kSlots % 2 == 0)) ? kSlots + 1 : kSlots;
    static constexpr int kRowStride = kWide ? kPadSlots * 16 : RB;
    static constexpr int kLdsBytes = ((TILE * kRowStride + 15) / 16) * 16;
    static constexpr int kTileElems = TILE * C;
    static constexpr int kNVec = (TILE * RB) / 16; // TILE % 4 == 0 -> exact
    static constexpr int kIters = (kNVec + TILE - 1) / TILE;
    static_assert(TILE % 4 == 0, "tile must keep 16-byte alignment");

    struct Stage {
        V v[kIters];
    };

    // byte offset in the LDS image of global 16-byte vector q of the tile
    static __device__ __forceinline__ int lds_off(int q)
    {
        if constexpr (kWide && kPadSlots != kSlots) {
            const int row = q / kSlots, col = q - row * kSlots;
            return (row * kPadSlots + col) * 16;
        } else {
            return q * 16;
        }
    }

    // Phase 1: issue the global loads of this workgroup's tile (no waits).
    // g = address of the tile's first element; avail = elements readable from g.
    static __device__ __forceinline__ void issue(const T *__restrict__ g, int64_t avail, Stage &st)
    {
        const int tid = threadIdx.x;
        if (avail >= kTileElems) {
#pragma unroll
            for (int it = 0; it < kIters; ++it) {
                const int q = tid + it * TILE;
                if (kNVec % TILE == 0 || q < kNVec)
                    st.v[it] = NFM_LDG(reinterpret_cast<const VG *>(g) + q);
            }
        } else {
#pragma unroll
            for (int it = 0; it < kIters; ++it) {
                const int q = tid + it * TILE;
                V v;
#pragma unroll
                for (int k = 0; k < kVec; ++k) {
                    const int64_t e = (int64_t)q * kVec + k;
                    v[k] = (e < avail) ? g[e] : T(0);
                }
                st.v[it] = v;
            }
        }
    }

    // Phase 2: park the staged vectors in LDS (caller syncs afterwards).
    static __device__ __forceinline__ void commit(unsigned char *lds, const Stage &st)
    {
        const int tid = threadIdx.x;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int q = tid + it * TILE;
            if (kNVec % TILE == 0 || q < kNVec) *reinterpret_cast<V *>(lds + lds_off(q)) = st.v[it];
        }
    }

    // Phase 3: each lane reads its own record.
    static __device__ __forceinline__ void read_own(const unsigned char *lds, T (&r)[C])
    {
        const unsigned char *p = lds + threadIdx.x * kRowStride;
        if constexpr (kWide) {
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                V v = *reinterpret_cast<const V *>(p + s * 16);
#pragma unroll
                for (int k = 0; k < kVec; ++k) r[s * kVec + k] = v[k];
            }
        } else if constexpr (RB % 8 == 0 && sizeof(T) == 4) {
            typedef float F2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int s = 0; s < C / 2; ++s) {
                F2 v = *reinterpret_cast<const F2 *>(p + s * 8);
                r[2 * s] = v[0];
                r[2 * s + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) r[c] = *reinterpret_cast<const T *>(p + c * sizeof(T));
        }
    }

    static __device__ __forceinline__ void write_own(unsigned char *lds, const T (&r)[C])
    {
        unsigned char *p = lds + threadIdx.x * kRowStride;
        if constexpr (kWide) {
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                V v;
#pragma unroll
                for (int k = 0; k < kVec; ++k) v[k] = r[s * kVec + k];
                *reinterpret_cast<V *>(p + s * 16) = v;
            }
        } else if constexpr (RB % 8 == 0 && sizeof(T) == 4) {
            typedef float F2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int s = 0; s < C / 2; ++s) {
                F2 v;
                v[0] = r[2 * s];
                v[1] = r[2 * s + 1];
                *reinterpret_cast<F2 *>(p + s * 8) = v;
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) *reinterpret_cast<T *>(p + c * sizeof(T)) = r[c];
        }
    }

    // Cooperative store of the LDS image to global (caller synced after write_own).
    static __device__ __forceinline__ void flush(T *__restrict__ g, int64_t avail, const unsigned char *lds)
    {
        const int tid = threadIdx.x;
        if (avail >= kTileElems) {
#pragma unroll
            for (int it = 0; it < kIters; ++it) {
                const int q = tid + it * TILE;
                if (kNVec % TILE == 0 || q < kNVec) {
                    const V v = *reinterpret_cast<const V *>(lds + lds_off(q));
                    NFM_STG(static_cast<VG>(v), reinterpret_cast<VG *>(g) + q);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < kIters; ++it) {
                const int q = tid + it * TILE;
                if (kNVec % TILE == 0 || q < kNVec) {
                    V v = *reinterpret_cast<const V *>(lds + lds_off(q));
#pragma unroll
                    for (int k = 0; k < kVec; ++k) {
                        const int64_t e = (int64_t)q * kVec + k;
                        if (e < avail) g[e] = v[k];
                    }
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------
// SoaIO<T, R, Cc, TILE>: component-major ("SoA", channel-first) operands -- element i of
// component (r, c) at ptr[i + r*sr + c*sc].  Consecutive lanes already touch consecutive
// addresses, but one element per lane is only a 4-byte access; instead the workgroup
// streams every component's TILE-element run with ALIGNED 16-byte vectors (1 KiB per wave
// instruction) into an LDS image laid out [component][TILE + kVec], from which lane t
// reads element t of every component (stride-1 across lanes: conflict-free).
// A component run need not start on a 16-byte boundary (volumes with an odd number of
// voxels never do): the vectors are taken from the aligned address below the run, so the
// image of component c is shifted by a_c = (address of its first element / sizeof(T)) mod
// kVec, and only the (at most two) vectors that straddle the ends of the run are handled
// element by element -- neighbouring tiles own the other halves of those vectors.
// ---------------------------------------------------------------------------
template <typename T, int R, int Cc, int TILE>
struct SoaIO {
    using V = typename VecOf<T>::type;
    static constexpr int kVec = VecOf<T>::N;
    static constexpr int C = R * Cc;
    static constexpr int kVecPerComp = TILE / kVec; // + one "extra" vector per shifted component
    static constexpr int kPitch = TILE + kVec;      // elements per component row in LDS
    static constexpr int kNVec = C * kVecPerComp;
    static constexpr int kIters = (kNVec + TILE - 1) / TILE;
    static constexpr int kLdsBytes = C * kPitch * (int)sizeof(T);
    static_assert(TILE % kVec == 0, "tile must be a whole number of vectors");
    static constexpr int kXIters = (C + TILE - 1) / TILE; // extra vectors per lane (1 unless C > TILE)

    struct Stage {
        V v[kIters];
        V x[kXIters]; // the extra vector of component `threadIdx.x` (+ k * TILE): shifted runs only
    };

    static __device__ __forceinline__ int64_t comp_off(int comp, int64_t sr, int64_t sc)
    {
        const int r = comp / Cc, c = comp - r * Cc;
        return r * sr + c * sc;
    }
    // shift of a component's image: its first element sits `mis` elements above a 16-byte boundary
    static __device__ __forceinline__ int mis(const T *g, int64_t off)
    {
        return (int)(((int64_t)(reinterpret_cast<uintptr_t>(g) / sizeof(T)) + off) & (kVec - 1));
    }

    // g = address of element tile0 of component (0, 0); left = elements from tile0 to the end.
    // A vector that holds at least one element of the run is loaded whole: an aligned 16-byte
    // access cannot leave the page of that element, and the lanes outside the run only land
    // in image slots that no record reads.
    static __device__ __forceinline__ void issue(const T *__restrict__ g, int64_t sr, int64_t sc, int64_t left,
                                                 Stage &st)
    {
        const int tid = threadIdx.x;
        const int n = left < TILE ? (int)left : TILE; // elements of this tile
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int q = tid + it * TILE;
            if (kNVec % TILE == 0 || q < kNVec) {
                const int comp = q / kVecPerComp, jv = q - comp * kVecPerComp;
                const int64_t off = comp_off(comp, sr, sc);
                const int e0 = jv * kVec - mis(g, off); // tile element held by slot 0 of the vector
                if (e0 + kVec > 0 && e0 < n)
                    st.v[it] = NFM_LDG(reinterpret_cast<const V *>(g + off + e0));
            }
        }
#pragma unroll
        for (int k = 0; k < kXIters; ++k) {
            const int comp = tid + k * TILE;
            if (comp < C) {
                const int64_t off = comp_off(comp, sr, sc);
                const int a = mis(g, off);
                if (a != 0 && TILE - a < n)
                    st.x[k] = NFM_LDG(reinterpret_cast<const V *>(g + off + (TILE - a)));
            }
        }
    }

    static __device__ __forceinline__ int slot(int q) // LDS element index of main vector q
    {
        const int comp = q / kVecPerComp, jv = q - comp * kVecPerComp;
        return comp * kPitch + jv * kVec;
    }

    static __device__ __forceinline__ void commit(unsigned char *lds, const Stage &st)
    {
        const int tid = threadIdx.x;
        T *l = reinterpret_cast<T *>(lds);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int q = tid + it * TILE;
            if (kNVec % TILE == 0 || q < kNVec) *reinterpret_cast<V *>(l + slot(q)) = st.v[it];
        }
#pragma unroll
        for (int k = 0; k < kXIters; ++k) {
            const int comp = tid + k * TILE;
            if (comp < C) *reinterpret_cast<V *>(l + comp * kPitch + TILE) = st.x[k];
        }
    }

    static __device__ __forceinline__ void read_own(const unsigned char *lds, T (&r)[C], const T *g, int64_t sr,
                                                    int64_t sc)
    {
        const T *p = reinterpret_cast<const T *>(lds) + threadIdx.x;
#pragma unroll
        for (int c = 0; c < C; ++c) r[c] = p[c * kPitch + mis(g, comp_off(c, sr, sc))];
    }

    static __device__ __forceinline__ void write_own(unsigned char *lds, const T (&r)[C], const T *g, int64_t sr,
                                                     int64_t sc)
    {
        T *p = reinterpret_cast<T *>(lds) + threadIdx.x;
#pragma unroll
        for (int c = 0; c < C; ++c) p[c * kPitch + mis(g, comp_off(c, sr, sc))] = r[c];
    }

    // whole vectors inside the run are stored whole; the two that straddle its ends are stored
    // element by element (the other halves belong to the neighbouring tiles)
    static __device__ __forceinline__ void put(T *p, const V v, int e0, int n)
    {
        if (e0 >= 0 && e0 + kVec <= n) {
            NFM_STG(v, reinterpret_cast<V *>(p));
        } else {
#pragma unroll
            for (int k = 0; k < kVec; ++k)
                if (e0 + k >= 0 && e0 + k < n) p[k] = v[k];
        }
    }

    static __device__ __forceinline__ void flush(T *__restrict__ g, int64_t sr, int64_t sc, int64_t left,
                                                 const unsigned char *lds)
    {
        const int tid = threadIdx.x;
        const int n = left < TILE ? (int)left : TILE;
        const T *l = reinterpret_cast<const T *>(lds);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int q = tid + it * TILE;
            if (kNVec % TILE == 0 || q < kNVec) {
                const int comp = q / kVecPerComp, jv = q - comp * kVecPerComp;
                const int64_t off = comp_off(comp, sr, sc);
                const int e0 = jv * kVec - mis(g, off);
                if (e0 + kVec > 0 && e0 < n) put(g + off + e0, *reinterpret_cast<const V *>(l + slot(q)), e0, n);
            }
        }
#pragma unroll
        for (int k = 0; k < kXIters; ++k) {
            const int comp = tid + k * TILE;
            if (comp < C) {
                const int64_t off = comp_off(comp, sr, sc);
                const int a = mis(g, off);
                if (a != 0 && TILE - a < n)
                    put(g + off + (TILE - a), *reinterpret_cast<const V *>(l + comp * kPitch + TILE), TILE - a, n);
            }
        }
    }
};

// ------------------------------------------------------------------ host side
// Can the operand take the LDS-transposed path?  It must be a single contiguous
// batch-major block: records of C elements back to back, 16-byte aligned base.
inline bool tile_ok(const nfm_operand *op, int C, int rows, int cols, int64_t n_outer, int64_t n_inner,
                    size_t elem)
{
    if (op->ptr == nullptr) return false;
    // any element-aligned base: the tile's 16-byte global accesses only need dword alignment
    // (VecOf::gtype; measured: a base 4 bytes off a 16-byte line streams at the aligned rate,
    // 5.85 vs 5.63 TB/s for the 4x4 solve, profiles/r02/layouts_table.md)
    if (reinterpret_cast<uintptr_t>(op->ptr) % elem != 0) return false;
    if (n_outer != 1) return false; // the facade collapses contiguous outer levels into one
    if (op->stride_inner != C) return false;
    if (rows > 1) {
        if (op->stride_row != cols || op->stride_col != 1) return false;
    } else if (C > 1 && op->stride_col != 1) {
        return false;
    }
    return true;
}

// Can the operand be moved with one packed 4/8/16-byte access per lane?  Records back
// to back along the inner batch level, outer slabs a whole number of records apart.
inline bool vec_ok(const nfm_operand *op, int C, int rows, int cols, int64_t n_outer, size_t elem)
{
    if (op->ptr == nullptr) return false;
    if (reinterpret_cast<uintptr_t>(op->ptr) % elem != 0) return false; // PackOf is element-aligned
    if (op->stride_inner != C) return false;
    if (n_outer > 1 && (op->stride_outer % C) != 0) return false;
    if (rows > 1) {
        if (op->stride_row != cols || op->stride_col != 1) return false;
    } else if (C > 1 && op->stride_col != 1) {
        return false;
    }
    return true;
}

// Can the operand take the component-major tile path?  Unit stride along the inner batch
// level is all it takes (component runs may start at any element: SoaIO shifts their image).
inline bool soa_ok(const nfm_operand *op, int C, int rows, size_t elem)
{
    if (op->ptr == nullptr || C < 2) return false;
    if (reinterpret_cast<uintptr_t>(op->ptr) % elem != 0) return false;
    return op->stride_inner == 1;
}

// Is the record contiguous inside (whatever the batch strides, 0 included)?  Then a lane moves it
// with whole 16-byte accesses (MODE_PACKED).  Records of fewer than two elements gain nothing.
inline bool packed_ok(const nfm_operand *op, int C, int rows, int cols, size_t elem)
{
    if (op->ptr == nullptr || C < 2) return false;
    if (reinterpret_cast<uintptr_t>(op->ptr) % elem != 0) return false;
    if (rows > 1) return op->stride_row == cols && op->stride_col == 1;
    return op->stride_col == 1;
}

// the covering span of a step-2 record is live in registers all at once: up to 72 of them
__host__ __device__ constexpr bool packed2_fits(int C, size_t elem) { return C > 1 && (2 * C - 1) * elem <= 288; }

// elements of the record two apart (and rows of a matrix record 2 * cols apart): the covering span
// of 2C - 1 elements is fetched packed (inputs only: the gaps are not ours to write)
inline bool packed2_ok(const nfm_operand *op, int C, int rows, int cols, size_t elem)
{
    if (op->ptr == nullptr || C < 2) return false;
    if (reinterpret_cast<uintptr_t>(op->ptr) % elem != 0) return false;
    if (rows > 1) return op->stride_row == 2 * cols && op->stride_col == 2;
    return op->stride_col == 2;
}

// a full matrix stored transposed (column-major inside the record)
inline bool packedt_ok(const nfm_operand *op, int rows, int cols, size_t elem)
{
    if (op->ptr == nullptr || rows < 2 || cols < 2) return false;
    if (reinterpret_cast<uintptr_t>(op->ptr) % elem != 0) return false;
    return op->stride_row == 1 && op->stride_col == rows;
}

inline Opnd make_opnd(const nfm_operand *op, int tiled)
{
    Opnd d;
    d.ptr = static_cast<char *>(op->ptr);
    d.so = op->stride_outer;
    d.si = op->stride_inner;
    d.sr = op->stride_row;
    d.sc = op->stride_col;
    d.tiled = tiled;
    return d;
}

inline int check_common(int dtype, int64_t n_outer, int64_t n_inner)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (n_outer < 0 || n_inner < 0) return NFM_EINVAL;
    if (n_outer > 65535) return NFM_ESIZE;
    return NFM_OK;
}

inline int check_operand(const nfm_operand *op, int dtype, bool nonempty)
{
    if (op == nullptr) return NFM_EINVAL;
    if (nonempty && op->ptr == nullptr) return NFM_EINVAL;
    const size_t e = dtype == NFM_F32 ? 4 : 8;
    if (reinterpret_cast<uintptr_t>(op->ptr) % e != 0) return NFM_EALIGN;
    return NFM_OK;
}

inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NFM_OK : (int)e;
}

// More than 64 KiB of dynamic LDS needs an opt-in (hipFuncSetAttribute) per kernel function AND
// per device: the attribute lives in the device's copy of the code object.  `mask` is the
// kernel's own record of the devices that already have it (bit d = device d; devices >= 64 are
// never recorded and simply re-apply it).  Keyed on the CURRENT device of the calling thread,
// lock-free, and idempotent when two host threads race (both set the same value).  The HIP
// status is propagated: a failed opt-in must not be followed by a launch that needs it.
inline int lds_opt_in(std::atomic<uint64_t> &mask, const void *kernel, size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && ((mask.load(std::memory_order_acquire) >> dev) & 1u)) return NFM_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (tracked) mask.fetch_or(uint64_t(1) << dev, std::memory_order_release);
    return NFM_OK;
}

// pick the number of lanes per workgroup so that the LDS image stays <= ~32 KiB
// (>= 4-5 workgroups per CU) and never below one wave
__host__ __device__ constexpr int pick_tile(int bytes_per_lane)
{
    return bytes_per_lane * 256 <= 36 * 1024 ? 256 : (bytes_per_lane * 128 <= 36 * 1024 ? 128 : 64);
}

} // namespace nfm

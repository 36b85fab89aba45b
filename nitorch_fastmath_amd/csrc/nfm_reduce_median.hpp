// nfm_reduce_median.hpp -- pieces shared by nfm_reduce_median.hip (radix selection) and
// nfm_reduce_median_lane.hip (one row per lane): the order-preserving integer keys.
#pragma once
#include "nfm_reduce_common.hpp"

namespace nfm {
namespace med {

template <typename T>
struct Key;
template <>
struct Key<float> {
    using U = uint32_t;
    static constexpr int digits = 4;
    static __device__ __forceinline__ U of(float x)
    {
        const U u = (U)__float_as_int(x);
        if (x != x) return ~U(0);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    static __device__ __forceinline__ float back(U k)
    {
        const U u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return __int_as_float((int)u);
    }
    // the lane kernels' hot path: the key WITHOUT the NaN rule (three instructions), NaNs found afterwards from
    // the key itself -- a NaN of either sign lands outside the keys of -inf .. +inf
    static __device__ __forceinline__ U bits(float x) { return (U)__float_as_int(x); }
    static __device__ __forceinline__ U raw(float x)
    {
        const int u = __float_as_int(x);
        return (U)u ^ ((U)(u >> 31) | 0x80000000u);
    }
    static __device__ __forceinline__ bool raw_is_nan(U k) { return k > 0xff800000u || k < 0x007fffffu; }
};
template <>
struct Key<double> {
    using U = uint64_t;
    static constexpr int digits = 8;
    static __device__ __forceinline__ U of(double x)
    {
        const U u = (U)__double_as_longlong(x);
        if (x != x) return ~U(0);
        return (u >> 63) ? ~u : (u | (U(1) << 63));
    }
    static __device__ __forceinline__ double back(U k)
    {
        const U u = (k >> 63) ? (k & ~(U(1) << 63)) : ~k;
        return __longlong_as_double((long long)u);
    }
    static __device__ __forceinline__ U bits(double x) { return (U)__double_as_longlong(x); }
    static __device__ __forceinline__ U raw(double x)
    {
        const long long u = __double_as_longlong(x);
        return (U)u ^ ((U)(u >> 63) | (U(1) << 63));
    }
    static __device__ __forceinline__ bool raw_is_nan(U k) { return k > 0xfff0000000000000ull || k < 0x000fffffffffffffull; }
};

// longest row sorted by one lane (nfm_reduce_median_lane.hip): the keys live in registers (128 / 2 x 64
// VGPRs) and the workgroup's LDS image stays at 32 KiB with 64-lane tiles
template <typename T>
struct LaneMax {
    static constexpr int value = sizeof(T) == 4 ? 128 : 64;
};
// rows up to 1.5 x that length are sorted by one lane too, padded to one of 4 bucket lengths per dtype
// (median_lane_pad_kernel): bucket b lives in part first_part + b and is reached with red passed NEGATED
// float32: 129..192 (4 buckets of 16, parts 0..3); float64: 65..96 (4 buckets of 8, parts 4..7).  Rows up to
// 2 x LaneMax (193..256 / 97..128) take four lanes per row (median_lane_quad_kernel, part 5 / part 1): a 256-key
// network in one lane is 7 700 instructions on more registers than a lane has architectural ones -- a quarter of
// an hour of compile time per kernel.
template <typename T>
struct LanePadBuckets {
    static constexpr int value = 4;
    static constexpr int first_part = sizeof(T) == 4 ? 0 : 4;
};
template <typename T>
struct LanePadMax {
    static constexpr int value = LaneMax<T>::value + LanePadBuckets<T>::value * (LaneMax<T>::value / 8);
};
constexpr int kLaneParts = 8; // the lane kernels are compiled in 8 objects: row lengths by their residue mod 8

// part `p` serves the row lengths red with red % 8 == p; returns NFM_EINVAL for a length it does not hold.
// rows = number of rows; inner = 1 for contiguous rows, else the (outer, red, inner) layout (rows = outer * inner)
#define NFM_MED_LANE_DECL(p) \
    int lane_part##p(int dtype, int red, int omitnan, int64_t rows, int64_t inner, const void *x, void *val, void *idx, \
                     void *stream);
NFM_MED_LANE_DECL(0) NFM_MED_LANE_DECL(1) NFM_MED_LANE_DECL(2) NFM_MED_LANE_DECL(3)
NFM_MED_LANE_DECL(4) NFM_MED_LANE_DECL(5) NFM_MED_LANE_DECL(6) NFM_MED_LANE_DECL(7)
#undef NFM_MED_LANE_DECL

} // namespace med
} // namespace nfm

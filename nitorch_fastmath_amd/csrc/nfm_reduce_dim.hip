// nfm_reduce_dim.hip -- NaN-aware reductions over the middle axis of a contiguous
// (outer, red, inner) view: sum/max/min families (+ first-occurrence indices), one-pass
// moments and the mean/var/std built on them (reference `reduce.py:49-142`, `:513-763`).
//
// Everything is HBM-bound (4-8 B per element, a handful of VALU ops), so the only design
// question is how lanes map to memory.  Three streaming kernels, all reading with 16-byte
// nontemporal loads when the view allows it (VEC = 4 floats / 2 doubles, else scalar):
//   GROUP  small slabs (red x inner elements <= 4 KiB; inner 1, 2 or VEC times a power of two):
//          2^g <= 8 lanes share a slab (>= 128 contiguous bytes per group and load), several
//          slabs or several loads in flight per lane, xor-shuffle merge inside the group;
//   FLAT   small inner (inner / gcd(inner, VEC) <= 64): the (red x inner) slab of one `outer`
//          index is a contiguous array; one wavefront streams a chunk of it with A <= 64 lanes,
//          A * VEC a multiple of inner, so every lane component always meets the SAME output
//          column; columns are merged through LDS (or shuffles when inner == 1);
//   COL    large inner: lanes lie along inner (VEC adjacent columns per lane) and walk the
//          reduced axis with four loads in flight.
// Few outputs + a long reduced axis: FLAT/COL cut the axis into chunks (one partial record
// per chunk and output in the caller's workspace) and a second kernel folds the chunks in a
// fixed order, so results do not depend on timing.  The plan is a pure function of the shape.
#include "nfm_reduce_common.hpp"

namespace nfm {

struct FinParams {
    void *out;
    int out_dtype;
    int64_t *idx;
    int stat; // < 0: raw moments [n, s, q, K]; else NFM_STAT_* (+ flags)
    int64_t red;
};

__device__ __forceinline__ void put_value(const FinParams &f, int64_t e, double v)
{
    if (f.out_dtype == NFM_F32) static_cast<float *>(f.out)[e] = (float)v;
    else static_cast<double *>(f.out)[e] = v;
}

// ---- accumulators: init / fold(value, position) / merge / shfl / save+load (partials) / finish
template <typename T, int OP>
struct ValAcc {
    static constexpr int NP = 1;
    double a;
    __device__ __forceinline__ void init_empty() { a = RedOp<OP>::identity(); }
    __device__ __forceinline__ void init(const T *, int64_t, int64_t) { init_empty(); }
    template <int VEC>
    __device__ __forceinline__ void init_vals(const T (&)[VEC], bool) { init_empty(); }
    __device__ __forceinline__ void init_own(T, bool) { init_empty(); }
    __device__ __forceinline__ void fold(T v, int64_t) { RedOp<OP>::fold(a, v); }
    __device__ __forceinline__ void merge(const ValAcc &o) { a = RedOp<OP>::merge(a, o.a); }
    __device__ __forceinline__ ValAcc shfl(int off) const
    {
        ValAcc o;
        o.a = __shfl_xor(a, off, kWave);
        return o;
    }
    __device__ __forceinline__ void save(double *p) const { p[0] = a; }
    __device__ __forceinline__ void load(const double *p) { a = p[0]; }
    __device__ __forceinline__ void finish(const FinParams &f, int64_t e) const { put_value(f, e, a); }
};

template <typename T, int OP>
struct PickAcc {
    static constexpr int NP = 2;
    T a;
    int64_t i; // -1: nothing seen yet
    __device__ __forceinline__ void init_empty()
    {
        a = (T)RedOp<OP>::identity();
        i = -1;
    }
    __device__ __forceinline__ void init(const T *, int64_t, int64_t) { init_empty(); }
    template <int VEC>
    __device__ __forceinline__ void init_vals(const T (&)[VEC], bool) { init_empty(); }
    __device__ __forceinline__ void init_own(T, bool) { init_empty(); }
    __device__ __forceinline__ void fold(T v, int64_t r)
    {
        const T w = Pick<OP>::see(v);
        if (i < 0 || Pick<OP>::better(w, a)) {
            a = w;
            i = r;
        }
    }
    __device__ __forceinline__ void merge(const PickAcc &o)
    {
        const bool take = o.i >= 0 && (i < 0 || Pick<OP>::better(o.a, a) || (Pick<OP>::same(o.a, a) && o.i < i));
        if (take) {
            a = o.a;
            i = o.i;
        }
    }
    __device__ __forceinline__ PickAcc shfl(int off) const
    {
        PickAcc o;
        o.a = __shfl_xor(a, off, kWave);
        o.i = __shfl_xor(i, off, kWave);
        return o;
    }
    __device__ __forceinline__ void save(double *p) const
    {
        p[0] = (double)a;
        p[1] = __longlong_as_double(i);
    }
    __device__ __forceinline__ void load(const double *p)
    {
        a = (T)p[0];
        i = __double_as_longlong(p[1]);
    }
    __device__ __forceinline__ void finish(const FinParams &f, int64_t e) const
    {
        put_value(f, e, (double)a);
        if (f.idx) f.idx[e] = i < 0 ? 0 : i;
    }
};

template <typename T>
struct MomAcc {
    static constexpr int NP = 4;
    Mom m;
    double k;
    __device__ __forceinline__ void init_empty()
    {
        m = {0.0, 0.0, 0.0};
        k = 0.0;
    }
    __device__ __forceinline__ void init(const T *p0, int64_t cnt, int64_t stride)
    {
        m = {0.0, 0.0, 0.0};
        k = pick_shift(p0, cnt, stride);
    }
    // shift = the first finite value among the lane's own elements (GROUP plan, inner == 1: no extra loads)
    template <int VEC>
    __device__ __forceinline__ void init_vals(const T (&v)[VEC], bool have)
    {
        m = {0.0, 0.0, 0.0};
        k = 0.0;
        if (have) {
#pragma unroll
            for (int j = VEC - 1; j >= 0; --j) {
                const double d = (double)v[j];
                k = (d - d == 0.0) ? d : k;
            }
        }
    }
    // shift = this accumulator's own first value (one element per lane and column)
    __device__ __forceinline__ void init_own(T v, bool have)
    {
        m = {0.0, 0.0, 0.0};
        const double d = (double)v;
        k = (have && d - d == 0.0) ? d : 0.0;
    }
    __device__ __forceinline__ void fold(T v, int64_t) { mom_fold(m, v, k); }
    // the two sides may use different shifts: re-centre the other side's sums on ours
    // (sum(x - K1) = sum(x - K2) + n (K2 - K1), likewise for the squares)
    __device__ __forceinline__ void merge(const MomAcc &o)
    {
        const double kk = (m.n == 0.0) ? o.k : k;
        const double d = o.k - kk;
        m.s += o.m.s + o.m.n * d;
        m.q += o.m.q + (2.0 * d) * o.m.s + o.m.n * d * d;
        m.n += o.m.n;
        k = kk;
    }
    __device__ __forceinline__ MomAcc shfl(int off) const
    {
        MomAcc o;
        o.m.n = __shfl_xor(m.n, off, kWave);
        o.m.s = __shfl_xor(m.s, off, kWave);
        o.m.q = __shfl_xor(m.q, off, kWave);
        o.k = __shfl_xor(k, off, kWave);
        return o;
    }
    __device__ __forceinline__ void save(double *p) const
    {
        p[0] = m.n;
        p[1] = m.s;
        p[2] = m.q;
        p[3] = k;
    }
    __device__ __forceinline__ void load(const double *p)
    {
        m = {p[0], p[1], p[2]};
        k = p[3];
    }
    __device__ __forceinline__ void finish(const FinParams &f, int64_t e) const
    {
        if (f.stat < 0) {
            double *o = static_cast<double *>(f.out) + 4 * e;
            o[0] = m.n;
            o[1] = m.s;
            o[2] = m.q;
            o[3] = k;
            return;
        }
        // mean = K + s / n; var = (q - s^2 / n) / n [* n / (n - 1)]  (`reduce.py:591-594`, `:679-684`)
        const int kind = f.stat & 3;
        double v;
        if (kind == NFM_STAT_MEAN) {
            v = k + m.s / m.n;
        } else {
            // (q - s^2 / n) / n * (n / (n - 1)) == (q - s^2 / n) / (n - 1): one division each
            const double ss = m.q - m.s * (m.s / m.n);
            const double den = (f.stat & NFM_STAT_UNBIASED) ? m.n - 1.0 : m.n;
            v = (ss < 0.0 ? 0.0 : ss) / den;
            if (den <= 0.0) v = __builtin_nan(""); // fewer values than degrees of freedom (0 / 0 upstream)
            if (kind == NFM_STAT_STD) v = sqrt(v);
        }
        // without omitnan a NaN anywhere in the reduced slice propagates
        if (!(f.stat & NFM_STAT_OMITNAN) && m.n != (double)f.red) v = __builtin_nan("");
        put_value(f, e, v);
    }
};

template <typename T, int VEC>
struct LoadV;
template <typename T>
struct LoadV<T, 1> {
    __device__ static __forceinline__ void ld(const T *p, T (&v)[1]) { v[0] = __builtin_nontemporal_load(p); }
};
template <>
struct LoadV<float, 4> {
    __device__ static __forceinline__ void ld(const float *p, float (&v)[4])
    {
        const VecOf<float>::type t = __builtin_nontemporal_load(reinterpret_cast<const VecOf<float>::type *>(p));
        v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
    }
};
template <>
struct LoadV<double, 2> {
    __device__ static __forceinline__ void ld(const double *p, double (&v)[2])
    {
        const VecOf<double>::type t = __builtin_nontemporal_load(reinterpret_cast<const VecOf<double>::type *>(p));
        v[0] = t[0], v[1] = t[1];
    }
};

// ---- GROUP: small slabs (red * inner elements, at most 4 KiB).  G = 2^lgG lanes share a slab:
// lane `sub` of the group reads vectors sub, sub + G, ... of it (G * 16 B >= 128 B contiguous
// per group and load).  G * VEC is a multiple of inner, so a lane component always meets the
// same column; lanes sub and sub + inner / VEC (or, for inner < VEC, components k and k + inner)
// are merged by xor-shuffles at the end.  One vector per lane (IT == 1): four slabs in flight
// per group; longer slabs: one slab per group, four loads in flight.
template <typename T, class Acc, int VEC, bool ONE>
__device__ __forceinline__ void group_finish(Acc (&acc)[ONE ? 1 : VEC], int G, int sub, int inner, int64_t slab,
                                             bool valid, const FinParams &f)
{
    if constexpr (ONE) { // inner == 1: a single accumulator per lane
        for (int off = G >> 1; off > 0; off >>= 1) acc[0].merge(acc[0].shfl(off));
        if (valid && sub == 0) acc[0].finish(f, slab);
    } else {
        const int inner_v = inner >= VEC ? inner / VEC : 1;
        if constexpr (VEC == 4) {
            if (inner == 2) { // components k and k + 2 share a column
                acc[0].merge(acc[2]);
                acc[1].merge(acc[3]);
            }
        }
        for (int off = G >> 1; off >= inner_v; off >>= 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if (inner >= VEC || k < inner) acc[k].merge(acc[k].shfl(off));
        }
        if (valid && sub < inner_v) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if (inner >= VEC || k < inner) acc[k].finish(f, slab * inner + sub * VEC + k);
        }
    }
}

template <typename T, class Acc, int VEC, bool ONE>
__global__ __launch_bounds__(256) void reduce_group_k(const T *__restrict__ x, int64_t outer, int L, int inner,
                                                      int lgG, int IT, FinParams f)
{
    constexpr int NA = ONE ? 1 : VEC; // accumulators per lane
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int G = 1 << lgG, sub = lane & (G - 1), grp = lane >> lgG, spw = 64 >> lgG;
    const int pos0 = sub * VEC;
    int r0[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) r0[k] = ONE ? pos0 + k : (pos0 + k) / inner;
    if (IT == 1) {
        constexpr int P = 4; // slabs in flight per lane group
        const int64_t s0 = wave * (int64_t)(spw * P) + grp;
        const bool in_slab = pos0 < L;
        T vals[P][VEC];
        Acc keep;
        keep.init_empty();
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t sl = s0 + (int64_t)p * spw;
            if (sl < outer && in_slab) LoadV<T, VEC>::ld(x + sl * L + pos0, vals[p]);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int64_t sl = s0 + (int64_t)p * spw;
            const bool have = sl < outer && in_slab;
            Acc acc[NA];
            if constexpr (ONE) {
                acc[0].template init_vals<VEC>(vals[p], have);
            } else { // one element per lane and column: its own value is the shift
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k].init_own(vals[p][k], have);
            }
            if (have) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[ONE ? 0 : k].fold(vals[p][k], r0[k]);
            }
            if constexpr (ONE) {
                // after the xor-merge every lane of the group holds the row's result: lane `p` of
                // the group keeps row p, so that the (division-heavy) finish runs once, not P times
                for (int off = G >> 1; off > 0; off >>= 1) acc[0].merge(acc[0].shfl(off));
                if (G >= P) {
                    if (p == 0 || sub == p) keep = acc[0];
                } else if (sl < outer && sub == 0) {
                    acc[0].finish(f, sl);
                }
            } else {
                group_finish<T, Acc, VEC, ONE>(acc, G, sub, inner, sl, sl < outer, f);
            }
        }
        if constexpr (ONE) {
            const int64_t sl = s0 + (int64_t)sub * spw;
            if (G >= P && sub < P && sl < outer) keep.finish(f, sl);
        }
    } else {
        const int64_t sl = wave * (int64_t)spw + grp;
        const bool valid = sl < outer; // pos0 < L always: the slab has more than G vectors
        const int step = G * VEC, rstep = ONE ? step : step / inner;
        const T *slab = x + sl * L;
        Acc acc[NA];
        T v0[VEC], v1[VEC], v2[VEC], v3[VEC];
        if (valid) LoadV<T, VEC>::ld(slab + pos0, v0);
        if constexpr (ONE) {
            acc[0].template init_vals<VEC>(v0, valid);
        } else { // shift from the component's own column (first finite of its first elements)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if (valid) acc[k].init(slab + (pos0 + k) % inner, L / inner, inner);
                else acc[k].init_empty();
            }
        }
        if (valid) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[ONE ? 0 : k].fold(v0[k], r0[k]);
            int pos = pos0 + step, rr = rstep;
            for (; pos + 3 * step < L; pos += 4 * step, rr += 4 * rstep) {
                LoadV<T, VEC>::ld(slab + pos, v0);
                LoadV<T, VEC>::ld(slab + pos + step, v1);
                LoadV<T, VEC>::ld(slab + pos + 2 * step, v2);
                LoadV<T, VEC>::ld(slab + pos + 3 * step, v3);
                if constexpr (ONE) { // keep the positions increasing inside the one accumulator
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[0].fold(v0[k], r0[k] + rr);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[0].fold(v1[k], r0[k] + rr + rstep);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[0].fold(v2[k], r0[k] + rr + 2 * rstep);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[0].fold(v3[k], r0[k] + rr + 3 * rstep);
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        acc[k].fold(v0[k], r0[k] + rr);
                        acc[k].fold(v1[k], r0[k] + rr + rstep);
                        acc[k].fold(v2[k], r0[k] + rr + 2 * rstep);
                        acc[k].fold(v3[k], r0[k] + rr + 3 * rstep);
                    }
                }
            }
            for (; pos < L; pos += step, rr += rstep) {
                LoadV<T, VEC>::ld(slab + pos, v0);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[ONE ? 0 : k].fold(v0[k], r0[k] + rr);
            }
        }
        group_finish<T, Acc, VEC, ONE>(acc, G, sub, inner, sl, valid, f);
    }
}

// ---- FLAT: one wavefront per (outer index, chunk of the slab)
enum { COMB_ALL = 0, COMB_SHFL = 1, COMB_SUBV = 2, COMB_LDS = 3 };
template <typename T, class Acc, int VEC>
__global__ __launch_bounds__(256) void reduce_flat_k(const T *__restrict__ x, int64_t outer, int64_t red,
                                                     int64_t inner, int A, int comb, int64_t chunk_len,
                                                     int nchunk, double *__restrict__ partial, FinParams f)
{
    __shared__ Acc lds[4][64 * VEC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t item = (int64_t)blockIdx.x * 4 + wv;
    const bool live = item < outer * nchunk;
    const int64_t o = live ? item / nchunk : 0;
    const int64_t c = live ? item - o * nchunk : 0;
    const int64_t L = red * inner;
    const T *slab = x + o * L;
    const int64_t start = c * chunk_len;
    int64_t end = start + chunk_len;
    if (end > L) end = L;
    const bool active = live && lane < A;
    const int64_t step = (int64_t)A * VEC;
    const int64_t rstep = step / inner;

    Acc acc[VEC];
    int64_t r[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        if (active) {
            const int64_t col = (int64_t)(lane * VEC + k) % inner;
            acc[k].init(slab + col, red, inner);
            r[k] = (start + lane * VEC + k) / inner;
        } else {
            acc[k].init_empty();
            r[k] = 0;
        }
    }
    if (active) {
        int64_t pos = start + (int64_t)lane * VEC;
        for (; pos + 3 * step + VEC <= end; pos += 4 * step) {
            T v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            LoadV<T, VEC>::ld(slab + pos, v0);
            LoadV<T, VEC>::ld(slab + pos + step, v1);
            LoadV<T, VEC>::ld(slab + pos + 2 * step, v2);
            LoadV<T, VEC>::ld(slab + pos + 3 * step, v3);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                acc[k].fold(v0[k], r[k]);
                acc[k].fold(v1[k], r[k] + rstep);
                acc[k].fold(v2[k], r[k] + 2 * rstep);
                acc[k].fold(v3[k], r[k] + 3 * rstep);
                r[k] += 4 * rstep;
            }
        }
        for (; pos + VEC <= end; pos += step) {
            T v0[VEC];
            LoadV<T, VEC>::ld(slab + pos, v0);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                acc[k].fold(v0[k], r[k]);
                r[k] += rstep;
            }
        }
        if (VEC > 1 && pos < end) { // ragged end of the slab, shorter than one vector
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if (pos + k < end) acc[k].fold(slab[pos + k], r[k]);
        }
    }

    const int64_t nout = outer * inner;
    if (comb == COMB_ALL) { // inner == 1: everything meets in one output
        Acc t = acc[0];
#pragma unroll
        for (int k = 1; k < VEC; ++k) t.merge(acc[k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t.merge(t.shfl(off));
        if (lane == 0 && live) {
            if (nchunk == 1) t.finish(f, o);
            else t.save(partial + (c * nout + o) * Acc::NP);
        }
    } else if (comb == COMB_SHFL) { // inner / VEC a power of two: lanes l and l + inner / VEC share columns
        const int inner_v = (int)inner / VEC;
        for (int off = 32; off >= inner_v; off >>= 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k].merge(acc[k].shfl(off));
        }
        if (lane < inner_v && live) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int64_t e = o * inner + lane * VEC + k;
                if (nchunk == 1) acc[k].finish(f, e);
                else acc[k].save(partial + (c * nout + e) * Acc::NP);
            }
        }
    } else if (comb == COMB_SUBV) { // inner divides VEC: components k and k + inner share a column
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (k >= (int)inner) {
                // (VEC <= 4: the only cases are inner == 2 with k = 2, 3)
                if ((k & 1) == 0) acc[0].merge(acc[k]);
                else acc[1].merge(acc[k]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            acc[0].merge(acc[0].shfl(off));
            if (VEC > 1) acc[VEC > 1 ? 1 : 0].merge(acc[VEC > 1 ? 1 : 0].shfl(off));
        }
        if (lane == 0 && live) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if (k < (int)inner) {
                    const int64_t e = o * inner + k;
                    if (nchunk == 1) acc[k].finish(f, e);
                    else acc[k].save(partial + (c * nout + e) * Acc::NP);
                }
            }
        }
    } else { // any other small inner: through LDS, floor(64 / inner) lanes per column, then one
        const int nent = A * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) lds[wv][lane * VEC + k] = acc[k];
        __syncthreads();
        if (inner <= 64) {
            const int Q = 64 / (int)inner, col = lane % (int)inner, q = lane / (int)inner;
            Acc t;
            t.init_empty();
            if (q < Q) {
                for (int j = col + q * (int)inner; j < nent; j += Q * (int)inner) t.merge(lds[wv][j]);
            }
            __syncthreads();
            lds[wv][lane] = t;
            __syncthreads();
            if (lane < inner && live) {
                t = lds[wv][lane];
                for (int qq = 1; qq < Q; ++qq) t.merge(lds[wv][lane + qq * (int)inner]);
                const int64_t e = o * inner + lane;
                if (nchunk == 1) t.finish(f, e);
                else t.save(partial + (c * nout + e) * Acc::NP);
            }
        } else if (live) {
            for (int col = lane; col < inner; col += 64) {
                Acc t = lds[wv][col];
                for (int j = col + (int)inner; j < nent; j += (int)inner) t.merge(lds[wv][j]);
                const int64_t e = o * inner + col;
                if (nchunk == 1) t.finish(f, e);
                else t.save(partial + (c * nout + e) * Acc::NP);
            }
        }
    }
}

// ---- COL: lanes along inner (VEC adjacent columns each), chunks of the reduced axis on grid.y
template <typename T, class Acc, int VEC>
__global__ __launch_bounds__(256) void reduce_col_k(const T *__restrict__ x, int64_t outer, int64_t red,
                                                    int64_t inner, int64_t chunk_len, int nchunk,
                                                    double *__restrict__ partial, FinParams f)
{
    const int64_t inner_v = inner / VEC;
    const int64_t ev = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ev >= outer * inner_v) return;
    const int64_t o = ev / inner_v, iv = ev - o * inner_v;
    const int64_t c = blockIdx.y;
    const T *p = x + o * red * inner + iv * VEC;
    const int64_t r0 = c * chunk_len;
    int64_t r1 = r0 + chunk_len;
    if (r1 > red) r1 = red;
    Acc acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        if (red > 0) acc[k].init(p + k, red, inner);
        else acc[k].init_empty();
    }
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
        T v0[VEC], v1[VEC], v2[VEC], v3[VEC];
        LoadV<T, VEC>::ld(p + r * inner, v0);
        LoadV<T, VEC>::ld(p + (r + 1) * inner, v1);
        LoadV<T, VEC>::ld(p + (r + 2) * inner, v2);
        LoadV<T, VEC>::ld(p + (r + 3) * inner, v3);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            acc[k].fold(v0[k], r);
            acc[k].fold(v1[k], r + 1);
            acc[k].fold(v2[k], r + 2);
            acc[k].fold(v3[k], r + 3);
        }
    }
    for (; r < r1; ++r) {
        T v0[VEC];
        LoadV<T, VEC>::ld(p + r * inner, v0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k].fold(v0[k], r);
    }
    const int64_t nout = outer * inner;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int64_t e = o * inner + iv * VEC + k;
        if (nchunk == 1) acc[k].finish(f, e);
        else acc[k].save(partial + (c * nout + e) * Acc::NP);
    }
}

// ---- second stage: fold the chunk partials of every output in a fixed order.
// Few chunks: one lane per output.  Many chunks: one wavefront per output, lane l folds chunks
// l, l + 64, ... and the 64 lane results are merged by shuffles.
template <class Acc>
__global__ __launch_bounds__(256) void reduce_fold_k(const double *__restrict__ partial, int64_t nout, int nchunk,
                                                     FinParams f)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nout) return;
    Acc t;
    t.load(partial + e * Acc::NP);
    for (int c = 1; c < nchunk; ++c) {
        Acc u;
        u.load(partial + ((int64_t)c * nout + e) * Acc::NP);
        t.merge(u);
    }
    t.finish(f, e);
}

template <class Acc>
__global__ __launch_bounds__(256) void reduce_fold_wave_k(const double *__restrict__ partial, int64_t nout,
                                                          int nchunk, FinParams f)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    Acc t;
    t.init_empty();
    if (e < nout) {
        int c = lane;
        for (; c + 192 < nchunk; c += 256) {
            Acc u0, u1, u2, u3;
            u0.load(partial + ((int64_t)c * nout + e) * Acc::NP);
            u1.load(partial + ((int64_t)(c + 64) * nout + e) * Acc::NP);
            u2.load(partial + ((int64_t)(c + 128) * nout + e) * Acc::NP);
            u3.load(partial + ((int64_t)(c + 192) * nout + e) * Acc::NP);
            t.merge(u0);
            t.merge(u1);
            t.merge(u2);
            t.merge(u3);
        }
        for (; c < nchunk; c += 64) {
            Acc u;
            u.load(partial + ((int64_t)c * nout + e) * Acc::NP);
            t.merge(u);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t.merge(t.shfl(off));
    if (lane == 0 && e < nout) t.finish(f, e);
}

// ---- plan: a pure function of (element size, shape, 16-byte alignment of the base pointer)
enum { PLAN_GROUP = 0, PLAN_FLAT = 1, PLAN_COL = 2 };
struct DimPlan {
    int kind, vec, lgG, IT, A, comb, nchunk;
    int64_t chunk_len;
};

static int64_t gcd64(int64_t a, int64_t b)
{
    while (b) {
        const int64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

static DimPlan plan_dim(int esz, int64_t outer, int64_t red, int64_t inner, bool aligned)
{
    const int VF = 16 / esz;
    const int64_t L = red * inner;
    DimPlan p{};
    p.nchunk = 1;
    p.A = 64;
    {
        // GROUP: slabs of at most 4 KiB whose columns fold by shuffles
        const int v = (aligned && L % VF == 0) ? VF : 1;
        const int64_t iv = inner / v;
        const bool cols_ok = inner == 1 || (inner < v && v % inner == 0) ||
                             (inner % v == 0 && (iv & (iv - 1)) == 0);
        if (cols_ok && L * esz <= 4096) {
            const int64_t nv = (L + v - 1) / v;
            const int gmax = 128 / (v * esz); // lanes per 128 contiguous bytes
            int lg = 0;
            while ((1 << lg) < nv && (1 << lg) < gmax) ++lg;
            while (inner >= v && (1 << lg) < iv) ++lg; // the group must span all columns
            if ((1 << lg) <= 64) {
                p.kind = PLAN_GROUP;
                p.vec = v;
                p.lgG = lg;
                p.IT = (int)((nv + (1 << lg) - 1) >> lg);
                if (p.IT < 1) p.IT = 1;
                return p;
            }
        }
    }
    // FLAT: the slab must start 16-byte aligned for every outer index
    const int vflat = (aligned && (L % VF == 0 || outer == 1)) ? VF : 1;
    const int64_t inner_g = inner / gcd64(inner, vflat);
    if (inner_g <= 64 && L >= 64 * vflat) {
        p.kind = PLAN_FLAT;
        p.vec = vflat;
        p.A = (int)(64 - 64 % inner_g);
        if (inner == 1) p.comb = COMB_ALL;
        else if (inner % vflat == 0 && ((inner / vflat) & (inner / vflat - 1)) == 0) p.comb = COMB_SHFL;
        else if (inner < vflat && vflat % inner == 0) p.comb = COMB_SUBV;
        else p.comb = COMB_LDS;
        const int64_t step = (int64_t)p.A * vflat;
        int64_t want = outer >= 8192 ? 1 : (8192 + outer - 1) / outer; // ~one wave per SIMD slot
        const int64_t most = L / (step * 16);                         // >= 16 iterations per chunk
        if (want > most) want = most;
        if (want < 1) want = 1;
        int64_t len = (L + want - 1) / want;
        len = (len + step - 1) / step * step;
        p.chunk_len = len;
        p.nchunk = (int)((L + len - 1) / len);
        return p;
    }
    p.kind = PLAN_COL;
    p.vec = (aligned && inner % VF == 0) ? VF : 1;
    const int64_t lanes = outer * (inner / p.vec);
    int64_t want = 1;
    if (lanes < (1 << 19) && red >= 64) {
        want = (1 << 20) / (lanes > 0 ? lanes : 1);
        if (want > red / 16) want = red / 16;
        if (want > 65535) want = 65535;
        if (want < 1) want = 1;
    }
    p.chunk_len = (red + want - 1) / want;
    if (p.chunk_len < 1) p.chunk_len = 1;
    p.nchunk = (int)((red + p.chunk_len - 1) / p.chunk_len);
    if (p.nchunk < 1) p.nchunk = 1;
    return p;
}

static size_t plan_workspace(const DimPlan &p, int64_t outer, int64_t inner, int np)
{
    return p.nchunk > 1 ? (size_t)p.nchunk * (size_t)(outer * inner) * (size_t)np * sizeof(double) : 0;
}

static size_t dim_workspace_bytes(int esz, int64_t outer, int64_t red, int64_t inner, int np)
{
    const size_t a = plan_workspace(plan_dim(esz, outer, red, inner, true), outer, inner, np);
    const size_t b = plan_workspace(plan_dim(esz, outer, red, inner, false), outer, inner, np);
    return a > b ? a : b;
}

template <typename T, class Acc, int VEC>
static int run_plan_v(const DimPlan &p, const T *x, int64_t outer, int64_t red, int64_t inner, double *ws,
                      const FinParams &f, hipStream_t s)
{
    const int64_t nout = outer * inner;
    if (p.kind == PLAN_GROUP) {
        const int64_t slabs_per_wave = (int64_t)(64 >> p.lgG) * (p.IT == 1 ? 4 : 1);
        const int64_t waves = (outer + slabs_per_wave - 1) / slabs_per_wave;
        const int64_t nblk = (waves + 3) / 4;
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
        if (inner == 1)
            hipLaunchKernelGGL((reduce_group_k<T, Acc, VEC, true>), dim3((unsigned)nblk), dim3(256), 0, s, x, outer,
                               (int)red, 1, p.lgG, p.IT, f);
        else
            hipLaunchKernelGGL((reduce_group_k<T, Acc, VEC, false>), dim3((unsigned)nblk), dim3(256), 0, s, x, outer,
                               (int)(red * inner), (int)inner, p.lgG, p.IT, f);
        return launch_status();
    }
    if (p.kind == PLAN_FLAT) {
        const int64_t nblk = (outer * p.nchunk + 3) / 4;
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
        hipLaunchKernelGGL((reduce_flat_k<T, Acc, VEC>), dim3((unsigned)nblk), dim3(256), 0, s, x, outer, red, inner,
                           p.A, p.comb, p.chunk_len, p.nchunk, ws, f);
    } else {
        const int64_t nblk = (outer * (inner / VEC) + 255) / 256;
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
        hipLaunchKernelGGL((reduce_col_k<T, Acc, VEC>), dim3((unsigned)nblk, (unsigned)p.nchunk, 1), dim3(256), 0, s,
                           x, outer, red, inner, p.chunk_len, p.nchunk, ws, f);
    }
    if (p.nchunk >= 64) {
        const int64_t nblk = (nout + 3) / 4;
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
        hipLaunchKernelGGL((reduce_fold_wave_k<Acc>), dim3((unsigned)nblk), dim3(256), 0, s, ws, nout, p.nchunk, f);
    } else if (p.nchunk > 1) {
        const int64_t nblk = (nout + 255) / 256;
        if (nblk > 0x7fffffffLL) return NFM_ESIZE;
        hipLaunchKernelGGL((reduce_fold_k<Acc>), dim3((unsigned)nblk), dim3(256), 0, s, ws, nout, p.nchunk, f);
    }
    return launch_status();
}

template <typename T, class Acc>
static int run_dim(const void *x, int64_t outer, int64_t red, int64_t inner, void *ws, size_t ws_bytes,
                   const FinParams &f, hipStream_t s)
{
    const bool aligned = reinterpret_cast<uintptr_t>(x) % 16 == 0;
    const DimPlan p = plan_dim((int)sizeof(T), outer, red, inner, aligned);
    if (plan_workspace(p, outer, inner, Acc::NP) > 0) {
        if (ws == nullptr) return NFM_EINVAL;
        if (reinterpret_cast<uintptr_t>(ws) % 8 != 0) return NFM_EALIGN;
        if (ws_bytes < plan_workspace(p, outer, inner, Acc::NP)) return NFM_EWORKSPACE;
    }
    const T *xp = static_cast<const T *>(x);
    double *wp = static_cast<double *>(ws);
    constexpr int VF = VecOf<T>::N;
    if (p.vec == VF) return run_plan_v<T, Acc, VF>(p, xp, outer, red, inner, wp, f, s);
    return run_plan_v<T, Acc, 1>(p, xp, outer, red, inner, wp, f, s);
}

template <typename T, int OP>
static int reduce_dim_t(const void *x, int64_t outer, int64_t red, int64_t inner, void *ws, size_t ws_bytes,
                        const FinParams &f, hipStream_t s)
{
    if constexpr (!RedOp<OP>::is_sum) {
        if (f.idx) return run_dim<T, PickAcc<T, OP>>(x, outer, red, inner, ws, ws_bytes, f, s);
    }
    return run_dim<T, ValAcc<T, OP>>(x, outer, red, inner, ws, ws_bytes, f, s);
}

static int check_dim_args(int dtype, int64_t outer, int64_t red, int64_t inner, const void *x, const void *out)
{
    if (dtype != NFM_F32 && dtype != NFM_F64) return NFM_EDTYPE;
    if (outer < 0 || red < 0 || inner < 0) return NFM_EINVAL;
    if (outer == 0 || inner == 0) return NFM_OK;
    if (out == nullptr || (red > 0 && x == nullptr)) return NFM_EINVAL;
    if (reinterpret_cast<uintptr_t>(x) % (dtype == NFM_F32 ? 4 : 8) != 0) return NFM_EALIGN;
    if (red > 0 && (outer > INT64_MAX / red || outer * red > INT64_MAX / inner)) return NFM_ESIZE;
    return 1; // go on
}

} // namespace nfm

using namespace nfm;

extern "C" {

size_t nfm_reduce_dim_workspace_bytes(int dtype, int op, int64_t outer, int64_t red, int64_t inner, int want_idx)
{
    if ((dtype != NFM_F32 && dtype != NFM_F64) || outer <= 0 || red < 0 || inner <= 0) return 0;
    const bool pick = want_idx && (op == NFM_RED_NANMAX || op == NFM_RED_NANMIN || op == NFM_RED_MAX ||
                                   op == NFM_RED_MIN);
    return dim_workspace_bytes(dtype == NFM_F32 ? 4 : 8, outer, red, inner, pick ? 2 : 1);
}

int nfm_reduce_dim(int dtype, int op, int out_dtype, int64_t outer, int64_t red, int64_t inner, const void *x,
                   void *workspace, size_t workspace_bytes, void *out, int64_t *idx, void *stream)
{
    if (out_dtype != NFM_F32 && out_dtype != NFM_F64) return NFM_EDTYPE;
    const int st = check_dim_args(dtype, outer, red, inner, x, out);
    if (st <= 0) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const FinParams f{out, out_dtype, idx, -1, red};
    if (dtype == NFM_F32) {
        NFM_SWITCH_OP(op, return (reduce_dim_t<float, OP>(x, outer, red, inner, workspace, workspace_bytes, f, s)))
    } else {
        NFM_SWITCH_OP(op, return (reduce_dim_t<double, OP>(x, outer, red, inner, workspace, workspace_bytes, f, s)))
    }
    return NFM_EINVAL;
}

size_t nfm_reduce_moments_workspace_bytes(int dtype, int64_t outer, int64_t red, int64_t inner)
{
    if ((dtype != NFM_F32 && dtype != NFM_F64) || outer <= 0 || red < 0 || inner <= 0) return 0;
    const size_t a = dim_workspace_bytes(dtype == NFM_F32 ? 4 : 8, outer, red, inner, 4);
    const size_t b = nfm_reduce_workspace_bytes();
    return a > b ? a : b;
}

int nfm_reduce_moments(int dtype, int64_t outer, int64_t red, int64_t inner, const void *x, void *workspace,
                       size_t workspace_bytes, double *out, void *stream)
{
    const int st = check_dim_args(dtype, outer, red, inner, x, out);
    if (st <= 0) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (outer == 1 && inner == 1) { // full reduction: the two-kernel streaming path of nfm_reduce.hip
        if (workspace == nullptr) return NFM_EINVAL;
        if (workspace_bytes < nfm_reduce_workspace_bytes()) return NFM_EWORKSPACE;
        return moments_all_launch(dtype, x, red, workspace, out, s);
    }
    const FinParams f{out, NFM_F64, nullptr, -1, red};
    if (dtype == NFM_F32) return run_dim<float, MomAcc<float>>(x, outer, red, inner, workspace, workspace_bytes, f, s);
    return run_dim<double, MomAcc<double>>(x, outer, red, inner, workspace, workspace_bytes, f, s);
}

int nfm_reduce_stat(int dtype, int stat, int out_dtype, int64_t outer, int64_t red, int64_t inner, const void *x,
                    void *workspace, size_t workspace_bytes, void *out, void *stream)
{
    if (out_dtype != NFM_F32 && out_dtype != NFM_F64) return NFM_EDTYPE;
    if (stat < 0 || (stat & 3) > NFM_STAT_STD || (stat & ~(3 | NFM_STAT_OMITNAN | NFM_STAT_UNBIASED))) return NFM_EINVAL;
    const int st = check_dim_args(dtype, outer, red, inner, x, out);
    if (st <= 0) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const FinParams f{out, out_dtype, nullptr, stat, red};
    if (dtype == NFM_F32) return run_dim<float, MomAcc<float>>(x, outer, red, inner, workspace, workspace_bytes, f, s);
    return run_dim<double, MomAcc<double>>(x, outer, red, inner, workspace, workspace_bytes, f, s);
}

} // extern "C"

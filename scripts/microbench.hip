// microbench.hip -- kernel-structure experiments for the headline path (sym_solve 4x4
// fp32 AoS, n = 1e8) on one MI355X, timed with hipEvents, interleaved rounds in ONE
// process (cdna_hip_programming.md rule 24).  Not part of the product; it includes the
// product headers so that every variant runs the SAME arithmetic (bit-compared to the
// library path).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I. scripts/microbench.hip \
//         -Lnitorch_fastmath_amd -lnfm_hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#include <functional>
#include "../nitorch_fastmath_amd/csrc/nfm_record_kernel.hpp"
#include "../nitorch_fastmath_amd/csrc/nfm_smallmat.hpp"

using namespace nfm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f4 f4u __attribute__((aligned(8)));

__global__ void fill_kernel(float *mat, float *vec, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned s = (unsigned)(i * 2654435761u) | 1u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int c = 0; c < 4; ++c) mat[i * 10 + c] = 2.0f + rnd();
    for (int c = 4; c < 10; ++c) mat[i * 10 + c] = 0.5f * rnd();
    for (int c = 0; c < 4; ++c) vec[i * 4 + c] = rnd();
}

__global__ void fill6_kernel(float *mat, float *vec, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned s = (unsigned)(i * 2654435761u) | 1u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int c = 0; c < 6; ++c) mat[i * 21 + c] = 2.0f + rnd();
    for (int c = 6; c < 21; ++c) mat[i * 21 + c] = 0.4f * rnd();
    for (int c = 0; c < 6; ++c) vec[i * 6 + c] = rnd();
}

__global__ void fill8_kernel(double *a8, double *m3, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned s = (unsigned)(i * 2654435761u) | 1u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (int c = 0; c < 64; ++c) a8[i * 64 + c] = 2.0 * rnd() + ((c / 8 == c % 8) ? 8.0 : 0.0);
    for (int c = 0; c < 3; ++c) m3[i * 6 + c] = 2.0 + rnd();
    for (int c = 3; c < 6; ++c) m3[i * 6 + c] = 0.5 * rnd();
}

// ---- ceiling: same bytes (56 B in, 16 B out per element), no transpose, trivial math
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const f4 *__restrict__ mat, const f4 *__restrict__ vec,
                                                   f4 *__restrict__ out, int64_t n)
{
    // block = 512 elements: 1280 mat float4 (5 per thread), 512 vec float4 (2 per thread)
    const int64_t b = blockIdx.x;
    const int t = threadIdx.x;
    if ((b + 1) * 512 > n) return;
    const f4 *pm = mat + b * 1280;
    const f4 *pv = vec + b * 512;
    f4 a[5], v[2];
#pragma unroll
    for (int k = 0; k < 5; ++k) a[k] = NT ? __builtin_nontemporal_load(pm + t + 256 * k) : pm[t + 256 * k];
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] = NT ? __builtin_nontemporal_load(pv + t + 256 * k) : pv[t + 256 * k];
    f4 s = a[0] + a[1] + a[2] + a[3] + a[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        f4 r = v[k] + s;
        if (NT) __builtin_nontemporal_store(r, out + b * 512 + t + 256 * k);
        else out[b * 512 + t + 256 * k] = r;
    }
}

// ---- 50/50 read/write ceiling: plain streaming copy, 4 x 16 B in flight per lane
template <bool NT>
__global__ __launch_bounds__(256) void copy5050_kernel(const f4 *__restrict__ in, f4 *__restrict__ out, int64_t nvec)
{
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (q + 3 * stride >= nvec) return;
    f4 v0, v1, v2, v3;
    if (NT) {
        v0 = __builtin_nontemporal_load(in + q); v1 = __builtin_nontemporal_load(in + q + stride);
        v2 = __builtin_nontemporal_load(in + q + 2 * stride); v3 = __builtin_nontemporal_load(in + q + 3 * stride);
        __builtin_nontemporal_store(v0, out + q); __builtin_nontemporal_store(v1, out + q + stride);
        __builtin_nontemporal_store(v2, out + q + 2 * stride); __builtin_nontemporal_store(v3, out + q + 3 * stride);
    } else {
        v0 = in[q]; v1 = in[q + stride]; v2 = in[q + 2 * stride]; v3 = in[q + 3 * stride];
        out[q] = v0; out[q + stride] = v1; out[q + 2 * stride] = v2; out[q + 3 * stride] = v3;
    }
}

// ---- direct: every lane loads its own 40-byte record (x4, x4, x2), no LDS
template <int WAVES, bool NT>
__global__ __launch_bounds__(256, WAVES) void direct_kernel(const float *__restrict__ mat,
                                                            const float *__restrict__ vec,
                                                            float *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = mat + i * 10;
    f4 a = NT ? __builtin_nontemporal_load((const f4u *)p) : *(const f4u *)p;
    f4 b = NT ? __builtin_nontemporal_load((const f4u *)(p + 4)) : *(const f4u *)(p + 4);
    f2 c = NT ? __builtin_nontemporal_load((const f2 *)(p + 8)) : *(const f2 *)(p + 8);
    f4 vv = NT ? __builtin_nontemporal_load((const f4 *)(vec + i * 4)) : *(const f4 *)(vec + i * 4);
    float m[10] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1]};
    float v[4] = {vv[0], vv[1], vv[2], vv[3]}, x[4];
    sym_solve_closed<float, 4>(m, v, x);
    f4 r = {x[0], x[1], x[2], x[3]};
    if (NT) __builtin_nontemporal_store(r, (f4 *)(out + i * 4));
    else *(f4 *)(out + i * 4) = r;
}

// ---- tiled (library structure) with explicit knobs: TILE lanes, min waves/SIMD,
//      ELEMS elements per lane (ILP), nontemporal or not
template <int TILE, int WAVES, int ELEMS, bool NT>
__global__ __launch_bounds__(TILE, WAVES) void tiled_kernel(const float *__restrict__ mat,
                                                            const float *__restrict__ vec,
                                                            float *__restrict__ out, int64_t n)
{
    constexpr int RT = TILE * ELEMS;       // records per block
    constexpr int NVM = RT * 10 / 4;       // mat float4 per block
    constexpr int ITM = (NVM + TILE - 1) / TILE;
    __shared__ __align__(16) float smat[RT * 10];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * RT;
    if (r0 + RT > n) return;   // (microbench: n is a multiple of every RT used)
    const f4 *pm = (const f4 *)(mat + r0 * 10);
    f4 st[ITM];
#pragma unroll
    for (int k = 0; k < ITM; ++k) {
        const int q = t + k * TILE;
        if (NVM % TILE == 0 || q < NVM) st[k] = NT ? __builtin_nontemporal_load(pm + q) : pm[q];
    }
    f4 vv[ELEMS];
#pragma unroll
    for (int e = 0; e < ELEMS; ++e) {
        const f4 *pv = (const f4 *)(vec + (r0 + e * TILE + t) * 4);
        vv[e] = NT ? __builtin_nontemporal_load(pv) : *pv;
    }
#pragma unroll
    for (int k = 0; k < ITM; ++k) {
        const int q = t + k * TILE;
        if (NVM % TILE == 0 || q < NVM) ((f4 *)smat)[q] = st[k];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < ELEMS; ++e) {
        const f2 *ps = (const f2 *)(smat + (e * TILE + t) * 10);
        float m[10], v[4] = {vv[e][0], vv[e][1], vv[e][2], vv[e][3]}, x[4];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            f2 w = ps[s];
            m[2 * s] = w[0];
            m[2 * s + 1] = w[1];
        }
        sym_solve_closed<float, 4>(m, v, x);
        f4 r = {x[0], x[1], x[2], x[3]};
        f4 *po = (f4 *)(out + (r0 + e * TILE + t) * 4);
        if (NT) __builtin_nontemporal_store(r, po);
        else *po = r;
    }
}

// ---- persistent + software prefetch: a workgroup walks tiles with stride gridDim.x and
//      issues tile k+1's global loads before doing tile k's arithmetic (double-buffered LDS)
template <int TILE, int WAVES, bool NT>
__global__ __launch_bounds__(TILE, WAVES) void pipe_kernel(const float *__restrict__ mat,
                                                           const float *__restrict__ vec,
                                                           float *__restrict__ out, int64_t n)
{
    constexpr int NVM = TILE * 10 / 4;
    constexpr int ITM = (NVM + TILE - 1) / TILE;
    __shared__ __align__(16) float smat[2][TILE * 10];
    const int t = threadIdx.x;
    const int64_t ntiles = n / TILE;
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    f4 st[ITM], vv;
    auto issue = [&](int64_t tl) {
        const f4 *pm = (const f4 *)(mat + tl * TILE * 10);
#pragma unroll
        for (int k = 0; k < ITM; ++k) {
            const int q = t + k * TILE;
            if (NVM % TILE == 0 || q < NVM) st[k] = NT ? __builtin_nontemporal_load(pm + q) : pm[q];
        }
        const f4 *pv = (const f4 *)(vec + (tl * TILE + t) * 4);
        vv = NT ? __builtin_nontemporal_load(pv) : *pv;
    };
    issue(tile);
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
#pragma unroll
        for (int k = 0; k < ITM; ++k) {
            const int q = t + k * TILE;
            if (NVM % TILE == 0 || q < NVM) ((f4 *)smat[buf])[q] = st[k];
        }
        float v[4] = {vv[0], vv[1], vv[2], vv[3]};
        __syncthreads();
        const int64_t nxt = tile + gridDim.x;
        if (nxt < ntiles) issue(nxt);     // in flight during the arithmetic below
        const f2 *ps = (const f2 *)(smat[buf] + t * 10);
        float m[10], x[4];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            f2 w = ps[s];
            m[2 * s] = w[0];
            m[2 * s + 1] = w[1];
        }
        sym_solve_closed<float, 4>(m, v, x);
        f4 r = {x[0], x[1], x[2], x[3]};
        f4 *po = (f4 *)(out + (tile * TILE + t) * 4);
        if (NT) __builtin_nontemporal_store(r, po);
        else *po = r;
        // buffer `buf` is rewritten two iterations from now, after the next barrier
    }
}

// ---- batchinv 8x8 fp64 experiments: PASSES sub-tiles through a smaller LDS image
typedef double d2 __attribute__((ext_vector_type(2)));
template <int PASSES, int WAVES>
__global__ __launch_bounds__(64, WAVES) void binv_kernel(const double *__restrict__ a, double *__restrict__ out,
                                                         int64_t n)
{
    constexpr int RPP = 64 / PASSES;          // records per pass
    constexpr int IPP = 32 / PASSES;          // staged vectors (per lane) per pass
    __shared__ __align__(16) d2 lds[RPP * 33]; // 32 slots per record + 1 pad
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    if (r0 + 64 > n) return;
    const d2 *pa = (const d2 *)(a + r0 * 64);
    d2 st[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) st[k] = __builtin_nontemporal_load(pa + t + 64 * k);
    double m[8][8];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        if (p > 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < IPP; ++k) {
            const int row = 2 * k + t / 32, col = t % 32;
            lds[row * 33 + col] = st[p * IPP + k];
        }
        __syncthreads();
        if (t / RPP == p) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                d2 v = lds[(t % RPP) * 33 + j];
                m[j / 4][(j % 4) * 2] = v[0];
                m[j / 4][(j % 4) * 2 + 1] = v[1];
            }
        }
    }
    gj_inverse<double, 8>(m);
    d2 *po = (d2 *)(out + r0 * 64);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        __syncthreads();
        if (t / RPP == p) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                d2 v = {m[j / 4][(j % 4) * 2], m[j / 4][(j % 4) * 2 + 1]};
                lds[(t % RPP) * 33 + j] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < IPP; ++k) {
            const int row = 2 * k + t / 32, col = t % 32;
            __builtin_nontemporal_store(lds[row * 33 + col], po + t + 64 * (p * IPP + k));
        }
    }
}

struct Variant {
    std::string name;
    std::function<void()> run;
    std::vector<float> ms;
    double bytes = 0;   // algorithmic bytes per launch (0 = the 4x4 solve default)
    bool check = true;
};

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : 100000000LL; // multiple of 1024 expected
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    float *mat, *vec, *out, *ref;
    CK(hipMalloc(&mat, n * 10 * 4));
    CK(hipMalloc(&vec, n * 4 * 4));
    CK(hipMalloc(&out, n * 4 * 4));
    CK(hipMalloc(&ref, n * 4 * 4));
    hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, mat, vec, n);
    CK(hipDeviceSynchronize());
    int ncu = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, n = %lld\n", prop.name, ncu, (long long)n);

    nfm_operand om = {mat, 0, 10, 0, 1}, ov = {vec, 0, 4, 0, 1}, oo = {out, 0, 4, 0, 1}, orf = {ref, 0, 4, 0, 1};
    int rc = nfm_sym_solve(NFM_F32, 4, NFM_MAT_SYM, 1, n, &om, &ov, &orf, nullptr, nullptr);
    if (rc) { printf("lib rc %d\n", rc); return 1; }
    CK(hipDeviceSynchronize());

    std::vector<Variant> vs;
    auto add = [&](std::string name, std::function<void()> f) { vs.push_back({name, f, {}}); };
    add("lib (rec_kernel)", [&] { nfm_sym_solve(NFM_F32, 4, NFM_MAT_SYM, 1, n, &om, &ov, &oo, nullptr, nullptr); });
    add("copy ceiling nt", [&] { hipLaunchKernelGGL((copy_kernel<true>), dim3(n / 512), dim3(256), 0, 0, (const f4 *)mat, (const f4 *)vec, (f4 *)out, n); });
    add("copy ceiling plain", [&] { hipLaunchKernelGGL((copy_kernel<false>), dim3(n / 512), dim3(256), 0, 0, (const f4 *)mat, (const f4 *)vec, (f4 *)out, n); });
#define DIRECT(W, NT) add(std::string("direct w" #W) + (NT ? " nt" : " plain"), [&] { hipLaunchKernelGGL((direct_kernel<W, NT>), dim3((n + 255) / 256), dim3(256), 0, 0, mat, vec, out, n); })
    DIRECT(1, true); DIRECT(1, false); DIRECT(6, true); DIRECT(8, true); DIRECT(8, false);
#define TILED(T, W, E, NT) add(std::string("tiled T" #T " w" #W " e" #E) + (NT ? " nt" : " plain"), [&] { hipLaunchKernelGGL((tiled_kernel<T, W, E, NT>), dim3(n / (T * E)), dim3(T), 0, 0, mat, vec, out, n); })
    TILED(256, 1, 1, true); TILED(256, 1, 1, false); TILED(256, 6, 1, true); TILED(256, 8, 1, true);
    TILED(128, 1, 1, true); TILED(512, 1, 1, true); TILED(1024, 1, 1, true);
    TILED(256, 1, 2, true); TILED(256, 1, 4, true); TILED(128, 1, 2, true); TILED(64, 1, 4, true);
#define PIPE(T, W, NT, BPC) add(std::string("pipe T" #T " w" #W " x" #BPC) + (NT ? " nt" : " plain"), [&] { hipLaunchKernelGGL((pipe_kernel<T, W, NT>), dim3(ncu * BPC), dim3(T), 0, 0, mat, vec, out, n); })
    PIPE(256, 1, true, 4); PIPE(256, 1, true, 5); PIPE(256, 1, true, 8); PIPE(256, 6, true, 6); PIPE(256, 8, true, 8);
    PIPE(512, 1, true, 2); PIPE(512, 1, true, 4); PIPE(128, 1, true, 8); PIPE(128, 1, true, 16); PIPE(256, 1, false, 5);

    // ---- other library entry points (their own buffers), for per-config numbers
    const int64_t n6 = n, n8 = n / 10, n3 = n / 10;
    float *mat6, *vec6, *out6;
    double *a8, *o8, *m3, *o3;
    CK(hipMalloc(&mat6, n6 * 21 * 4)); CK(hipMalloc(&vec6, n6 * 6 * 4)); CK(hipMalloc(&out6, n6 * 6 * 4));
    CK(hipMalloc(&a8, n8 * 64 * 8)); CK(hipMalloc(&o8, n8 * 64 * 8));
    CK(hipMalloc(&m3, n3 * 6 * 8)); CK(hipMalloc(&o3, n3 * 6 * 8));
    hipLaunchKernelGGL(fill6_kernel, dim3((n6 + 255) / 256), dim3(256), 0, 0, mat6, vec6, n6);
    hipLaunchKernelGGL(fill8_kernel, dim3((n8 + 255) / 256), dim3(256), 0, 0, a8, m3, n8);
    CK(hipDeviceSynchronize());
    nfm_operand om6 = {mat6, 0, 21, 0, 1}, ov6 = {vec6, 0, 6, 0, 1}, oo6 = {out6, 0, 6, 0, 1};
    nfm_operand oa8 = {a8, 0, 64, 8, 1}, oo8 = {o8, 0, 64, 8, 1};
    nfm_operand om3 = {m3, 0, 6, 0, 1}, oo3 = {o3, 0, 6, 0, 1};
    auto addb = [&](std::string name, double bytes, std::function<void()> f) { vs.push_back({name, f, {}, bytes, false}); };
    addb("lib sym_solve 6x6 f32", n6 * 132.0, [&] { nfm_sym_solve(NFM_F32, 6, NFM_MAT_SYM, 1, n6, &om6, &ov6, &oo6, nullptr, nullptr); });
    addb("lib sym_matvec 4x4 f32", n * 72.0, [&] { nfm_sym_matvec(NFM_F32, 4, NFM_MAT_SYM, 0, 1, n, &om, &ov, nullptr, &oo, nullptr); });
    addb("lib sym_invert 4x4 f32", n * 80.0, [&] { nfm_operand oi = {mat6, 0, 10, 0, 1}; nfm_sym_invert(NFM_F32, 4, 0, 1, n, &om, &oi, nullptr); });
    addb("lib batchinv 8x8 f64", n8 * 1024.0, [&] { nfm_batch_inv(NFM_F64, 8, 0, 1, n8, &oa8, &oo8, nullptr); });
    addb("lib sym_invert 3x3 f64", n3 * 96.0, [&] { nfm_sym_invert(NFM_F64, 3, 0, 1, n3, &om3, &oo3, nullptr); });
#define BINV(P, W) addb("binv8 f64 passes" #P " w" #W, n8 * 1024.0, [&] { hipLaunchKernelGGL((binv_kernel<P, W>), dim3(n8 / 64), dim3(64), 0, 0, a8, o8, n8); })
    BINV(1, 1); BINV(1, 2); BINV(2, 1); BINV(2, 2); BINV(2, 3); BINV(4, 1); BINV(4, 2); BINV(4, 3);
    {
        const int64_t nvec = n8 * 512 / 16;   // the 5.12 GB batchinv input, copied to its output buffer
        addb("copy 50/50 ceiling nt", 2.0 * nvec * 16, [&, nvec] { hipLaunchKernelGGL((copy5050_kernel<true>), dim3(nvec / 1024), dim3(256), 0, 0, (const f4 *)a8, (f4 *)o8, nvec + 1); });
        addb("copy 50/50 ceiling plain", 2.0 * nvec * 16, [&, nvec] { hipLaunchKernelGGL((copy5050_kernel<false>), dim3(nvec / 1024), dim3(256), 0, 0, (const f4 *)a8, (f4 *)o8, nvec + 1); });
    }
    addb("hipMemcpy D2D 4 GB (2x bytes)", 2.0 * n * 40, [&] { hipMemcpyAsync(mat6, mat, n * 40, hipMemcpyDeviceToDevice, 0); });

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // correctness of every variant vs the library result (bitwise), then interleaved timing
    std::vector<float> h_ref(4096 * 4), h_out(4096 * 4);
    for (auto &v : vs) {
        if (!v.check) { v.run(); CK(hipDeviceSynchronize()); continue; }
        CK(hipMemset(out, 0, n * 16));
        v.run();
        CK(hipDeviceSynchronize());
        bool ok = true;
        for (int64_t off : {(int64_t)0, n / 2 - 1024, n - 4096}) {
            CK(hipMemcpy(h_ref.data(), ref + off * 4, 4096 * 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h_out.data(), out + off * 4, 4096 * 16, hipMemcpyDeviceToHost));
            ok = ok && memcmp(h_ref.data(), h_out.data(), 4096 * 16) == 0;
        }
        if (!ok && v.name.find("copy") == std::string::npos) printf("MISMATCH in %s\n", v.name.c_str());
    }
    for (int r = 0; r < rounds; ++r)
        for (auto &v : vs) {
            CK(hipEventRecord(e0, 0));
            v.run();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms);
        }
    printf("%-32s %9s %9s %9s %8s\n", "variant", "med ms", "min ms", "GB/s med", "frac8TB");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        double gbs = (v.bytes > 0 ? v.bytes : n * 72.0) / (med * 1e-3) / 1e9;
        printf("%-32s %9.4f %9.4f %9.1f %8.3f\n", v.name.c_str(), med, mn, gbs, gbs / 8000.0);
    }
    return 0;
}

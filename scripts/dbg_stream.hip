// debug: which ingredient breaks the f32 streaming inverse at N >= 13?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../nitorch_fastmath_amd/csrc/nfm_batched_ops.hpp"
using namespace nfm;

// opaque select: one v_cndmask per 32-bit half under a wave mask, invisible to LLVM's
// select <-> branch transformations
__device__ __forceinline__ float asel(unsigned long long m, float a, float b)
{
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ int asel(unsigned long long m, int a, int b)
{
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ double asel(unsigned long long m, double a, double b)
{
    union { double d; int i[2]; } ua, ub, ur;
    ua.d = a; ub.d = b;
    ur.i[0] = asel(m, ua.i[0], ub.i[0]);
    ur.i[1] = asel(m, ua.i[1], ub.i[1]);
    return ur.d;
}

template <typename T, int N, bool SB>
__device__ __forceinline__ void lu_f(T (&a)[N][N], T (&rowid)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) rowid[i] = T(i);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int p = k;
        T best = fabs_(a[k][k]);
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const T x = fabs_(a[i][k]);
            const unsigned long long g = __ballot(x > best);
            best = asel(g, x, best);
            p = asel(g, i, p);
        }
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const unsigned long long s = __ballot(p == i);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const T t = a[k][j];
                a[k][j] = asel(s, a[i][j], t);
                a[i][j] = asel(s, t, a[i][j]);
            }
            const T ti = rowid[k];
            rowid[k] = asel(s, rowid[i], ti);
            rowid[i] = asel(s, ti, rowid[i]);
        }
        const T rp = T(1) / a[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const T l = a[i][k] * rp;
            a[i][k] = l;
#pragma unroll
            for (int j = k + 1; j < N; ++j) a[i][j] -= l * a[k][j];
        }
        if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
    }
}
template <typename T, int N, bool SB>
__device__ __forceinline__ void lu_s(const T (&lu)[N][N], const T (&rowid)[N], int c, T (&x)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = asel(__ballot(rowid[i] == T(c)), T(1), T(0));
#pragma unroll
    for (int i = 1; i < N; ++i) {
        T s = x[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= lu[i][j] * x[j];
        x[i] = s;
        if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        T s = x[i];
#pragma unroll
        for (int j = i + 1; j < N; ++j) s -= lu[i][j] * x[j];
        x[i] = s / lu[i][i];
        if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
    }
}

// VAR bit0: sched barriers; bit1: column loop unrolled; bit2: write to global directly instead of LDS
template <typename T, int N, int VAR, int W = 1>
__global__ __launch_bounds__(64, W) void k(const T *__restrict__ in, T *__restrict__ out, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    T a[N][N];
    T rowid[N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) a[r][c] = in[(size_t)i * N * N + r * N + c];
    lu_f<T, N, (VAR & 1) != 0>(a, rowid);
    T *o = out + (size_t)i * N * N;
    if constexpr (VAR & 2) {
#pragma unroll
        for (int c = 0; c < N; ++c) {
            T x[N];
            lu_s<T, N, (VAR & 1) != 0>(a, rowid, c, x);
#pragma unroll
            for (int r = 0; r < N; ++r) o[r * N + c] = x[r];
        }
    } else {
#pragma unroll 1
        for (int c = 0; c < N; ++c) {
            T x[N];
            lu_s<T, N, (VAR & 1) != 0>(a, rowid, c, x);
#pragma unroll
            for (int r = 0; r < N; ++r) o[r * N + c] = x[r];
        }
    }
}

template <typename T, int N, int VAR, int W = 1>
void run(const char *name)
{
    const int n = 256;
    std::vector<T> h(n * N * N), r(n * N * N);
    unsigned s = 12345;
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                s = s * 1664525u + 1013904223u;
                h[(b * N + i) * N + j] = (T)(((s >> 8) & 0xffff) / 65536.0 - 0.5) + (i == j ? (T)4 : (T)0);
            }
    T *din, *dout;
    hipMalloc(&din, h.size() * sizeof(T));
    hipMalloc(&dout, h.size() * sizeof(T));
    hipMemcpy(din, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<T, N, VAR, W>), dim3((n + 63) / 64), dim3(64), 0, 0, din, dout, n);
    hipMemcpy(r.data(), dout, h.size() * sizeof(T), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double acc = 0;
                for (int q = 0; q < N; ++q) acc += (double)h[(b * N + i) * N + q] * (double)r[(b * N + q) * N + j];
                double e = std::fabs(acc - (i == j));
                if (!(e <= worst)) worst = e;
            }
    printf("%-40s max |A inv - I| = %.3e\n", name, worst);
    hipFree(din);
    hipFree(dout);
}

int main()
{
    run<float, 12, 1>("f32 N=12 sb=1 loop");
    run<float, 14, 0>("f32 N=14 sb=0 loop");
    run<float, 14, 1>("f32 N=14 sb=1 loop");
    run<float, 14, 2>("f32 N=14 sb=0 unrolled");
    run<float, 14, 3>("f32 N=14 sb=1 unrolled");
    run<float, 16, 0>("f32 N=16 sb=0 loop");
    run<float, 16, 1>("f32 N=16 sb=1 loop");
    run<double, 14, 1>("f64 N=14 sb=1 loop");
    run<float, 14, 1, 2>("f32 N=14 sb=1 loop, <=256 VGPR (no AGPR)");
    run<float, 16, 1, 2>("f32 N=16 sb=1 loop, <=256 VGPR (no AGPR)");
    run<float, 14, 0, 2>("f32 N=14 sb=0 loop, <=256 VGPR (no AGPR)");
    return 0;
}

// Streaming-bandwidth probe for MI355X: which copy shapes reach what fraction of HBM, and where the fir kernels sit.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe -Lmaua_stylegan2_amd/csrc -lmaua_hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../include/maua_hip.h"

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_store(float4 v, float4* p) { f4v w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(p)); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void copy_exact(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 v = NT ? nt_load(x + i) : x[i];
        if (NT) nt_store(v, y + i); else y[i] = v;
    }
}
template <bool NT, int U>
__global__ __launch_bounds__(256) void copy_chunk(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {
    // each workgroup owns a contiguous chunk of U*256 quads; U loads in flight per lane
    for (int64_t base = (int64_t)blockIdx.x * (256 * U); base < n4; base += (int64_t)gridDim.x * (256 * U)) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { int64_t i = base + u * 256 + threadIdx.x; v[u] = i < n4 ? (NT ? nt_load(x + i) : x[i]) : make_float4(0,0,0,0); }
#pragma unroll
        for (int u = 0; u < U; ++u) { int64_t i = base + u * 256 + threadIdx.x; if (i < n4) { if (NT) nt_store(v[u], y + i); else y[i] = v[u]; } }
    }
}
__global__ __launch_bounds__(256) void read_only(const float4* __restrict__ x, float* __restrict__ out, int64_t n4) {
    float s = 0.f;
    for (int64_t base = (int64_t)blockIdx.x * 1024; base < n4; base += (int64_t)gridDim.x * 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { int64_t i = base + u * 256 + threadIdx.x; v[u] = i < n4 ? x[i] : make_float4(0,0,0,0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void write_only(float4* __restrict__ y, int64_t n4) {
    for (int64_t base = (int64_t)blockIdx.x * 1024; base < n4; base += (int64_t)gridDim.x * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { int64_t i = base + u * 256 + threadIdx.x; if (i < n4) y[i] = make_float4(1.f, 2.f, 3.f, 4.f); }
    }
}
template <int U>
__global__ __launch_bounds__(256) void copy_dword_chunk(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t base = (int64_t)blockIdx.x * (256 * U); base < n; base += (int64_t)gridDim.x * (256 * U)) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { int64_t i = base + u * 256 + threadIdx.x; v[u] = i < n ? x[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < U; ++u) { int64_t i = base + u * 256 + threadIdx.x; if (i < n) y[i] = v[u]; }
    }
}

template <class F>
float time_ms(F f, hipStream_t st, int iters = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int64_t n = 1ll << 28, n4 = n / 4;  // 1 GiB
    float *x, *y; CK(hipMalloc(&x, n * 4 + 64)); CK(hipMalloc(&y, n * 4 + 64));
    CK(hipMemsetAsync(x, 1, n * 4, st));
    auto rep = [&](const char* name, float ms, double bytes) { printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6); };
    const double rw = 8.0 * n;
    rep("copy exact grid", time_ms([&] { copy_exact<false><<<(unsigned)(n4 / 256), 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
    rep("copy exact grid nt", time_ms([&] { copy_exact<true><<<(unsigned)(n4 / 256), 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "copy chunk U=1 grid %d", g); rep(nm, time_ms([&] { copy_chunk<false, 1><<<g, 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
        snprintf(nm, 64, "copy chunk U=4 grid %d", g); rep(nm, time_ms([&] { copy_chunk<false, 4><<<g, 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
        snprintf(nm, 64, "copy chunk U=8 grid %d", g); rep(nm, time_ms([&] { copy_chunk<false, 8><<<g, 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
        snprintf(nm, 64, "copy chunk U=4 nt grid %d", g); rep(nm, time_ms([&] { copy_chunk<true, 4><<<g, 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
    }
    rep("copy chunk U=4 exact (n4/1024 blocks)", time_ms([&] { copy_chunk<false, 4><<<(unsigned)(n4 / 1024), 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
    rep("copy chunk U=8 exact (n4/2048 blocks)", time_ms([&] { copy_chunk<false, 8><<<(unsigned)(n4 / 2048), 256, 0, st>>>((float4*)x, (float4*)y, n4); }, st), rw);
    rep("read only U=4 grid 4096", time_ms([&] { read_only<<<4096, 256, 0, st>>>((float4*)x, y, n4); }, st), 4.0 * n);
    rep("write only U=4 grid 4096", time_ms([&] { write_only<<<4096, 256, 0, st>>>((float4*)y, n4); }, st), 4.0 * n);
    rep("copy dword U=8 grid 4096", time_ms([&] { copy_dword_chunk<8><<<4096, 256, 0, st>>>(x, y, n); }, st), rw);
    rep("copy dword U=16 grid 8192", time_ms([&] { copy_dword_chunk<16><<<8192, 256, 0, st>>>(x, y, n); }, st), rw);
    rep("hipMemcpyDtoD", time_ms([&] { hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, st); }, st), rw);

    // fir kernels through the C ABI: B=8, C=32, 1025^2 -> 1024^2
    const int planes = 256, r = 1024;
    float *fx, *fy, *fk; CK(hipMalloc(&fx, (size_t)planes * (r + 1) * (r + 1) * 4 + 64)); CK(hipMalloc(&fy, (size_t)planes * r * r * 4 + 64));
    CK(hipMalloc(&fk, 64));
    float hk[16]; for (int i = 0; i < 16; ++i) hk[i] = 0.0625f * (1 + (i % 4 == 1 || i % 4 == 2) * 2) * (1 + (i / 4 == 1 || i / 4 == 2) * 2) / 4.f;
    CK(hipMemcpy(fk, hk, 64, hipMemcpyHostToDevice));
    CK(hipMemsetAsync(fx, 0, (size_t)planes * (r + 1) * (r + 1) * 4, st));
    const double fb = 4.0 * planes * ((double)(r + 1) * (r + 1) + (double)r * r);
    for (int path : {1, 2, 6, 7, 9, 13, 21, 0}) {
        maua_tuning_set(0, path);
        char nm[64]; snprintf(nm, 64, "fir path %d (1 tile, 2 vec4, 6+n strips of n+1... , 0 auto)", path);
        rep(nm, time_ms([&] { maua_upfirdn2d_f32(fx, fk, fy, planes, r + 1, r + 1, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, st); }, st), fb);
    }
    float *nzb, *nwb, *bsb; CK(hipMalloc(&nzb, (size_t)r * r * 4)); CK(hipMalloc(&nwb, 4)); CK(hipMalloc(&bsb, 32 * 4));
    CK(hipMemsetAsync(nzb, 0, (size_t)r * r * 4, st)); CK(hipMemsetAsync(nwb, 0, 4, st)); CK(hipMemsetAsync(bsb, 0, 128, st));
    for (int path : {1, 7, 9, 0}) {
        maua_tuning_set(0, path);
        char nm[64]; snprintf(nm, 64, "fir+noise+act tail path %d", path);
        rep(nm, time_ms([&] { maua_blur_noise_act_f32(fx, fk, fy, 8, 32, r + 1, r + 1, 4, 4, 1, 1, nullptr, nzb, 0, nwb, bsb, nullptr, 0, nullptr, 0, st); }, st), fb);
    }
    maua_tuning_set(0, 0);
    rep("fir vec4 (aligned ptrs)", time_ms([&] { maua_upfirdn2d_f32(fx, fk, fy, planes, r + 1, r + 1, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, st); }, st), fb);
    rep("fir dword tile (misaligned ptr)", time_ms([&] { maua_upfirdn2d_f32(fx + 1, fk, fy, planes, r + 1, r + 1, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, st); }, st), fb);
    CK(hipStreamSynchronize(st));
    return 0;
}

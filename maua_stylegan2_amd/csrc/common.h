// Shared helpers for libmaua_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/maua_hip.h"

#define MAUA_LAUNCH_CHECK()                       \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that uses more than the default 64 KB of LDS.  The attribute belongs to
// (function, device): SUCCESS is remembered per device in the caller's `done` bit mask (one static per launcher), a failure is
// returned as the launch's hipError_t (> 0, the contract of include/maua_hip.h) and retried by the next call instead of being cached
// for the life of the process.  Relaxed atomics: two threads racing on the first call both set the attribute, which is harmless.
static inline int maua_allow_full_lds(const void* kern, unsigned long long* done, int bytes = 160 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done, __ATOMIC_RELAXED) & bit) return 0;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
    return 0;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bijective XCD-aware remap (cdna_hip_programming.md §5 "XCD swizzle must be bijective"): block b runs on
// XCD b % 8; give every XCD one contiguous chunk of the logical tile range so that neighbouring tiles
// (which share halos / weight panels) hit the same 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int NX = 8;
    int q = nblocks / NX, r = nblocks % NX;
    int xcd = bid % NX, idx = bid / NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float lrelu_gain(float v) { return (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f; }

// modconv_w2d.hip (mode 5 of maua_modconv3x3_f32 / maua_styledconv_torgb_f32): 2-D Winograd F(2x4, 3x3) plain convolution
int maua_w2d_tiles(int cin, int cout, int h, int w, int* tm, int* tn);
const char* maua_w2d_last_instance();
int maua_w2d_launch(const float* x, const float* wq, const float* s, int s_stride, const float* d, float* y, int batch, int cin,
                    int cout, int h, int w, float wscale, int fuse_act, const float* noise, int64_t noise_batch_stride,
                    const float* noise_w, const float* bias, const float* rgb_w, const float* rgb_s, float rgb_wscale,
                    const float* rgb_bias, const float* rgb_skip, const float* rgb_k4, float* rgb_out, uint8_t* rgb_u8,
                    int rgb_mode, const maua_frame_source_t* src, int noise_slot, const float* post_s, void* stream);

// modconv_up2d.hip (mode 6 of maua_modconv3x3_f32): transposed convolution with F(2,2) on both axes of its polyphase form
const char* maua_up2d_last_instance();
int64_t maua_up2d_ws_floats(int batch, int cin, int h);
int maua_up2d_launch(const float* x, const float* wq, const float* s, int s_stride, const float* d, float* y, float* ws, int batch, int cin,
                     int cout, int h, int w, float wscale, void* stream);

// ... its form for 16-wide inputs with K split over several workgroups per tile (partial raw maps in slabs; maua_upconv_blur_lowres_f32, up = 6)
int maua_up2d16_ok(int cin, int cout, int h, int w);
int maua_up2d16_splits(int batch, int cin, int cout, int h, int w);
int maua_up2d16_launch(const float* x, const float* wq, const float* s, int s_stride, float* y, float* xcol, int batch, int cin, int cout, int h,
                       int w, float wscale, int* splits_out, void* stream);

// modconv_sbf16.hip (mode 7 of maua_modconv3x3_f32; side measurement, off by default): plain 3x3 convolution with split-bf16 products
int maua_sbf16_ok(int cin, int cout, int h, int w);
const char* maua_sbf16_last_instance();
int maua_sbf16_launch(const float* x, const void* wq, const float* s, int s_stride, const float* d, float* y, float* ws, int batch, int cin,
                      int cout, int h, int w, int up, float wscale, int fuse_act, const float* noise, int64_t noise_batch_stride,
                      const float* noise_w, const float* bias, const maua_frame_source_t* src, int noise_slot, void* stream);
int maua_up2d_edge_launch(const float* x, const float* edge_taps, const float* s, int s_stride, const float* d, float* y, const float* xcol,
                          int batch, int cin, int cout, int h, int w, float wscale, void* stream);

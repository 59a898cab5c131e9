// Runtime glue of libmaua_hip.so: ABI version, device query, hipGraph capture/replay, HIP-event timing and the
// uint8 frame epilogue (render.py:40-43).
#include "common.h"

#include <stddef.h>
#include <string.h>

extern "C" int maua_abi_version(void) { return 5; }

// src->frame0 = frame0 on `stream`: the only per-replay input traffic of a captured forward (include/maua_hip.h, frame source)
extern "C" int maua_frame_source_seek(maua_frame_source_t* src, int frame0, void* stream) {
    if (!src || frame0 < 0) return MAUA_EINVAL;
    static_assert(offsetof(maua_frame_source_t, frame0) == 0, "frame0 is the first word of the struct");
    return (int)hipMemsetD32Async((hipDeviceptr_t)src, frame0, 1, (hipStream_t)stream);
}

extern "C" int maua_device_info(int* cu_count, int* lds_bytes, char* name, int name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return 0;
}

#ifdef MAUA_EXPERIMENTS
// Ablation / A-B switches of tools/ (NOT part of the product ABI: include/maua_hip.h does not declare it and the default build does
// not export it).  key 1: conv ablation mask, 2: conv tile-shape switches (the 2-D Winograd kernel has compile-time masks only).
int maua_conv_debug_set(int v);
int maua_conv_cfg_set(int v);
extern "C" int maua_tuning_set(int key, int value) {
    if (key == 1) return maua_conv_debug_set(value);
    if (key == 2) return maua_conv_cfg_set(value);
    return MAUA_EINVAL;
}
#endif

// ------------------------------------------------------------------------------------------------ hipGraph
extern "C" int maua_graph_begin_capture(void* stream) {
    return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
}

extern "C" int maua_graph_end_capture(void* stream, void** graph_exec_out) {
    if (!graph_exec_out) return MAUA_EINVAL;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return (int)e;
    *graph_exec_out = (void*)exec;
    return 0;
}

extern "C" int maua_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return MAUA_EINVAL;
    return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
}

extern "C" int maua_graph_destroy(void* graph_exec) {
    if (!graph_exec) return 0;
    return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

// ------------------------------------------------------------------------------------------------ events
extern "C" int maua_event_create(void** ev) {
    if (!ev) return MAUA_EINVAL;
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return (int)r;
    *ev = (void*)e;
    return 0;
}
extern "C" int maua_event_record(void* ev, void* stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
extern "C" int maua_event_elapsed_ms(void* a, void* b, float* ms) {
    hipError_t r = hipEventSynchronize((hipEvent_t)b);
    if (r != hipSuccess) return (int)r;
    return (int)hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b);
}
extern "C" int maua_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }

// ------------------------------------------------------------------------------------------------ frame epilogue
namespace {
// [B,3,H,W] fp32 -> [B,H,W,3] uint8.  Each thread converts 4 consecutive pixels of one row: three 16-byte loads
// (one per colour plane) and one 12-byte (3 x dword) store — 1 read + 0.25 write bytes per input byte, HBM-bound.
__global__ __launch_bounds__(256) void frames_to_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out,
                                                           int64_t quads_per_frame, int64_t plane, int64_t total) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t b = q / quads_per_frame;
        const int64_t p = (q - b * quads_per_frame) * 4;
        const float* base = img + b * 3 * plane + p;
        const float4 r = *reinterpret_cast<const float4*>(base);
        const float4 g = *reinterpret_cast<const float4*>(base + plane);
        const float4 bl = *reinterpret_cast<const float4*>(base + 2 * plane);
        auto cv = [](float v) -> uint32_t {
            v = fminf(fmaxf(v, -1.f), 1.f);
            return (uint32_t)((v + 1.f) * 127.5f);  // truncating cast, as numpy astype(uint8) on [0,255]
        };
        const uint32_t w0 = cv(r.x) | (cv(g.x) << 8) | (cv(bl.x) << 16) | (cv(r.y) << 24);
        const uint32_t w1 = cv(g.y) | (cv(bl.y) << 8) | (cv(r.z) << 16) | (cv(g.z) << 24);
        const uint32_t w2 = cv(bl.z) | (cv(r.w) << 8) | (cv(g.w) << 16) | (cv(bl.w) << 24);
        uint32_t* o = reinterpret_cast<uint32_t*>(out + (b * plane + p) * 3);
        o[0] = w0;
        o[1] = w1;
        o[2] = w2;
    }
}

__global__ __launch_bounds__(256) void frames_to_u8_scalar_kernel(const float* __restrict__ img, uint8_t* __restrict__ out,
                                                                  int64_t plane, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % 3;
        const int64_t p = (i / 3) % plane;
        const int64_t b = i / (3 * plane);
        float v = img[(b * 3 + c) * plane + p];
        v = fminf(fmaxf(v, -1.f), 1.f);
        out[i] = (uint8_t)((v + 1.f) * 127.5f);
    }
}
}  // namespace

namespace {
// Crop + bilinear resize of uint8 NHWC frames, the wide-output delivery of render.py:97-104 (2048-px frames: crop 112 px from both
// ends of the long side, PIL resize(BILINEAR) to 1920x1080 / 1080x1920).  Pillow's 8-bit resampling (ImagingResample) is restated
// exactly: for an UP-scale the filter is the 2-tap triangle at source position (dst + 0.5) * in / out - 0.5 (a rational number:
// evaluated in integers), taps clamped to the image, coefficients rounded to 22 fractional bits, a horizontal pass whose result is
// rounded (half up) to uint8, then the vertical pass on that — bit-equal to PIL (tests/test_host_logic.py holds the same formula
// in numpy against PIL itself).  One thread per output pixel; 12 bytes read per 3 written at most: HBM-bound and tiny.
constexpr int RESIZE_BITS = 22;
struct ResizeTap {
    int a, b;    // source indices of the two taps (clamped)
    int k0, k1;  // fixed-point weights
};
__device__ __forceinline__ ResizeTap resize_tap(int o, int n_in, int n_out) {
    const long long num = (2ll * o + 1) * n_in - n_out, den = 2ll * n_out;
    long long i0 = num / den;
    if (num < 0 && i0 * den != num) --i0;  // floor division
    const double f = (double)(num - i0 * den) / (double)den;
    ResizeTap t;
    t.k1 = (int)floor(0.5 + f * (double)(1 << RESIZE_BITS));
    t.k0 = (int)floor(0.5 + (1.0 - f) * (double)(1 << RESIZE_BITS));
    t.a = min(max((int)i0, 0), n_in - 1);
    t.b = min(max((int)i0 + 1, 0), n_in - 1);
    return t;
}
__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int in_h, int in_w,
                                                             int x0, int y0, int cw, int ch, int out_w, int out_h, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ox = (int)(idx % out_w);
        const int oy = (int)((idx / out_w) % out_h);
        const int64_t b = idx / ((int64_t)out_w * out_h);
        const ResizeTap tx = resize_tap(ox, cw, out_w), ty = resize_tap(oy, ch, out_h);
        const uint8_t* ra = in + ((b * in_h + y0 + ty.a) * in_w + x0) * 3;
        const uint8_t* rb = in + ((b * in_h + y0 + ty.b) * in_w + x0) * 3;
        uint8_t* o = out + idx * 3;
        constexpr int HALF = 1 << (RESIZE_BITS - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // horizontal pass on the two source rows, rounded to 8 bits as PIL's intermediate image is; then the vertical pass
            const int ha = min(max((tx.k0 * (int)ra[tx.a * 3 + c] + tx.k1 * (int)ra[tx.b * 3 + c] + HALF) >> RESIZE_BITS, 0), 255);
            const int hb = min(max((tx.k0 * (int)rb[tx.a * 3 + c] + tx.k1 * (int)rb[tx.b * 3 + c] + HALF) >> RESIZE_BITS, 0), 255);
            o[c] = (uint8_t)min(max((ty.k0 * ha + ty.k1 * hb + HALF) >> RESIZE_BITS, 0), 255);
        }
    }
}
}  // namespace

extern "C" int maua_crop_resize_u8(const uint8_t* in, uint8_t* out, int batch, int in_h, int in_w, int x0, int y0, int crop_w, int crop_h,
                                   int out_w, int out_h, void* stream) {
    if (!in || !out || batch <= 0 || in_h <= 0 || in_w <= 0 || crop_w <= 0 || crop_h <= 0 || out_w <= 0 || out_h <= 0) return MAUA_EINVAL;
    if (x0 < 0 || y0 < 0 || x0 + crop_w > in_w || y0 + crop_h > in_h) return MAUA_EINVAL;
    if (out_w < crop_w || out_h < crop_h) return MAUA_ENOSYS;  // a down-scale widens PIL's filter support: not the 2-tap form
    const int64_t total = (int64_t)batch * out_w * out_h;
    const int64_t blocks = ceil_div64(total, 256);
    hipLaunchKernelGGL(crop_resize_u8_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, in, out,
                       in_h, in_w, x0, y0, crop_w, crop_h, out_w, out_h, total);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_frames_to_u8(const float* img, uint8_t* out, int batch, int h, int w, void* stream) {
    if (!img || !out || batch <= 0 || h <= 0 || w <= 0) return MAUA_EINVAL;
    const int64_t plane = (int64_t)h * w;
    hipStream_t st = (hipStream_t)stream;
    if (plane % 4 == 0 && (((uintptr_t)img) & 15) == 0 && (((uintptr_t)out) & 3) == 0) {
        const int64_t total = (int64_t)batch * plane / 4;
        const int64_t blocks = ceil_div64(total, 256);
        hipLaunchKernelGGL(frames_to_u8_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, img,
                           out, plane / 4, plane, total);
    } else {
        const int64_t total = (int64_t)batch * plane * 3;
        const int64_t blocks = ceil_div64(total, 256);
        hipLaunchKernelGGL(frames_to_u8_scalar_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st,
                           img, out, plane, total);
    }
    MAUA_LAUNCH_CHECK();
    return 0;
}

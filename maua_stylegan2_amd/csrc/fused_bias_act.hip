// fused bias + activation for gfx950.
//
// Behavioural contract: /root/reference/op/fused_bias_act_kernel.cu:18-98 (act*10+grad switch, bias index
// (i / step_b) % size_b, out = y * scale) as bound by op/fused_act.py:56 (forward: act 3, grad 0).
// Pure HBM streaming: 16 bytes per lane per access whenever a float4 never straddles a bias boundary
// (step_b % 4 == 0, true for every feature map of the generator), scalar otherwise; grid-stride over
// <= 8 workgroups per CU.
#include "common.h"

#include <hip/hip_fp16.h>

namespace {

__device__ __forceinline__ float act_apply(float v, float r, int code, float alpha) {
    switch (code) {
        case 12:
        case 32:
            return 0.f;
        case 30:
            return v > 0.f ? v : v * alpha;
        case 31:
            return r > 0.f ? v : v * alpha;
        default:  // 10, 11 and anything else: identity (reference `default:` label)
            return v;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(const float* x, const float* __restrict__ b,
                                                       const float* __restrict__ ref, float* y,
                                                       int64_t n, int size_b, int step_b, int code, float alpha,
                                                       float scale) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (VEC) {
        const int64_t n4 = n >> 2;
        const int step4 = step_b >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(x)[i];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ref) r = reinterpret_cast<const float4*>(ref)[i];
            float bb = 0.f;
            if (b) bb = b[(i / step4) % size_b];
            v.x = act_apply(v.x + bb, r.x, code, alpha) * scale;
            v.y = act_apply(v.y + bb, r.y, code, alpha) * scale;
            v.z = act_apply(v.z + bb, r.z, code, alpha) * scale;
            v.w = act_apply(v.w + bb, r.w, code, alpha) * scale;
            reinterpret_cast<float4*>(y)[i] = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            float v = x[i];
            if (b) v += b[(i / step_b) % size_b];
            const float r = ref ? ref[i] : 0.f;
            y[i] = act_apply(v, r, code, alpha) * scale;
        }
    }
}

// half / double instantiations of the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF (op/fused_bias_act_kernel.cu:79):
// arithmetic in the tensor's own type for double, in fp32 for half (the reference's scalar_t arithmetic on __half also
// rounds through float on the device); scalar path only — these dtypes are off the fp32 hot path.
template <typename T, typename A>
__global__ __launch_bounds__(256) void bias_act_typed_kernel(const T* x, const T* __restrict__ b, const T* __restrict__ ref, T* y,
                                                             int64_t n, int size_b, int step_b, int code, A alpha, A scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        A v = (A)x[i];
        if (b) v += (A)b[(i / step_b) % size_b];
        const A r = ref ? (A)ref[i] : (A)0;
        A out;
        switch (code) {
            case 12:
            case 32: out = (A)0; break;
            case 30: out = v > (A)0 ? v : v * alpha; break;
            case 31: out = r > (A)0 ? v : v * alpha; break;
            default: out = v; break;
        }
        y[i] = (T)(out * scale);
    }
}

template <typename T, typename A>
int launch_bias_act_typed(const void* x, const void* b, const void* ref, void* y, int64_t size_x, int size_b, int step_b, int act,
                          int grad, float alpha, float scale, void* stream) {
    if (size_x < 0 || (size_x > 0 && (!x || !y)) || step_b <= 0) return MAUA_EINVAL;
    if (size_x == 0) return 0;
    if (size_b <= 0) b = nullptr;
    if (!b) size_b = 1;
    const int64_t blocks = ceil_div64(size_x, 256);
    hipLaunchKernelGGL((bias_act_typed_kernel<T, A>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)b, (const T*)ref, (T*)y, size_x, size_b, step_b, act * 10 + grad, (A)alpha, (A)scale);
    MAUA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int maua_fused_bias_act_f16(const void* x, const void* b, const void* ref, void* y, int64_t size_x, int size_b, int step_b,
                                       int act, int grad, float alpha, float scale, void* stream) {
    return launch_bias_act_typed<__half, float>(x, b, ref, y, size_x, size_b, step_b, act, grad, alpha, scale, stream);
}

extern "C" int maua_fused_bias_act_f64(const void* x, const void* b, const void* ref, void* y, int64_t size_x, int size_b, int step_b,
                                       int act, int grad, float alpha, float scale, void* stream) {
    return launch_bias_act_typed<double, double>(x, b, ref, y, size_x, size_b, step_b, act, grad, alpha, scale, stream);
}

extern "C" int maua_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y, int64_t size_x,
                                       int size_b, int step_b, int act, int grad, float alpha, float scale,
                                       void* stream) {
    if (size_x < 0 || (size_x > 0 && (!x || !y)) || step_b <= 0) return MAUA_EINVAL;
    if (size_x == 0) return 0;
    if (size_b <= 0) b = nullptr;
    if (!b) size_b = 1;
    const int code = act * 10 + grad;
    const bool aligned = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)ref) & 15) == 0;
    const bool vec = aligned && (size_x % 4 == 0) && (step_b % 4 == 0);
    const int64_t work = vec ? size_x / 4 : size_x;
    const int64_t blocks = ceil_div64(work, 256);
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(bias_act_kernel<true>, dim3(grid), dim3(256), 0, st, x, b, ref, y, size_x, size_b, step_b,
                           code, alpha, scale);
    else
        hipLaunchKernelGGL(bias_act_kernel<false>, dim3(grid), dim3(256), 0, st, x, b, ref, y, size_x, size_b, step_b,
                           code, alpha, scale);
    MAUA_LAUNCH_CHECK();
    return 0;
}

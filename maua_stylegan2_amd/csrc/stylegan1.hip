// StyleGAN1 layer epilogue (G_style, `--stylegan1`): one kernel for what /root/reference/models/stylegan1.py:258-318
// (LayerEpilogue) runs as five modules — bias add (MyConv2d :101-102 / InputBlock :354), NoiseLayer (:106-123: per-channel
// weight x per-pixel noise), LeakyReLU(0.2), nn.InstanceNorm2d (biased variance, eps 1e-5, no affine) and StyleMod
// (:126-136: x * (style[:, 0] + 1) + style[:, 1]):
//
//     v  = lrelu_0.2( x[b,c] + bias[c] + noise_w[c] * noise[b or 0, 0] )
//     y  = (v - mean_hw(v)) * rsqrt(var_hw(v) + 1e-5) * (s[b, c] + 1) + s[b, C + c]
//
// One workgroup per (b, c) plane: pass 1 forms v and its mean, pass 2 the centred second moment (two-pass variance: no
// cancellation on 1024^2 planes), pass 3 writes; passes 2 and 3 re-read the plane through L2.  HBM-bound: 4 B read + 4 B
// written per element (+ the shared noise plane).
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sg1_epilogue_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                           const float* __restrict__ noise, int64_t noise_batch_stride,
                                                           const float* __restrict__ noise_w, const float* __restrict__ style,
                                                           int style_stride, float* __restrict__ y, int channels, int64_t plane,
                                                           int instance_norm) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* xp = x + ((size_t)b * channels + c) * plane;
    float* yp = y + ((size_t)b * channels + c) * plane;
    const float bv = bias ? bias[c] : 0.f;
    const float nw = noise ? noise_w[c] : 0.f;
    const float* np_ = noise ? noise + (size_t)b * noise_batch_stride : nullptr;
    auto value = [&](int64_t i) {
        float v = xp[i] + bv;
        if (np_) v = fmaf(nw, np_[i], v);
        return v > 0.f ? v : 0.2f * v;
    };
    float mean = 0.f, inv_std = 1.f;
    if (instance_norm) {
        float s1 = 0.f;
        for (int64_t i = threadIdx.x; i < plane; i += 256) s1 += value(i);
        mean = block_sum(s1, red) / (float)plane;
        float s2 = 0.f;
        for (int64_t i = threadIdx.x; i < plane; i += 256) {
            const float d = value(i) - mean;
            s2 = fmaf(d, d, s2);
        }
        inv_std = rsqrtf(block_sum(s2, red) / (float)plane + 1e-5f);
    }
    float gain = inv_std, shift = 0.f;
    if (style) {
        const float s0 = style[(size_t)b * style_stride + c] + 1.f, s1 = style[(size_t)b * style_stride + channels + c];
        gain = inv_std * s0;
        shift = s1;
    }
    for (int64_t i = threadIdx.x; i < plane; i += 256) yp[i] = fmaf(value(i) - mean, gain, shift);
}

}  // namespace

extern "C" int maua_sg1_epilogue_f32(const float* x, const float* bias, const float* noise, int64_t noise_batch_stride,
                                     const float* noise_w, const float* style, int style_stride, float* y, int batch, int channels,
                                     int h, int w, int instance_norm, void* stream) {
    if (!x || !y || batch <= 0 || channels <= 0 || h <= 0 || w <= 0 || batch > 65535) return MAUA_EINVAL;
    if (noise && !noise_w) return MAUA_EINVAL;
    hipLaunchKernelGGL(sg1_epilogue_kernel, dim3((unsigned)channels, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, bias, noise,
                       noise_batch_stride, noise_w, style, style_stride, y, channels, (int64_t)h * w, instance_norm);
    MAUA_LAUNCH_CHECK();
    return 0;
}

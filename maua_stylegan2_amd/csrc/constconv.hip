// The first StyledConv of a generator with a ConstantInput (models/stylegan2.py:269-278, :547-549: conv1 on the learned 4 x 4 constant).
//
// Behavioural contract: /root/reference/models/stylegan2.py:217-254 (ModulatedConv2d, plain branch) + :338-343 (noise, bias, leaky ReLU)
// applied to input = const.repeat(batch) (:276-278).  The reference (and maua_modconv3x3_f32) treats this layer like any other: a
// 512 x 512 x 3 x 3 convolution over [B, 512, 4, 4] — 0.6 GFLOP per batch of 8, a split-K launch + its reduction (38 us: the kernel is
// latency-bound, its tiles half padding).  But the input is the SAME constant for every frame, scaled per (frame, channel) by the styles:
//
//     y[b, o, p] = sum_i ( sum_taps W[o, i, ky, kx] c[i, p + (ky, kx) - 1] ) s[b, i]  =  sum_i T[o, p, i] s[b, i],
//
// with T a function of the checkpoint alone (maua_pack_const_conv_f32, once per weight version): the layer becomes a [Cout * 16, Cin] x
// [Cin, B] product — 1/9 of the multiply-adds, no spatial gather, no split-K — bound by one read of T (16.8 MB for 512 channels).
// The same launch applies demodulation, noise, bias, leaky ReLU (reduce_tail_kernel's order) and, for the ToRGB that follows, leaves
// per-32-channel-group partial sums for maua_torgb_f32's plane-sum form.  fp32 rounding differs from the convolution's only in the
// association of the same products (the parity tests hold both against the oracle).
//
// Layout of T: [Cout / 32][16 positions][Cin / 8][32 channels][8] — a workgroup (one channel group, one position) reads 1 KB per step.
#include "common.h"

namespace {

constexpr int CC_HW = 16;  // the constant is 4 x 4
constexpr int CC_MAXB = 8; // frames per pass (the LDS layout and the two 16-byte reads per channel assume 8)

__global__ __launch_bounds__(256) void pack_const_conv_kernel(const float* __restrict__ w, const float* __restrict__ c, float* __restrict__ T,
                                                              int cout, int cin) {
    const int64_t total = (int64_t)cout * cin;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int i = (int)(idx % cin), o = (int)(idx / cin);
        float k[3][3], v[4][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) k[a][b] = w[((size_t)o * cin + i) * 9 + a * 3 + b];
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) v[y][x] = c[(size_t)i * CC_HW + y * 4 + x];
        const int g = o / 32, ol = o % 32, kc = i / 8, il = i % 8;
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                float acc = 0.f;  // F.conv2d, padding 1: out[y][x] = sum w[ky][kx] in[y + ky - 1][x + kx - 1]
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const int yy = y + a - 1, xx = x + b - 1;
                        if (yy >= 0 && yy < 4 && xx >= 0 && xx < 4) acc = fmaf(k[a][b], v[yy][xx], acc);
                    }
                T[((((size_t)g * CC_HW + (y * 4 + x)) * (cin / 8) + kc) * 32 + ol) * 8 + il] = acc;
            }
    }
}

// One workgroup per (group of 32 output channels, position): thread (channel ol = tid / 8, lane il = tid % 8) walks input channels
// il, il + 8, ...; the styles of up to eight frames sit in LDS.
__global__ __launch_bounds__(256) void const_conv_kernel(const float* __restrict__ T, const float* __restrict__ s, int s_stride,
                                                         const float* __restrict__ d, float* __restrict__ y, const float* __restrict__ noise,
                                                         int64_t noise_batch_stride, const float* __restrict__ noise_w,
                                                         const float* __restrict__ bias, const float* __restrict__ rgb_w,
                                                         const float* __restrict__ rgb_s, float rgb_wscale, float* __restrict__ rgb_part,
                                                         int batch, int cin, int cout, float wscale,
                                                         const maua_frame_source_t* __restrict__ src, int noise_slot) {
    extern __shared__ __attribute__((aligned(16))) float S[];  // [cin][CC_MAXB]: the eight frames' styles of a channel side by side (two 16-byte reads)
    __shared__ float part[3][32][CC_MAXB];
    const int tid = threadIdx.x, il = tid & 7, ol = tid >> 3;
    const int p = blockIdx.x % CC_HW, g = blockIdx.x / CC_HW;
    const int o = g * 32 + ol, groups = cout / 32, nk = cin / 8;
    if (src) {
        noise_batch_stride = src->noise_stride[noise_slot];
        noise = src->noise[noise_slot];
        if (noise) noise += (int64_t)src->frame0 * noise_batch_stride;
    }
    const float nw = noise ? noise_w[0] : 0.f;
    const float bv = bias ? bias[o] : 0.f;
    const float* Tp = T + (((size_t)g * CC_HW + p) * nk * 32 + ol) * 8 + il;
    // the first 64 steps of this thread's row of T (all of it up to 512 input channels): it does not depend on the frames, so it is on its
    // way before the styles are staged — 64 loads in flight, ONE round trip (16 at a time behind the staging barrier: 15.5 us per launch)
    float t0[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) t0[k] = (k < nk) ? Tp[(size_t)k * 256] : 0.f;
    for (int b0 = 0; b0 < batch; b0 += CC_MAXB) {
        const int nb = min(CC_MAXB, batch - b0);
        __syncthreads();
        for (int e0 = 0; e0 < CC_MAXB * cin; e0 += 16 * 256) {  // (16 loads per thread in flight: as a plain loop the staging was 16 dependent round trips)
            float sv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + tid + 256 * q;
                const int bb = e / cin, i = e - bb * cin;
                sv[q] = (e < CC_MAXB * cin && bb < nb) ? s[(size_t)(b0 + bb) * s_stride + i] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + tid + 256 * q;
                const int bb = e / cin, i = e - bb * cin;
                if (e < CC_MAXB * cin) S[i * CC_MAXB + bb] = sv[q];
            }
        }
        // the epilogue's operands of this thread's frame (lane il finishes frame b0 + il)
        const bool mine = il < nb;
        const float dv = (mine && d) ? d[(size_t)(b0 + il) * cout + o] : 1.f;
        const float nzv = (mine && nw != 0.f) ? noise[(size_t)(b0 + il) * noise_batch_stride + p] : 0.f;
        float rw[3] = {0.f, 0.f, 0.f};
        if (mine && rgb_part) {
            const float ms = rgb_s[(size_t)(b0 + il) * s_stride + o];
#pragma unroll
            for (int c = 0; c < 3; ++c) rw[c] = (rgb_wscale * rgb_w[c * cout + o]) * ms;  // (as torgb_kernel forms them)
        }
        __syncthreads();
        float acc[CC_MAXB];
#pragma unroll
        for (int bb = 0; bb < CC_MAXB; ++bb) acc[bb] = 0.f;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const int i = (k < nk) ? k * 8 + il : 0;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(S + i * CC_MAXB), hi = *reinterpret_cast<const f32x4*>(S + i * CC_MAXB + 4);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) acc[bb] = fmaf(t0[k], lo[bb], acc[bb]), acc[4 + bb] = fmaf(t0[k], hi[bb], acc[4 + bb]);
        }
        for (int k0 = 64; k0 < nk; k0 += 64) {  // (more than 512 input channels: the rest of the row, 64 steps at a time)
            float t[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) t[k] = (k0 + k < nk) ? Tp[(size_t)(k0 + k) * 256] : 0.f;
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int i = (k0 + k < nk) ? (k0 + k) * 8 + il : 0;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(S + i * CC_MAXB), hi = *reinterpret_cast<const f32x4*>(S + i * CC_MAXB + 4);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) acc[bb] = fmaf(t[k], lo[bb], acc[bb]), acc[4 + bb] = fmaf(t[k], hi[bb], acc[4 + bb]);
            }
        }
        // the eight input-channel lanes of an output channel are neighbours: butterfly over 1, 2, 4; lane il keeps frame il
        float v = 0.f;
#pragma unroll
        for (int bb = 0; bb < CC_MAXB; ++bb) {
            float a = acc[bb];
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 4, 64);
            if (il == bb) v = a;
        }
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if (mine) {
#pragma clang fp contract(off)  // (reduce_tail_kernel's roundings: (x * wscale) * d, + nw * noise, + bias)
            float tt = v * wscale;
            tt = tt * dv;
            const float nz = nw * nzv;
            tt = lrelu_gain((tt + nz) + bv);
            y[((size_t)(b0 + il) * cout + o) * CC_HW + p] = tt;
            r0 = rw[0] * tt, r1 = rw[1] * tt, r2 = rw[2] * tt;
        }
        if (rgb_part) {
            part[0][ol][il] = r0, part[1][ol][il] = r1, part[2][ol][il] = r2;
            __syncthreads();
            if (tid < 3 * CC_MAXB) {
                const int c = tid / CC_MAXB, bb = tid % CC_MAXB;
                if (bb < nb) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 32; ++k) a += part[c][k][bb];
                    rgb_part[((size_t)(b0 + bb) * 3 * groups + 3 * g + c) * CC_HW + p] = a;
                }
            }
        }
    }
}

}  // namespace

extern "C" int maua_const_conv_ok(int cin, int cout, int h, int w) { return h == 4 && w == 4 && cin > 0 && cin % 8 == 0 && cout > 0 && cout % 32 == 0; }

extern "C" int maua_pack_const_conv_f32(const float* w, const float* c, float* T, int cout, int cin, int h, int wd, void* stream) {
    if (!w || !c || !T) return MAUA_EINVAL;
    if (!maua_const_conv_ok(cin, cout, h, wd)) return MAUA_ENOSYS;
    const int64_t blocks = ceil_div64((int64_t)cout * cin, 256);
    hipLaunchKernelGGL(pack_const_conv_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, w, c, T, cout, cin);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_const_styledconv_f32(const float* T, const float* s, int s_stride, const float* d, float* y, const float* noise,
                                         int64_t noise_batch_stride, const float* noise_w, const float* bias, const float* rgb_w,
                                         const float* rgb_s, float rgb_wscale, float* rgb_partial, const maua_frame_source_t* src,
                                         int noise_slot, int batch, int cin, int cout, int h, int w, float wscale, void* stream) {
    if (!T || !s || !y || batch <= 0) return MAUA_EINVAL;
    if (!maua_const_conv_ok(cin, cout, h, w)) return MAUA_ENOSYS;
    if ((noise || src) && !noise_w) return MAUA_EINVAL;
    if (src && (noise_slot < 0 || noise_slot >= MAUA_MAX_NOISE_SLOTS)) return MAUA_EINVAL;
    if (rgb_partial && (!rgb_w || !rgb_s)) return MAUA_EINVAL;
    const size_t lds = (size_t)CC_MAXB * cin * sizeof(float);
    if (lds > 48 * 1024) return MAUA_ENOSYS;
    hipLaunchKernelGGL(const_conv_kernel, dim3((unsigned)((cout / 32) * CC_HW)), dim3(256), lds, (hipStream_t)stream, T, s, s_stride, d, y, noise,
                       noise_batch_stride, noise_w, bias, rgb_w, rgb_s, rgb_wscale, rgb_partial, batch, cin, cout, wscale, src, noise_slot);
    MAUA_LAUNCH_CHECK();
    return 0;
}

// ToRGB for gfx950: 1x1 modulated conv (no demod) + bias + FIR-upsampled skip, one HBM-bound launch.
//
// Behavioural contract: /root/reference/models/stylegan2.py:346-365 (ToRGB), :34-52 (Upsample = upfirdn2d up 2,
// pad (2,1)), :353 demodulate=False.  The reference runs linear, mul, grouped conv, add, upfirdn2d, add (6 launches
// and 4 extra passes over [B,3,H,W]); here the feature map is read exactly once with 16-byte loads, the three
// modulated weight rows live in LDS, and the 2x2 polyphase taps of the skip are gathered on the fly (the skip is
// 3 channels at quarter area — it stays in L2).  Small planes split the channel loop across threads (KS slices)
// and combine through LDS so a 4x4x512 layer is not a 512-deep serial dependency chain.
#include "common.h"

namespace {

template <int V>
__device__ __forceinline__ float4 load_px(const float* p) {
    if (V == 4) return *reinterpret_cast<const float4*>(p);
    return make_float4(p[0], 0.f, 0.f, 0.f);
}

template <int V>
__global__ __launch_bounds__(256) void torgb_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ s, int s_stride,
                                                    const float* __restrict__ bias, const float* __restrict__ skip,
                                                    const float* __restrict__ k4, float* __restrict__ y, int cin, int h,
                                                    int wdt, float wscale, int ks_log2, int quads_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // wm[3][cin] then partial[KS][QPB][12]
    float* wm = lds;
    float* part = lds + ((3 * cin + 3) & ~3);
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int plane = h * wdt;
    const int quads = plane / V;
    for (int e = tid; e < 3 * cin; e += 256) {
        const int i = e % cin;
        wm[e] = wscale * w[e] * s[(size_t)b * s_stride + i];
    }
    __syncthreads();

    const int ks = 1 << ks_log2;
    const int qi = tid & (quads_per_block - 1);
    const int slice = tid / quads_per_block;
    const int q = blockIdx.x * quads_per_block + qi;
    const bool active = q < quads;
    float4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const int per = (cin + ks - 1) / ks;
        const int i0 = slice * per;
        const int i1 = min(cin, i0 + per);
        const float* xp = x + ((size_t)b * cin) * plane + (size_t)q * V;
        int i = i0;
        for (; i + 8 <= i1; i += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load_px<V>(xp + (size_t)(i + u) * plane);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float wv = wm[c * cin + i + u];
                    acc[c].x = fmaf(wv, v[u].x, acc[c].x);
                    acc[c].y = fmaf(wv, v[u].y, acc[c].y);
                    acc[c].z = fmaf(wv, v[u].z, acc[c].z);
                    acc[c].w = fmaf(wv, v[u].w, acc[c].w);
                }
        }
        for (; i < i1; ++i) {
            const float4 v = load_px<V>(xp + (size_t)i * plane);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float wv = wm[c * cin + i];
                acc[c].x = fmaf(wv, v.x, acc[c].x);
                acc[c].y = fmaf(wv, v.y, acc[c].y);
                acc[c].z = fmaf(wv, v.z, acc[c].z);
                acc[c].w = fmaf(wv, v.w, acc[c].w);
            }
        }
    }
    if (ks > 1) {
        float* mine = part + ((size_t)slice * quads_per_block + qi) * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(mine + c * 4) = acc[c];
        __syncthreads();
        if (slice == 0) {
            for (int sl = 1; sl < ks; ++sl) {
                const float* o = part + ((size_t)sl * quads_per_block + qi) * 12;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float4 v = *reinterpret_cast<const float4*>(o + c * 4);
                    acc[c].x += v.x, acc[c].y += v.y, acc[c].z += v.z, acc[c].w += v.w;
                }
            }
        }
    }
    if (!active || slice != 0) return;

    const int pix = q * V;
    const int Y = pix / wdt, X0 = pix - Y * wdt;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float out[4] = {acc[c].x, acc[c].y, acc[c].z, acc[c].w};
        const float bc = bias ? bias[c] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] += bc;
        if (skip) {
            const int sh = h >> 1, sw = wdt >> 1;
            const float* sp = skip + ((size_t)b * 3 + c) * sh * sw;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int X = X0 + e;
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cy = Y + i - 2;  // zero-stuffed canvas row (pad0 = 2)
                    if (cy < 0 || (cy & 1)) continue;
                    const int iy = cy >> 1;
                    if (iy >= sh) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cx = X + j - 2;
                        if (cx < 0 || (cx & 1)) continue;
                        const int ix = cx >> 1;
                        if (ix >= sw) continue;
                        a = fmaf(k4[(3 - i) * 4 + (3 - j)], sp[iy * sw + ix], a);
                    }
                }
                out[e] += a;
            }
        }
        float* yp = y + ((size_t)b * 3 + c) * plane + pix;
        if (V == 4) *reinterpret_cast<float4*>(yp) = make_float4(out[0], out[1], out[2], out[3]);
        else yp[0] = out[0];
    }
}

// w == NULL (and s == NULL): x holds per-tile partial ToRGB sums [B, 3 M, H, W], plane 3 m + c feeding colour c (what the 2-D Winograd layers
// of more than one weight tile leave, maua_styledconv_torgb_partial_f32): y[b, c] = sum_m x[b, 3 m + c] + bias[c] + up2(skip)[b, c].
// Until round 6 this went through torgb_kernel with a 0 / 1 selection matrix: ~20 us per launch whatever the map size (five launches per
// 1024^2 forward, tools/rocpd_timeline.py), because that kernel is four DEPENDENT round trips — weights x styles into LDS, the channel loop,
// the slice combine, the skip gather with its tap loads.  Here a thread owns four pixels of one colour and every load it needs — M plane
// quads, the 2 x 4 window of the skip image, the taps (uniform) — is independent of every other: one round trip.
__global__ __launch_bounds__(256) void rgb_planes_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                         const float* __restrict__ skip, const float* __restrict__ k4,
                                                         float* __restrict__ y, int m_tiles, int h, int wdt) {
    const int plane = h * wdt;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y % 3, b = blockIdx.y / 3;
    if (q * 4 >= plane) return;
    const int pix = q * 4;
    const int Y = pix / wdt, X0 = pix - Y * wdt;
    const float* xp = x + ((size_t)b * 3 * m_tiles + c) * plane + pix;
    float4 acc = *reinterpret_cast<const float4*>(xp);
    // skip window: canvas row Y + i - 2 holds data for even values only: tap rows i0 = Y & 1 and i0 + 2 on skip rows iy0, iy0 + 1;
    // columns X0 / 2 - 1 .. X0 / 2 + 2 serve the four pixels (even X: taps 0, 2; odd X: taps 1, 3)
    float v[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float kf[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float bc = bias ? bias[c] : 0.f;
    if (skip) {
        const int sh = h >> 1, sw = wdt >> 1;
        const float* sp = skip + ((size_t)b * 3 + c) * sh * sw;
        const int i0 = Y & 1;
        const int iy0 = (Y + i0 - 2) >> 1;  // (-1 for Y = 0, 1: the canvas padding)
        const int ix0 = (X0 >> 1) - 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int iy = iy0 + r;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int ix = ix0 + cc;
                if (iy >= 0 && iy < sh && ix >= 0 && ix < sw) v[r][cc] = sp[iy * sw + ix];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) kf[r][j] = k4[(3 - (i0 + 2 * r)) * 4 + (3 - j)];  // (uniform per row parity: scalar loads)
        }
    }
    for (int m = 1; m < m_tiles; ++m) {
        const float4 t = *reinterpret_cast<const float4*>(xp + (size_t)3 * m * plane);
        acc.x += t.x, acc.y += t.y, acc.z += t.z, acc.w += t.w;
    }
    float out[4] = {acc.x + bc, acc.y + bc, acc.z + bc, acc.w + bc};
    if (skip) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j0 = e & 1, ca = (e + 1) >> 1;  // columns ca (tap j0), ca + 1 (tap j0 + 2) of the window
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                a = fmaf(kf[r][j0], v[r][ca], a);
                a = fmaf(kf[r][j0 + 2], v[r][ca + 1], a);
            }
            out[e] += a;
        }
    }
    *reinterpret_cast<float4*>(y + ((size_t)b * 3 + c) * plane + pix) = make_float4(out[0], out[1], out[2], out[3]);
}

}  // namespace

extern "C" int maua_torgb_f32(const float* x, const float* w, const float* s, int s_stride, const float* bias,
                              const float* skip, const float* k4, float* y, int batch, int cin, int h, int wdt,
                              float wscale, void* stream) {
    if (!x || !y || batch <= 0 || cin <= 0 || h <= 0 || wdt <= 0 || (!w != !s)) return MAUA_EINVAL;
    if (!w) {  // plane sum of per-tile partial ToRGB sums (see rgb_planes_kernel)
        if (cin % 3 || wdt % 4 || (skip && (!k4 || (h & 1) || (wdt & 1)))) return MAUA_EINVAL;
        hipLaunchKernelGGL(rgb_planes_kernel, dim3(ceil_div(h * wdt / 4, 256), 3 * batch), dim3(256), 0, (hipStream_t)stream, x, bias, skip,
                           k4, y, cin / 3, h, wdt);
        MAUA_LAUNCH_CHECK();
        return 0;
    }
    const int vec = (wdt % 4 == 0) ? 4 : 1;      // a 4-pixel group must stay inside one row
    if (skip && (!k4 || (h & 1) || (wdt & 1))) return MAUA_EINVAL;
    const int quads = h * wdt / vec;
    // Small planes are latency bound (a 512-deep channel loop is 64 dependent trips): trade pixel groups per workgroup
    // for channel slices until the grid has ~512 workgroups or a slice is down to 8 channels.
    int qpb = 256, ks_log2 = 0;
    while (qpb > 4 && (qpb / 2 >= quads || ((int64_t)ceil_div(quads, qpb) * batch < 512 && (cin >> ks_log2) > 8)))
        qpb >>= 1, ++ks_log2;
    const int ks = 1 << ks_log2;
    const size_t lds = ((size_t)((3 * cin + 3) & ~3) + (ks > 1 ? (size_t)ks * qpb * 12 : 0)) * sizeof(float);
    if (vec == 4)
        hipLaunchKernelGGL(torgb_kernel<4>, dim3(ceil_div(quads, qpb), batch), dim3(256), lds, (hipStream_t)stream, x, w,
                           s, s_stride, bias, skip, k4, y, cin, h, wdt, wscale, ks_log2, qpb);
    else
        hipLaunchKernelGGL(torgb_kernel<1>, dim3(ceil_div(quads, qpb), batch), dim3(256), lds, (hipStream_t)stream, x, w,
                           s, s_stride, bias, skip, k4, y, cin, h, wdt, wscale, ks_log2, qpb);
    MAUA_LAUNCH_CHECK();
    return 0;
}

// Style affines + demodulation factors for every layer of one generator forward: two table-driven launches.
//
// Behavioural contract: /root/reference/models/stylegan2.py:140-146 (EqualLinear), :207,220 (modulation, bias_init 1),
// :223-225 (demodulation), :541-543 (truncation lerp).  The reference runs 26 F.linear + 17 pow/sum/rsqrt launches
// per forward at 1024^2; these are GEMV-sized (<= 512 x 512 per layer), so the only goal here is to not pay 43
// launch boundaries: a workgroup owns 16 output rows of one layer (4 per wave, their weight rows all fetched up front so
// the loads overlap), keeps the (truncated) latent rows / squared styles of up to 16 frames in LDS and reduces with
// wave64 butterfly shuffles.  The forward waits on these two launches before its first conv, so they are sized for
// latency (832 small workgroups), not for bandwidth.
#include "common.h"

namespace {

constexpr int BCHUNK = 16;   // frames staged in LDS per pass
constexpr int MAX_PER_LANE = 16;  // style_dim, cin <= 1024
constexpr int ROWS = 16;          // output rows per workgroup
constexpr int RPW = ROWS / 4;     // rows per wave

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void style_affine_kernel(const float* __restrict__ latents, int batch, int n_latent,
                                                           int style_dim, const float* __restrict__ trunc,
                                                           const float* __restrict__ trunc_latent,
                                                           const maua_style_layer_t* __restrict__ table,
                                                           float* __restrict__ s, int s_stride,
                                                           const maua_frame_source_t* __restrict__ src) {
    extern __shared__ __attribute__((aligned(16))) float lat[];  // [BCHUNK][style_dim]
    if (src) {  // per-frame sequences resident in HBM: this launch starts at frame src->frame0 (uniform scalar loads)
        const int64_t f0 = src->frame0;
        latents = src->latents + f0 * n_latent * style_dim;
        trunc = src->trunc ? src->trunc + f0 : nullptr;
    }
    const maua_style_layer_t L = table[blockIdx.x];
    const int row0 = blockIdx.y * ROWS;
    if (row0 >= L.cin) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float inv = 1.0f / sqrtf((float)style_dim);
    const int per_lane = style_dim / 64;

    // this wave's weight rows: fetched BEFORE the latents are staged (they depend on the table entry only; behind the staging barrier their
    // round trip was a fourth dependent one on a launch the whole forward waits for)
    float w[RPW][MAX_PER_LANE], bias[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int i = min(row0 + wave * RPW + r, L.cin - 1);
#pragma unroll
        for (int q = 0; q < MAX_PER_LANE; ++q)
            w[r][q] = (q < per_lane) ? L.mod_w[(size_t)i * style_dim + q * 64 + lane] : 0.f;
        bias[r] = L.mod_b[i];
    }
    for (int bc = 0; bc < batch; bc += BCHUNK) {
        const int nb = min(BCHUNK, batch - bc);
        __syncthreads();
        for (int e0 = 0; e0 < nb * style_dim; e0 += 16 * 256) {  // 16 loads per thread in flight (a plain loop staged them as 16 dependent round trips)
            float v[16], tlv[16], tr[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + tid + 256 * q;
                const bool ok = e < nb * style_dim;
                const int b = ok ? e / style_dim : 0, j = ok ? e - b * style_dim : 0;
                v[q] = ok ? latents[((size_t)(bc + b) * n_latent + L.lat_idx) * style_dim + j] : 0.f;
                tlv[q] = (ok && trunc && trunc_latent) ? trunc_latent[j] : 0.f;
                tr[q] = (ok && trunc) ? trunc[bc + b] : 1.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + tid + 256 * q;
                if (e < nb * style_dim) lat[e] = trunc ? tlv[q] + tr[q] * (v[q] - tlv[q]) : v[q];
            }
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            float acc[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < MAX_PER_LANE; ++q)
                if (q < per_lane) {
                    const float lv = lat[b * style_dim + q * 64 + lane];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) acc[r] = fmaf(w[r][q], lv, acc[r]);
                }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float v = wave_sum(acc[r]);
                const int i = row0 + wave * RPW + r;
                if (lane == 0 && i < L.cin) s[(size_t)(bc + b) * s_stride + L.s_off + i] = v * inv + bias[r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void demod_kernel(const maua_style_layer_t* __restrict__ table,
                                                    const float* __restrict__ s, int s_stride, float* __restrict__ d,
                                                    int batch) {
    extern __shared__ __attribute__((aligned(16))) float s2[];  // [BCHUNK][cin]
    const maua_style_layer_t L = table[blockIdx.x];
    if (!L.wsq) return;
    const int row0 = blockIdx.y * ROWS;
    if (row0 >= L.cout) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_lane = (L.cin + 63) / 64;
    const float ws2 = L.wscale * L.wscale;

    float w[RPW][MAX_PER_LANE];  // (fetched before the styles are staged: see style_affine_kernel)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int o = min(row0 + wave * RPW + r, L.cout - 1);
#pragma unroll
        for (int q = 0; q < MAX_PER_LANE; ++q) {
            const int i = q * 64 + lane;
            w[r][q] = (q < per_lane && i < L.cin) ? L.wsq[(size_t)o * L.cin + i] : 0.f;
        }
    }
    for (int bc = 0; bc < batch; bc += BCHUNK) {
        const int nb = min(BCHUNK, batch - bc);
        __syncthreads();
        for (int e0 = 0; e0 < nb * L.cin; e0 += 16 * 256) {  // (16 loads per thread in flight: see style_affine_kernel)
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + tid + 256 * q;
                const bool ok = e < nb * L.cin;
                const int b = ok ? e / L.cin : 0, i = ok ? e - b * L.cin : 0;
                v[q] = ok ? s[(size_t)(bc + b) * s_stride + L.s_off + i] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + tid + 256 * q;
                if (e < nb * L.cin) s2[e] = v[q] * v[q];
            }
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            float acc[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < MAX_PER_LANE; ++q) {
                const int i = q * 64 + lane;
                if (q < per_lane && i < L.cin) {
                    const float sv = s2[b * L.cin + i];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) acc[r] = fmaf(w[r][q], sv, acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const float v = wave_sum(acc[r]);
                const int o = row0 + wave * RPW + r;
                if (lane == 0 && o < L.cout) d[L.d_off + (size_t)(bc + b) * L.cout + o] = rsqrtf(v * ws2 + 1e-8f);
            }
        }
    }
}

}  // namespace

extern "C" int maua_style_affine_f32(const float* latents, int batch, int n_latent, int style_dim, const float* trunc,
                                     const float* trunc_latent, const maua_style_layer_t* table, int n_layers,
                                     int max_cin, float* s, int s_stride, const maua_frame_source_t* src, void* stream) {
    if ((!latents && !src) || !table || !s || batch <= 0 || n_layers <= 0 || max_cin <= 0) return MAUA_EINVAL;
    if (style_dim <= 0 || style_dim % 64 || style_dim > 64 * MAX_PER_LANE) return MAUA_EINVAL;
    const size_t lds = (size_t)BCHUNK * style_dim * sizeof(float);
    hipLaunchKernelGGL(style_affine_kernel, dim3(n_layers, ceil_div(max_cin, ROWS)), dim3(256), lds, (hipStream_t)stream,
                       latents, batch, n_latent, style_dim, trunc, trunc_latent, table, s, s_stride, src);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_demod_f32(const maua_style_layer_t* table, int n_layers, int max_cout, const float* s, int s_stride,
                              float* d, int batch, void* stream) {
    if (!table || !s || !d || batch <= 0 || n_layers <= 0 || max_cout <= 0) return MAUA_EINVAL;
    const size_t lds = (size_t)BCHUNK * 64 * MAX_PER_LANE * sizeof(float);
    hipLaunchKernelGGL(demod_kernel, dim3(n_layers, ceil_div(max_cout, ROWS)), dim3(256), lds, (hipStream_t)stream, table,
                       s, s_stride, d, batch);
    MAUA_LAUNCH_CHECK();
    return 0;
}

// Audio-feature and temporal kernels for gfx950 (the device side of audioreactive/).
//
//   temporal_fir   <- /root/reference/audioreactive/signal.py:319-368 gaussian_filter (circular FIR along time)
//   stft_power     <- librosa.stft + |.|^2 as called at signal.py:51,93,119 (centred, reflect-padded, Hann)
//   filterbank     <- mel / chroma filterbank projection (+ power_to_db), signal.py:51,119
//   perlin3d       <- /root/reference/audioreactive/latent.py:188-246
//   affine warp    <- /root/reference/audioreactive/bend.py:52-102 (ReflectionPad2d -> kornia affine -> CenterCrop)
//
// None of these is near a roofline that matters for the frames/s metric; they exist so that latents / noise are
// produced on-device (north_star) instead of round-tripping through host numpy.  The temporal FIR is the one heavy
// pre-processing stage (sigma = 128 on [n_frames,1,256,256] noise = 121 GMAC at 1800 frames): it is
// register-tiled 16 outputs deep along time so each staged sample feeds 16 FMAs, lanes run along the contiguous
// feature axis (dense 256-byte loads), taps are broadcast from LDS.
#include "common.h"

namespace {

// Temporal FIR along dim 0 of x[T][F] (gaussian_filter, reference audioreactive/signal.py:335-343: circular within one wrap, zero beyond it).
// A thread owns one feature column and FIR_TT = 32 consecutive output times; the source rows are walked once, 16 at a time: each loaded value
// feeds the 32 accumulators through taps j - u, whose 48-tap neighbourhood (three 16-tap windows of a zero-padded LDS copy, four ds_read_b128 per
// 16 rows — the addresses are uniform, the windows live in registers) is indexed STATICALLY after unrolling: 512 FMAs per 16 global loads and 4 LDS
// reads.  (Rounds 1-4: 16 output times and one broadcast ds_read_b32 per FMA — LDS-issue bound at 1/16 of the VALU rate; the filters of the
// default plugin, sigma 5 ... 20 frames = 41 ... 161 taps over 175 k features x 900 frames, cost 0.13-0.19 s of a 1.1 s warm generate().)
constexpr int FIR_TT = 32;       // output times per thread
constexpr int FIR_PAD_LO = 32;   // zero taps in front of the LDS copy (k >= -32)
constexpr int FIR_PAD_HI = 64;   // ... and behind it (k < ntaps + 64: the walk is rounded up to 16 rows)

__global__ __launch_bounds__(256) void temporal_fir_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                           float* __restrict__ y, int T, int64_t F, int radius) {
    extern __shared__ __attribute__((aligned(16))) float tp[];  // [FIR_PAD_LO + 2 * radius + 1 + FIR_PAD_HI], zero-padded
    const int ntaps = 2 * radius + 1;
    const int padded = FIR_PAD_LO + ntaps + FIR_PAD_HI;
    for (int i = threadIdx.x; i < padded; i += 256) {
        const int k = i - FIR_PAD_LO;
        tp[i] = (k >= 0 && k < ntaps) ? taps[k] : 0.f;
    }
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * FIR_TT;
    if (f >= F) return;
    float acc[FIR_TT];
#pragma unroll
    for (int u = 0; u < FIR_TT; ++u) acc[u] = 0.f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    float w0[16], w1[16], w2[16];  // taps j0 - 32 .. j0 - 17, j0 - 16 .. j0 - 1, j0 .. j0 + 15
#pragma unroll
    for (int i = 0; i < 16; ++i) w0[i] = 0.f, w1[i] = 0.f;
    bool dirty = false;
    // y[t] = sum_k taps[k] * xpad[t + k], xpad[i] = x[(i - radius) mod T] inside one wrap, 0 beyond (radius > T branch)
    const int n_rows = ntaps + FIR_TT - 1;
    for (int j0 = 0; j0 < n_rows; j0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(tp + FIR_PAD_LO + j0 + 4 * q);
            w2[4 * q] = v4.x, w2[4 * q + 1] = v4.y, w2[4 * q + 2] = v4.z, w2[4 * q + 3] = v4.w;
        }
        float v[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int i = t0 + j0 + jj - radius;  // un-wrapped source time (uniform: scalar selects, the 16 loads issue back to back)
            const bool inside = i >= -T && i < 2 * T;
            int w = i < 0 ? i + T : (i >= T ? i - T : i);
            w = inside ? w : 0;
            const float got = x[(int64_t)w * F + f];
            v[jj] = inside ? got : 0.f;
            dirty |= (__float_as_uint(v[jj]) & 0x7f800000u) == 0x7f800000u;  // inf / NaN: see the exact pass below
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
#pragma unroll
            for (int u = 0; u < FIR_TT; ++u) {
                const int d = jj - u;  // tap j0 + d
                acc[u] = fmaf(d >= 0 ? w2[d] : (d >= -16 ? w1[16 + d] : w0[32 + d]), v[jj], acc[u]);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) w0[i] = w1[i], w1[i] = w2[i];
    }
    if (dirty) {
        // A non-finite sample met zero-PADDED taps above (every row of the walk multiplies all 32 outputs, the taps outside the filter's
        // support being zeros: 0 * inf = NaN up to 45 frames outside the support).  The reference's conv1d (audioreactive/signal.py:359)
        // confines it to the support: such a thread — there are none in a sane render — recomputes its outputs tap by tap.
        for (int u = 0; u < FIR_TT; ++u) {
            float sum = 0.f;
            for (int k = 0; k < ntaps; ++k) {
                const int i = t0 + u + k - radius;
                const bool inside = i >= -T && i < 2 * T;
                const int w = i < 0 ? i + T : (i >= T ? i - T : i);
                sum = fmaf(tp[FIR_PAD_LO + k], inside ? x[(int64_t)w * F + f] : 0.f, sum);
            }
            if (t0 + u < T) y[(int64_t)(t0 + u) * F + f] = sum;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < FIR_TT; ++u)
        if (t0 + u < T) y[(int64_t)(t0 + u) * F + f] = acc[u];
}

// One workgroup = one STFT frame: radix-2 DIT FFT of n_fft (<= 4096) complex points in LDS, twiddles from a table
// built once per workgroup with sincospif.
__global__ __launch_bounds__(256) void stft_power_kernel(const float* __restrict__ y, int64_t n, const float* __restrict__ win,
                                                         int n_fft, int log2n, int hop, float* __restrict__ p,
                                                         int n_frames) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* re = sm;
    float* im = sm + n_fft;
    float* twc = sm + 2 * n_fft;  // cos(-2 pi k / n), k < n/2
    float* tws = twc + n_fft / 2;
    const int frame = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t start = (int64_t)frame * hop - n_fft / 2;
    for (int i = tid; i < n_fft; i += 256) {
        int64_t src = start + i;  // reflect padding (numpy 'reflect': no edge repeat)
        if (n > 1) {
            const int64_t period = 2 * (n - 1);
            src %= period;
            if (src < 0) src += period;
            if (src >= n) src = period - src;
        } else {
            src = 0;
        }
        const int rev = (int)(__brev((unsigned)i) >> (32 - log2n));
        re[rev] = y[src] * win[i];
        im[rev] = 0.f;
    }
    for (int k = tid; k < n_fft / 2; k += 256) {
        float s_, c_;
        sincospif(-2.0f * (float)k / (float)n_fft, &s_, &c_);
        twc[k] = c_;
        tws[k] = s_;
    }
    __syncthreads();
    for (int st = 1; st <= log2n; ++st) {
        const int half = 1 << (st - 1);
        const int tw_stride = n_fft >> st;
        for (int bfly = tid; bfly < n_fft / 2; bfly += 256) {
            const int grp = bfly / half, k = bfly - grp * half;
            const int a = grp * 2 * half + k, b = a + half;
            const float c_ = twc[k * tw_stride], s_ = tws[k * tw_stride];
            const float br = re[b] * c_ - im[b] * s_;
            const float bi = re[b] * s_ + im[b] * c_;
            const float ar = re[a], ai = im[a];
            re[a] = ar + br, im[a] = ai + bi;
            re[b] = ar - br, im[b] = ai - bi;
        }
        __syncthreads();
    }
    for (int k = tid; k <= n_fft / 2; k += 256) p[(int64_t)k * n_frames + frame] = re[k] * re[k] + im[k] * im[k];
}

// ------------------------------------------------------------------------------------------------ HPSS building blocks
// Complex STFT / inverse STFT (same LDS radix-2 FFT as stft_power_kernel), 31-tap median filters along time and
// frequency, soft masks: the pieces of librosa.effects.percussive / harmonic as called at
// /root/reference/audioreactive/signal.py:49,150.
__device__ __forceinline__ void lds_fft_radix2(float* re, float* im, const float* twc, const float* tws, int n_fft, int log2n,
                                               int tid) {
    for (int st = 1; st <= log2n; ++st) {
        const int half = 1 << (st - 1);
        const int tw_stride = n_fft >> st;
        for (int bfly = tid; bfly < n_fft / 2; bfly += 256) {
            const int grp = bfly / half, k = bfly - grp * half;
            const int a = grp * 2 * half + k, b = a + half;
            const float c_ = twc[k * tw_stride], s_ = tws[k * tw_stride];
            const float br = re[b] * c_ - im[b] * s_;
            const float bi = re[b] * s_ + im[b] * c_;
            const float ar = re[a], ai = im[a];
            re[a] = ar + br, im[a] = ai + bi;
            re[b] = ar - br, im[b] = ai - bi;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void stft_complex_kernel(const float* __restrict__ y, int64_t n, const float* __restrict__ win,
                                                           int n_fft, int log2n, int hop, float* __restrict__ out_re,
                                                           float* __restrict__ out_im, int n_frames) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* re = sm;
    float* im = sm + n_fft;
    float* twc = sm + 2 * n_fft;
    float* tws = twc + n_fft / 2;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int64_t start = (int64_t)frame * hop - n_fft / 2;
    for (int i = tid; i < n_fft; i += 256) {
        int64_t src = start + i;
        if (n > 1) {
            const int64_t period = 2 * (n - 1);
            src %= period;
            if (src < 0) src += period;
            if (src >= n) src = period - src;
        } else {
            src = 0;
        }
        const int rev = (int)(__brev((unsigned)i) >> (32 - log2n));
        re[rev] = y[src] * win[i];
        im[rev] = 0.f;
    }
    for (int k = tid; k < n_fft / 2; k += 256) {
        float s_, c_;
        sincospif(-2.0f * (float)k / (float)n_fft, &s_, &c_);
        twc[k] = c_, tws[k] = s_;
    }
    __syncthreads();
    lds_fft_radix2(re, im, twc, tws, n_fft, log2n, tid);
    for (int k = tid; k <= n_fft / 2; k += 256) {
        out_re[(int64_t)k * n_frames + frame] = re[k];
        out_im[(int64_t)k * n_frames + frame] = im[k];
    }
}

// One workgroup = one frame: Hermitian-extend the half spectrum, inverse FFT (conj . FFT . conj / N), window, and write the
// windowed frame to frames[frame][n_fft]; overlap-add happens in istft_ola_kernel.
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ in_re, const float* __restrict__ in_im,
                                                           const float* __restrict__ win, int n_fft, int log2n,
                                                           int n_frames, float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* re = sm;
    float* im = sm + n_fft;
    float* twc = sm + 2 * n_fft;
    float* tws = twc + n_fft / 2;
    const int frame = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < n_fft; k += 256) {
        const int kk = k <= n_fft / 2 ? k : n_fft - k;
        const float r = in_re[(int64_t)kk * n_frames + frame];
        float i_ = in_im[(int64_t)kk * n_frames + frame];
        if (k > n_fft / 2) i_ = -i_;          // X[N-k] = conj(X[k])
        if (kk == 0 || kk == n_fft / 2) i_ = 0.f;  // DC / Nyquist are real for a real signal
        const int rev = (int)(__brev((unsigned)k) >> (32 - log2n));
        re[rev] = r;
        im[rev] = -i_;  // conj before the forward transform
    }
    for (int k = tid; k < n_fft / 2; k += 256) {
        float s_, c_;
        sincospif(-2.0f * (float)k / (float)n_fft, &s_, &c_);
        twc[k] = c_, tws[k] = s_;
    }
    __syncthreads();
    lds_fft_radix2(re, im, twc, tws, n_fft, log2n, tid);
    const float inv = 1.0f / (float)n_fft;
    for (int i = tid; i < n_fft; i += 256) frames[(int64_t)frame * n_fft + i] = re[i] * inv * win[i];
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ win, int n_fft,
                                                        int hop, int n_frames, float* __restrict__ y, int64_t n_samples) {
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < n_samples; s += (int64_t)gridDim.x * 256) {
        const int64_t pos = s + n_fft / 2;  // position in the centre-padded signal
        int64_t t_hi = pos / hop;
        if (t_hi > n_frames - 1) t_hi = n_frames - 1;
        float acc = 0.f, wss = 0.f;
        for (int64_t t = t_hi; t >= 0; --t) {
            const int64_t off = pos - t * hop;
            if (off >= n_fft) break;
            acc += frames[t * n_fft + off];
            const float w = win[off];
            wss = fmaf(w, w, wss);
        }
        y[s] = wss > 1.17549435e-38f ? acc / wss : acc;
    }
}

// Median of a centred `size`-tap window (size odd, <= 31) along one axis of x[rows, cols]; scipy.ndimage 'reflect'
// boundary (d c b a | a b c d | d c b a).  Rank counting instead of sorting: branch-free and register resident.
template <int SIZE>
__global__ __launch_bounds__(256) void median_filter_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols,
                                                            int axis) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
        const int len = axis == 0 ? rows : cols;
        const int pos = axis == 0 ? r : c;
        float v[SIZE];
#pragma unroll
        for (int i = 0; i < SIZE; ++i) {
            int q = pos + i - SIZE / 2;
            // reflect with edge repeat, period 2*len
            const int period = 2 * len;
            q %= period;
            if (q < 0) q += period;
            if (q >= len) q = period - 1 - q;
            v[i] = axis == 0 ? x[(int64_t)q * cols + c] : x[(int64_t)r * cols + q];
        }
        float med = v[0];
#pragma unroll
        for (int i = 0; i < SIZE; ++i) {
            int rank = 0;
#pragma unroll
            for (int j = 0; j < SIZE; ++j) rank += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
            if (rank == SIZE / 2) med = v[i];
        }
        y[idx] = med;
    }
}

// librosa.util.softmask + mask application: out = D * mask(X, X_ref * margin)   (X = |D| median-filtered one way,
// X_ref the other way).  mask = (X/Z)^p / ((X/Z)^p + (Xr/Z)^p), Z = max(X, Xr); where Z underflows: 0.5 if split else 0.
__global__ __launch_bounds__(256) void softmask_apply_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                             const float* __restrict__ xs, const float* __restrict__ xref,
                                                             float margin, float power, int split_zeros,
                                                             float* __restrict__ out_re, float* __restrict__ out_im, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = xs[i], b = xref[i] * margin;
        const float z = fmaxf(a, b);
        float m;
        if (z < 1.17549435e-38f) {
            m = split_zeros ? 0.5f : 0.f;
        } else {
            const float pa = powf(a / z, power), pb = powf(b / z, power);
            m = pa / (pa + pb);
        }
        out_re[i] = re[i] * m;
        out_im[i] = im[i] * m;
    }
}

__global__ __launch_bounds__(256) void filterbank_kernel(const float* __restrict__ fb, const float* __restrict__ p,
                                                         float* __restrict__ out, int m, int k, int n, int to_db,
                                                         float amin) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= n) return;
    float acc = 0.f;
    const float* fr = fb + (size_t)row * k;
    for (int kk = 0; kk < k; ++kk) acc = fmaf(fr[kk], p[(size_t)kk * n + col], acc);
    if (to_db) acc = 10.0f * log10f(fmaxf(amin, acc));
    out[(size_t)row * n + col] = acc;
}

__device__ __forceinline__ float fade5(float t) { return t * t * t * (t * (t * 6.f - 15.f) + 10.f); }

__global__ __launch_bounds__(256) void perlin3d_kernel(const float* __restrict__ grad, float* __restrict__ out, int n0,
                                                       int n1, int n2, int r0, int r1, int r2) {
    const int64_t total = (int64_t)n0 * n1 * n2;
    const int d0 = n0 / r0, d1 = n1 / r1, d2 = n2 / r2;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int i2 = (int)(idx % n2);
        const int i1 = (int)((idx / n2) % n1);
        const int i0 = (int)(idx / ((int64_t)n1 * n2));
        const int c0 = i0 / d0, c1 = i1 / d1, c2 = i2 / d2;
        const float f0 = (float)(i0 - c0 * d0) / (float)d0;
        const float f1 = (float)(i1 - c1 * d1) / (float)d1;
        const float f2 = (float)(i2 - c2 * d2) / (float)d2;
        float nv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int o0 = c & 1, o1 = (c >> 1) & 1, o2 = (c >> 2) & 1;
            const float* gv = grad + ((((size_t)(c0 + o0) * (r1 + 1)) + (c1 + o1)) * (r2 + 1) + (c2 + o2)) * 3;
            nv[c] = (f0 - o0) * gv[0] + (f1 - o1) * gv[1] + (f2 - o2) * gv[2];
        }
        const float t0 = fade5(f0), t1 = fade5(f1), t2 = fade5(f2);
        const float n00 = nv[0] * (1 - t0) + t0 * nv[1];
        const float n10 = nv[2] * (1 - t0) + t0 * nv[3];
        const float n01 = nv[4] * (1 - t0) + t0 * nv[5];
        const float n11 = nv[6] * (1 - t0) + t0 * nv[7];
        const float m0 = (1 - t1) * n00 + t1 * n10;
        const float m1 = (1 - t1) * n01 + t1 * n11;
        out[idx] = ((1 - t2) * m0 + t2 * m1) * 2.f - 1.f;
    }
}

__device__ __forceinline__ int reflect_idx(int i, int n) {  // ReflectionPad2d semantics (no edge repeat), |i| < 2n
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

// Output pixel (oy, ox) of the h x w centre crop -> padded-canvas coords via the per-frame inverse affine map m
// (canvas pixel units), bilinear taps, zeros outside the canvas; canvas pixel -> reflected source pixel.
__global__ __launch_bounds__(256) void affine_reflect_warp_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                                  float* __restrict__ y, int channels, int h, int w,
                                                                  int pad_l, int pad_r, int pad_t, int pad_b,
                                                                  const float* __restrict__ add_noise,
                                                                  const int* __restrict__ xmap,
                                                                  const int* __restrict__ ymap,
                                                                  const maua_frame_source_t* __restrict__ src) {
    const int b = blockIdx.z;
    if (src) m += (size_t)src->frame0 * 6;  // m is the map sequence of the whole render
    const int ch = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= h * w) return;
    const int oy = pix / w, ox = pix - oy * w;
    const int ch_h = h + pad_t + pad_b, cw = w + pad_l + pad_r;
    // centre crop origin inside the canvas (kornia CenterCrop: floor((canvas - out) / 2))
    const float cy = (float)(oy + (ch_h - h) / 2), cx = (float)(ox + (cw - w) / 2);
    const float* a = m + (size_t)b * 6;
    const float sx = a[0] * cx + a[1] * cy + a[2];
    const float sy = a[3] * cx + a[4] * cy + a[5];
    const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
    const float fx = sx - x0, fy = sy - y0;
    const float* xp = x + ((size_t)b * channels + ch) * h * w;
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int yy = y0 + dy, xx = x0 + dx;
            if (yy < 0 || yy >= ch_h || xx < 0 || xx >= cw) continue;
            const float wgt = (dy ? fy : 1.f - fy) * (dx ? fx : 1.f - fx);
            // canvas pixel -> source pixel: one reflection fold, or the host's table when the canvas was built by SEVERAL
            // stacked ReflectionPad2d (each re-reflects the already padded canvas: not a single triangular fold)
            const int srow = ymap ? ymap[yy] : reflect_idx(yy - pad_t, h);
            const int scol = xmap ? xmap[xx] : reflect_idx(xx - pad_l, w);
            float v = xp[srow * w + scol];
            if (add_noise) v += add_noise[(size_t)yy * cw + xx];
            acc = fmaf(wgt, v, acc);
        }
    y[((size_t)b * channels + ch) * h * w + pix] = acc;
}


// |constant-Q transform| by definition (Brown 1991): out[k][t] = | sum_n y[t hop - N_k/2 + n] w_k[n] e^{-2 pi i f_k (n - N_k/2) / sr} |
// / sqrt(N_k), w_k = periodic Hann of N_k = lengths[k] samples normalised to unit L1, reflect padding at the ends.
// One workgroup per (frame, bin); phases and sums in fp64 (up to ~35k taps per low bin).  1.7 G taps for 30 s of audio:
// a few milliseconds, so the multirate / sparse-kernel machinery librosa needs on a CPU is unnecessary here.
__global__ __launch_bounds__(256) void cqt_mag_kernel(const float* __restrict__ y, int64_t n_samples,
                                                      const float* __restrict__ freqs, const int* __restrict__ lengths,
                                                      int hop, float sr, float* __restrict__ out, int n_frames) {
    __shared__ double red_re[256], red_im[256];
    const int t = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    const int n = lengths[k];
    const double cyc = (double)freqs[k] / (double)sr;  // cycles per sample
    const int64_t start = (int64_t)t * hop - n / 2;
    const int64_t period = n_samples > 1 ? 2 * (n_samples - 1) : 1;
    double re = 0.0, im = 0.0;
    for (int m = tid; m < n; m += 256) {
        int64_t j = start + m;
        j %= period;  // general reflect padding: ... 2 1 0 1 2 ... L-1 L-2 ...
        if (j < 0) j += period;
        if (j >= n_samples) j = period - j;
        const double w = 0.5 - 0.5 * cospi(2.0 * (double)m / (double)n);
        const double ph = cyc * (double)(m - n / 2);
        double sn, cs;
        sincospi(2.0 * (ph - floor(ph)), &sn, &cs);
        const double v = (double)y[j] * w;
        re += v * cs;
        im -= v * sn;
    }
    red_re[tid] = re, red_im[tid] = im;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red_re[tid] += red_re[tid + off], red_im[tid] += red_im[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const double wsum = 0.5 * (double)n;  // sum of a periodic Hann window
        out[(size_t)k * n_frames + t] = (float)(sqrt(red_re[0] * red_re[0] + red_im[0] * red_im[0]) / wsum / sqrt((double)n));
    }
}

// ------------------------------------------------------------------------------------------------ chroma post-processing
// CENS (Mueller & Ewert 2011; what librosa.feature.chroma_cens does after its chromagram, signal.py:115): per-frame L1
// normalisation -> 4-level quantisation -> Hann smoothing along time ('same', zero padded) -> per-frame L2 normalisation.
// ch / out are [n_bins, n_frames].  One thread per output frame: the 41 neighbour frames are re-normalised and re-quantised
// on the fly (12 x 41 loads per thread) — there is no intermediate array.
constexpr int CENS_MAX_BINS = 32;
__global__ __launch_bounds__(256) void chroma_cens_kernel(const float* __restrict__ ch, float* __restrict__ out, int n_bins,
                                                          int n_frames, int win_len) {
    extern __shared__ float cens_win[];  // [win_len] sum-normalised Hann taps (the interior of a win_len + 2 point window)
    if ((int)threadIdx.x < win_len) cens_win[threadIdx.x] = 0.5f - 0.5f * cospif(2.f * (threadIdx.x + 1) / (win_len + 1));
    __syncthreads();
    float wsum = 0.f;
    for (int j = 0; j < win_len; ++j) wsum += cens_win[j];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_frames) return;
    float acc[CENS_MAX_BINS];
#pragma unroll
    for (int b = 0; b < CENS_MAX_BINS; ++b) acc[b] = 0.f;
    const int half = win_len / 2;
    for (int j = 0; j < win_len; ++j) {
        const int tt = t + j - half;
        if (tt < 0 || tt >= n_frames) continue;
        float v[CENS_MAX_BINS], l1 = 0.f;
#pragma unroll
        for (int b = 0; b < CENS_MAX_BINS; ++b) {
            v[b] = b < n_bins ? ch[(size_t)b * n_frames + tt] : 0.f;
            l1 += fabsf(v[b]);
        }
        const float inv = l1 > 1.17549435e-38f ? 1.f / l1 : 1.f;
        const float w = cens_win[win_len - 1 - j] / wsum;
#pragma unroll
        for (int b = 0; b < CENS_MAX_BINS; ++b) {
            const float c = v[b] * inv;
            const float q = 0.25f * ((c > 0.4f) + (c > 0.2f) + (c > 0.1f) + (c > 0.05f));
            acc[b] = fmaf(w, q, acc[b]);
        }
    }
    float l2 = 0.f;
#pragma unroll
    for (int b = 0; b < CENS_MAX_BINS; ++b) l2 = fmaf(acc[b], acc[b], l2);
    l2 = sqrtf(l2);
    const float inv2 = l2 > 1.17549435e-38f ? 1.f / l2 : 1.f;
    for (int b = 0; b < n_bins; ++b) out[(size_t)b * n_frames + t] = acc[b] * inv2;
}

// Nearest-neighbour median filter (the role of librosa.decompose.nn_filter(S, aggregate=np.median, metric="cosine"),
// signal.py:131).  One workgroup per frame i: cosine similarity to every frame in fp64 (the neighbour ORDER has to match
// the float64 oracle), the k best frames outside |i-j| < width by k rounds of block-wide arg-max (ties -> lower index,
// i.e. a stable descending sort), then a per-feature median over those k frames.
// SIMS_IN_LDS: the similarity row of the frame lives in LDS (tracks up to ~16k frames = 6 min at hop 512); otherwise every
// workgroup owns one row of a caller-provided fp64 workspace [gridDim.x][n_frames] and walks frames i = blockIdx.x,
// blockIdx.x + gridDim.x, ... — full-length songs keep the reference's semantics instead of skipping the filter.
template <bool SIMS_IN_LDS>
__global__ __launch_bounds__(256) void nn_median_kernel(const float* __restrict__ ch, float* __restrict__ out, int n_bins,
                                                        int n_frames, int k, int width, double* __restrict__ sims_ws) {
    extern __shared__ __attribute__((aligned(8))) unsigned char nn_lds[];
    double* sims = SIMS_IN_LDS ? reinterpret_cast<double*>(nn_lds) : sims_ws + (size_t)blockIdx.x * n_frames;  // [n_frames]
    int* picked = reinterpret_cast<int*>(nn_lds + (SIMS_IN_LDS ? (size_t)n_frames * sizeof(double) : 0));     // [k]
    float* vals = reinterpret_cast<float*>(picked + k);                    // [n_bins][k]
    __shared__ double red_v[256];
    __shared__ int red_i[256];
    const int tid = threadIdx.x;
    for (int i = blockIdx.x; i < n_frames; i += gridDim.x) {
    __syncthreads();  // (the previous frame's vals / picked are no longer read)
    double ui[CENS_MAX_BINS];
    {
        double ni = 0.0;
        for (int b = 0; b < n_bins; ++b) ni += (double)ch[(size_t)b * n_frames + i] * (double)ch[(size_t)b * n_frames + i];
        ni = sqrt(ni);
        const double inv_i = ni > 1.17549435e-38 ? 1.0 / ni : 1.0;
        for (int b = 0; b < CENS_MAX_BINS; ++b) ui[b] = b < n_bins ? (double)ch[(size_t)b * n_frames + i] * inv_i : 0.0;
    }
    for (int j = tid; j < n_frames; j += 256) {
        double nj = 0.0, vj[CENS_MAX_BINS];
        for (int b = 0; b < n_bins; ++b) {
            vj[b] = (double)ch[(size_t)b * n_frames + j];
            nj += vj[b] * vj[b];
        }
        nj = sqrt(nj);
        const double inv_j = nj > 1.17549435e-38 ? 1.0 / nj : 1.0;
        double dot = 0.0;
        for (int b = 0; b < n_bins; ++b) dot += ui[b] * (vj[b] * inv_j);
        const int dist = j > i ? j - i : i - j;
        sims[j] = dist < width ? -INFINITY : dot;
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = tid; j < n_frames; j += 256) {
            const double v = sims[j];
            if (v > bv || (v == bv && j < bi)) bv = v, bi = j;
        }
        red_v[tid] = bv, red_i[tid] = bi;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                const double v = red_v[tid + off];
                const int j = red_i[tid + off];
                if (v > red_v[tid] || (v == red_v[tid] && j < red_i[tid])) red_v[tid] = v, red_i[tid] = j;
            }
            __syncthreads();
        }
        if (tid == 0) {
            picked[r] = red_i[0];
            sims[red_i[0]] = -INFINITY;
        }
        __syncthreads();
    }
    for (int e = tid; e < n_bins * k; e += 256) {
        const int b = e / k, m = e - b * k;
        vals[e] = ch[(size_t)b * n_frames + picked[m]];
    }
    __syncthreads();
    if (tid < n_bins) {  // insertion sort of this feature's k values, then np.median
        float* v = vals + tid * k;
        for (int a = 1; a < k; ++a) {
            const float x = v[a];
            int c = a - 1;
            while (c >= 0 && v[c] > x) v[c + 1] = v[c], --c;
            v[c + 1] = x;
        }
        out[(size_t)tid * n_frames + i] = (k & 1) ? v[k / 2] : (float)(0.5 * ((double)v[k / 2 - 1] + (double)v[k / 2]));
    }
    }
}


// Fourier-method resampling along time (scipy.signal.resample as the reference calls it, signal.py:68,152) evaluated in
// the time domain: y[i] = (1/n) sum_j R(i, j) x[j] with the Dirichlet kernel of the m = min(n, num) retained bins,
//   R = sin((2K+1) pi th) / sin(pi th),  th = i/num - j/n,  K = (m-1)/2,
// plus, for even m, the shared Nyquist bin: 2 (-1)^i cos(pi m j / n) when shortening (the +-m/2 bins of the source fold
// into one real bin), (-1)^j cos(pi n i / num) when lengthening (the source's Nyquist bin is split in half).  Phases are
// reduced in 64-bit integers before sinpi/cospi, all arithmetic is fp64: envelopes are O(n_frames) numbers, and an FFT
// library would spend seconds planning the odd transform lengths (1293 -> 900 frames) that appear here.
constexpr int RS_F = 16;  // feature columns per pass

__global__ __launch_bounds__(256) void resample_kernel(const double* __restrict__ x, int n, int64_t features,
                                                       double* __restrict__ y, int num) {
    __shared__ double part[4][RS_F];
    const int i = blockIdx.x;
    const int64_t f0 = (int64_t)blockIdx.y * RS_F;
    const int nf = (int)(features - f0 < RS_F ? features - f0 : RS_F);
    const int m = n < num ? n : num, K = (m - 1) / 2;
    const int64_t D = (int64_t)n * num;
    const bool nyquist = (m % 2 == 0);
    double acc[RS_F];
#pragma unroll
    for (int f = 0; f < RS_F; ++f) acc[f] = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        int64_t a = ((int64_t)i * n - (int64_t)j * num) % D;
        if (a < 0) a += D;
        double r = (double)(2 * K + 1);
        if (a != 0) {
            const int64_t q = ((int64_t)(2 * K + 1) * a) % (2 * D);
            r = sinpi((double)q / (double)D) / sinpi((double)a / (double)D);
        }
        if (nyquist) {
            if (num < n)
                r += ((i & 1) ? -2.0 : 2.0) * cospi((double)(((int64_t)m * j) % (2 * (int64_t)n)) / (double)n);
            else
                r += ((j & 1) ? -1.0 : 1.0) * cospi((double)(((int64_t)n * i) % (2 * (int64_t)num)) / (double)num);
        }
        const double* xr = x + (int64_t)j * features + f0;
#pragma unroll
        for (int f = 0; f < RS_F; ++f)
            if (f < nf) acc[f] = fma(r, xr[f], acc[f]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int f = 0; f < RS_F; ++f) {
        double v = acc[f];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) part[wave][f] = v;
    }
    __syncthreads();
    if (threadIdx.x < nf) {
        const int f = threadIdx.x;
        y[(int64_t)i * features + f0 + f] = ((part[0][f] + part[1][f]) + (part[2][f] + part[3][f])) / (double)n;
    }
}
}  // namespace

extern "C" int maua_temporal_fir_f32(const float* x, const float* taps, float* y, int n_frames, int64_t features,
                                     int radius, void* stream) {
    if (!x || !taps || !y || n_frames <= 0 || features <= 0 || radius < 0) return MAUA_EINVAL;
    const size_t lds = (size_t)(FIR_PAD_LO + 2 * radius + 1 + FIR_PAD_HI) * sizeof(float);
    if (lds > 64 * 1024) return MAUA_EINVAL;  // radius > MAUA_TEMPORAL_FIR_MAX_RADIUS (include/maua_hip.h)
    const int64_t bx = ceil_div64(features, 256);
    if (bx > 0x7fffffff) return MAUA_EINVAL;
    hipLaunchKernelGGL(temporal_fir_kernel, dim3((unsigned)bx, ceil_div(n_frames, FIR_TT)), dim3(256), lds,
                       (hipStream_t)stream, x, taps, y, n_frames, features, radius);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_stft_power_f32(const float* y, int64_t n_samples, const float* window, int n_fft, int hop, float* p,
                                   int n_frames, void* stream) {
    if (!y || !window || !p || n_samples <= 0 || hop <= 0 || n_frames <= 0) return MAUA_EINVAL;
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096) return MAUA_EINVAL;
    const size_t lds = (size_t)3 * n_fft * sizeof(float);
    hipLaunchKernelGGL(stft_power_kernel, dim3(n_frames), dim3(256), lds, (hipStream_t)stream, y, n_samples, window,
                       n_fft, log2n, hop, p, n_frames);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_filterbank_f32(const float* fb, const float* p, float* out, int m, int k, int n, int to_db,
                                   float amin, void* stream) {
    if (!fb || !p || !out || m <= 0 || k <= 0 || n <= 0) return MAUA_EINVAL;
    hipLaunchKernelGGL(filterbank_kernel, dim3(ceil_div(n, 256), m), dim3(256), 0, (hipStream_t)stream, fb, p, out, m, k,
                       n, to_db, amin);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_perlin3d_f32(const float* grad, float* out, int n0, int n1, int n2, int r0, int r1, int r2,
                                 void* stream) {
    if (!grad || !out || n0 <= 0 || n1 <= 0 || n2 <= 0 || r0 <= 0 || r1 <= 0 || r2 <= 0) return MAUA_EINVAL;
    if (n0 % r0 || n1 % r1 || n2 % r2) return MAUA_EINVAL;
    const int64_t total = (int64_t)n0 * n1 * n2;
    const int64_t blocks = ceil_div64(total, 256);
    hipLaunchKernelGGL(perlin3d_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, grad, out, n0, n1, n2, r0, r1, r2);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_affine_reflect_warp_f32(const float* x, const float* m, float* y, int batch, int channels, int h,
                                            int w, int pad_l, int pad_r, int pad_t, int pad_b, const float* add_noise,
                                            void* stream) {
    if (!x || !m || !y || batch <= 0 || channels <= 0 || h <= 0 || w <= 0) return MAUA_EINVAL;
    if (pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0 || channels > 65535 || batch > 65535) return MAUA_EINVAL;
    hipLaunchKernelGGL(affine_reflect_warp_kernel, dim3(ceil_div(h * w, 256), channels, batch), dim3(256), 0,
                       (hipStream_t)stream, x, m, y, channels, h, w, pad_l, pad_r, pad_t, pad_b, add_noise,
                       (const int*)nullptr, (const int*)nullptr, (const maua_frame_source_t*)nullptr);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_affine_reflect_warp_mapped_f32(const float* x, const float* m, float* y, int batch, int channels, int h,
                                                   int w, int pad_l, int pad_r, int pad_t, int pad_b,
                                                   const float* add_noise, const int* xmap, const int* ymap,
                                                   const maua_frame_source_t* src, void* stream) {
    if (!x || !m || !y || batch <= 0 || channels <= 0 || h <= 0 || w <= 0) return MAUA_EINVAL;
    if (pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0 || channels > 65535 || batch > 65535) return MAUA_EINVAL;
    hipLaunchKernelGGL(affine_reflect_warp_kernel, dim3(ceil_div(h * w, 256), channels, batch), dim3(256), 0,
                       (hipStream_t)stream, x, m, y, channels, h, w, pad_l, pad_r, pad_t, pad_b, add_noise, xmap, ymap, src);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_stft_complex_f32(const float* y, int64_t n_samples, const float* window, int n_fft, int hop, float* out_re,
                                     float* out_im, int n_frames, void* stream) {
    if (!y || !window || !out_re || !out_im || n_samples <= 0 || hop <= 0 || n_frames <= 0) return MAUA_EINVAL;
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096) return MAUA_EINVAL;
    hipLaunchKernelGGL(stft_complex_kernel, dim3(n_frames), dim3(256), (size_t)3 * n_fft * sizeof(float), (hipStream_t)stream, y,
                       n_samples, window, n_fft, log2n, hop, out_re, out_im, n_frames);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_istft_f32(const float* in_re, const float* in_im, const float* window, int n_fft, int hop, int n_frames,
                              float* frames_ws, float* y, int64_t n_samples, void* stream) {
    if (!in_re || !in_im || !window || !frames_ws || !y || hop <= 0 || n_frames <= 0 || n_samples <= 0) return MAUA_EINVAL;
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096) return MAUA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(istft_frames_kernel, dim3(n_frames), dim3(256), (size_t)3 * n_fft * sizeof(float), st, in_re, in_im, window,
                       n_fft, log2n, n_frames, frames_ws);
    MAUA_LAUNCH_CHECK();
    const int64_t blocks = ceil_div64(n_samples, 256);
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, frames_ws, window, n_fft,
                       hop, n_frames, y, n_samples);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_median_filter_f32(const float* x, float* y, int rows, int cols, int size, int axis, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || (axis != 0 && axis != 1)) return MAUA_EINVAL;
    const int64_t blocks = ceil_div64((int64_t)rows * cols, 256);
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192));
    hipStream_t st = (hipStream_t)stream;
    switch (size) {
        case 31: hipLaunchKernelGGL(median_filter_kernel<31>, grid, dim3(256), 0, st, x, y, rows, cols, axis); break;
        case 17: hipLaunchKernelGGL(median_filter_kernel<17>, grid, dim3(256), 0, st, x, y, rows, cols, axis); break;
        case 9: hipLaunchKernelGGL(median_filter_kernel<9>, grid, dim3(256), 0, st, x, y, rows, cols, axis); break;
        case 5: hipLaunchKernelGGL(median_filter_kernel<5>, grid, dim3(256), 0, st, x, y, rows, cols, axis); break;
        case 3: hipLaunchKernelGGL(median_filter_kernel<3>, grid, dim3(256), 0, st, x, y, rows, cols, axis); break;
        default: return MAUA_ENOSYS;
    }
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_softmask_apply_f32(const float* re, const float* im, const float* x, const float* x_ref, float margin,
                                       float power, int split_zeros, float* out_re, float* out_im, int64_t n, void* stream) {
    if (!re || !im || !x || !x_ref || !out_re || !out_im || n <= 0) return MAUA_EINVAL;
    const int64_t blocks = ceil_div64(n, 256);
    hipLaunchKernelGGL(softmask_apply_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, re,
                       im, x, x_ref, margin, power, split_zeros, out_re, out_im, n);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_chroma_cens_f32(const float* ch, float* out, int n_bins, int n_frames, int win_len, void* stream) {
    if (!ch || !out || n_bins <= 0 || n_bins > CENS_MAX_BINS || n_frames <= 0 || win_len <= 0 || win_len > 255 || !(win_len & 1))
        return MAUA_EINVAL;
    hipLaunchKernelGGL(chroma_cens_kernel, dim3(ceil_div(n_frames, 256)), dim3(256), (size_t)win_len * sizeof(float),
                       (hipStream_t)stream, ch, out, n_bins, n_frames, win_len);
    MAUA_LAUNCH_CHECK();
    return 0;
}

// Similarity rows live in LDS while they fit; beyond (> ~16k frames) the caller supplies a workspace of
// maua_nn_median_ws_doubles() doubles.  A non-NULL `ws` selects the workspace path at ANY size (min(n_frames, 1024) rows,
// frames strided over the grid), so both paths of the kernel can be compared on one input.
static inline int nn_ws_rows(int n_frames) { return n_frames < 1024 ? n_frames : 1024; }

extern "C" int64_t maua_nn_median_ws_doubles(int n_bins, int n_frames, int k) {
    const size_t lds = (size_t)n_frames * sizeof(double) + (size_t)k * sizeof(int) + (size_t)n_bins * k * sizeof(float);
    if (lds <= 150 * 1024) return 0;  // the similarity row fits LDS
    return (int64_t)nn_ws_rows(n_frames) * n_frames;
}

extern "C" int maua_nn_median_f32(const float* ch, float* out, int n_bins, int n_frames, int k, int width, double* ws, void* stream) {
    if (!ch || !out || n_bins <= 0 || n_bins > CENS_MAX_BINS || n_frames <= 1 || k <= 0 || k >= n_frames || k > 2048 || width < 1)
        return MAUA_EINVAL;
    const size_t small = (size_t)k * sizeof(int) + (size_t)n_bins * k * sizeof(float);
    const size_t lds = (size_t)n_frames * sizeof(double) + small;
    static unsigned long long lds_ok = 0;  // per launcher: devices on which the attribute has been set (common.h)
    static unsigned long long lds_ok1 = 0;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(nn_median_kernel<true>), &lds_ok, 150 * 1024)) return rc;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(nn_median_kernel<false>), &lds_ok1, 150 * 1024)) return rc;
    if (!ws) {
        if (lds > 150 * 1024) return MAUA_EINVAL;  // long track: the caller must supply the workspace
        hipLaunchKernelGGL(nn_median_kernel<true>, dim3(n_frames), dim3(256), lds, (hipStream_t)stream, ch, out, n_bins, n_frames, k,
                           width, (double*)nullptr);
    } else {  // similarity rows in the caller's workspace, frames strided over the grid
        if (small > 150 * 1024) return MAUA_EINVAL;
        hipLaunchKernelGGL(nn_median_kernel<false>, dim3(nn_ws_rows(n_frames)), dim3(256), small, (hipStream_t)stream, ch, out, n_bins,
                           n_frames, k, width, ws);
    }
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_cqt_mag_f32(const float* y, int64_t n_samples, const float* freqs, const int* lengths, int n_bins,
                                int hop, float sr, float* out, int n_frames, void* stream) {
    if (!y || !freqs || !lengths || !out || n_samples <= 0 || n_bins <= 0 || n_bins > 65535 || hop <= 0 || sr <= 0.f ||
        n_frames <= 0)
        return MAUA_EINVAL;
    hipLaunchKernelGGL(cqt_mag_kernel, dim3(n_frames, n_bins), dim3(256), 0, (hipStream_t)stream, y, n_samples, freqs,
                       lengths, hop, sr, out, n_frames);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_resample_f64(const double* x, int n, int64_t features, double* y, int num, void* stream) {
    if (!x || !y || n <= 0 || num <= 0 || features <= 0) return MAUA_EINVAL;
    if ((int64_t)n * num > (1ll << 40)) return MAUA_EINVAL;  // (2K+1) * phase must stay inside int64
    hipStream_t st = (hipStream_t)stream;
    if (n == num) {
        hipError_t e = hipMemcpyAsync(y, x, (size_t)n * features * sizeof(double), hipMemcpyDeviceToDevice, st);
        return e == hipSuccess ? 0 : (int)e;
    }
    const int64_t fy = ceil_div64(features, RS_F);
    if (fy > 65535) return MAUA_EINVAL;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)num, (unsigned)fy), dim3(256), 0, st, x, n, features, y, num);
    MAUA_LAUNCH_CHECK();
    return 0;
}

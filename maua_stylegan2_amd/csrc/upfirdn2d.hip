// upfirdn2d for gfx950 — up-sample / FIR / down-sample of NCHW planes.
//
// Behavioural contract: /root/reference/op/upfirdn2d.py:145-200 and op/upfirdn2d_kernel.cu (semantics only:
// zero-stuff by `up`, pad (negative pad crops), TRUE convolution with the tap matrix, decimate by `down`).
// This is a new design for CDNA4, not a translation of the reference kernel:
//
//  * fir_tile_kernel (up = down = 1, the Blur-after-up-conv hot case, 96 % of the path's upfirdn2d bytes —
//    BASELINE.md §3.2): HBM-bound streaming.  One 256-thread workgroup owns a (WY*32) x (WX*64) output tile of one
//    plane.  The input tile (+ KH-1 / KW-1 halo) is fetched with dense, lane-consecutive dword BUFFER loads — rows of the
//    (2H+1)-wide up-conv output are only 4-byte aligned, so 16-byte vector loads are not available; the row offset of an
//    access is a scalar, the column offset one lane register, out-of-range columns come from the descriptor's range check —
//    ALL issued before the first LDS write (~36 loads/thread, ~36 KB in flight per workgroup, 4 workgroups per CU).  After
//    one barrier each wave walks down its 64 columns: lane = column, KW conflict-free ds_read_b32 per input row,
//    KH rotating accumulators, one dense 256-byte buffer store per wave per output row.  Optional fused tail
//    (demod gain, noise, bias, leaky-ReLU*sqrt2) = the rest of StyledConv.forward: the lane's 32 noise values are fetched with
//    the tile.  Logical tile order is (plane, tile_x, tile_y) with tile_y fastest and an XCD-aware remap so that vertically
//    adjacent tiles (which share KH-1 halo rows) are served by the same L2; with the tail the channel is fastest (planes that
//    share a noise tile run back-to-back).
//  * fir_up2_kernel (up = 2, down = 1, taps <= 4 x 4: the Upsample module as a standalone op): polyphase, store-bound; see the kernel.
//  * fir_down2_kernel (up = 1, down = 2, taps <= 4 x 4: the Downsample module / the backward of Upsample; off the inference path): load-bound.
//  * fir_generic_kernel: any up/down/pad/minor, one thread per output, polyphase tap skipping.
#include "common.h"

#include <hip/hip_fp16.h>

namespace {

struct FirTail {
    const float* gain;    // [planes] or null
    const float* noise;   // [B or 1, out_h, out_w] or null
    const float* noise_w; // [1]
    const float* bias;    // [channels]
    int64_t noise_batch_stride;
    int channels;
    const maua_frame_source_t* src;  // when set: noise / noise_batch_stride come from src->noise[noise_slot] at frame src->frame0
    int noise_slot;
    int plane_major;  // block order: 1 = plane by plane (sequential HBM rows), 0 = channel fastest (planes sharing a noise tile back to back)
    // the style fold (round 6): the stored map is multiplied by post_s[b * post_stride + c] — the styles of the modulated convolution that
    // consumes it (models/stylegan2.py:220-221: conv(W, x * s) with the multiply moved here, once per element)
    const float* post_s;
    int post_stride;
};

// Output rows per wave strip (TH).  Plain: 24 (27 KB of LDS: five workgroups per CU instead of four — 5.51 vs 5.33 TB/s on [8,32,1025,1025]
// with 32 rows, 16 rows 5.46).  With the tail 32 on the 1024-row maps and 16 below: a 16-row strip needs 57 registers
// instead of 106 (19 staged rows + 16 noise values in flight instead of 35 + 32), the launches of the 8^2..512^2 layers measure
// 0.027 -> 0.017 ms (8^2..64^2), 0.057 -> 0.054 (128^2), 0.152 -> 0.150 (256^2), ~0.23 (512^2: unchanged); the 1024^2 launch is 7 % slower with
// it (0.427 -> 0.457: more halo rows per output row) and keeps 32.
#ifndef MAUA_FIR_ROWS_TAIL_SMALL
#define MAUA_FIR_ROWS_TAIL_SMALL 16
#endif
constexpr int FIR_ROWS_PLAIN = 24, FIR_ROWS_TAIL = 32, FIR_ROWS_TAIL_SMALL = MAUA_FIR_ROWS_TAIL_SMALL;
constexpr int FIR_TAIL_SMALL_MAX_H = 512;
#ifndef MAUA_FIR_UP2_ROWS
#define MAUA_FIR_UP2_ROWS 16
#endif
#ifndef MAUA_FIR_DOWN2_ROWS
#define MAUA_FIR_DOWN2_ROWS 8
#endif
constexpr int FIR_DOWN2_ROWS = MAUA_FIR_DOWN2_ROWS;  // output rows per wave strip of fir_down2_kernel
constexpr int FIR_UP2_ROWS = MAUA_FIR_UP2_ROWS;  // output rows per wave strip of fir_up2_kernel ([8,32,512,512] -> 1024^2: 16 rows 0.239 ms, 32 0.267, 64 0.257)

#if defined(__HIP_DEVICE_COMPILE__)
#define MAUA_DEVICE_PASS 1
#endif
constexpr unsigned FIR_OOB = 0x80000000u;  // a voffset beyond every descriptor range: loads return 0, stores are dropped

// up = down = 1 FIR of one (WY*32) x (WX*64) output tile per 256-thread workgroup (the Blur after an up-convolution and its fused
// tail).  Round 4 form: every global access is a raw BUFFER instruction whose row offset is a SCALAR (soffset) and whose column
// offset is one lane register computed once — a wave instruction reads / writes 64 consecutive floats of one row, and which row
// that is is uniform over the wave.  The round-3 kernels (fir_strip_kernel / fir_tile_kernel, flat 64-bit addresses + predicates)
// spent 1400-1520 VALU instructions per wave on 512 useful FMAs (2 VALU of 64-bit address arithmetic per load, a compare + branch
// per access, a zero-initialising move per staged value): 4 waves per SIMD x 4 cycles made the VALU 70-80 % busy at the rates
// they reached, i.e. they were as much issue-bound as HBM-bound.  Here out-of-range columns are an out-of-range voffset (the
// descriptor returns 0 / drops the store), out-of-range rows a scalar branch.
template <int KH, int KW, int WX, bool TAIL, int TH>
__global__ __launch_bounds__(256) void fir_tile_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                       float* __restrict__ y, int planes, int in_h, int in_w,
                                                       int out_h, int out_w, int pad_x0, int pad_y0, int tiles_x,
                                                       int tiles_y, FirTail tail) {
    constexpr int WY = 4 / WX;
    constexpr int TW = 64 * WX;
    constexpr int RH = WY * TH + KH - 1;  // staged rows
    constexpr int RW = TW + KW - 1;       // staged cols
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblocks = gridDim.x;
    int t = xcd_remap(blockIdx.x, nblocks);
    int plane, tile_x, tile_y;
    if (TAIL && !tail.plane_major) {  // channel fastest: the planes that share one noise tile run back-to-back on one XCD (L2 hits)
        const int c = t % tail.channels;
        t /= tail.channels;
        tile_y = t % tiles_y;
        t /= tiles_y;
        tile_x = t % tiles_x;
        plane = (t / tiles_x) * tail.channels + c;
    } else {
        const int tiles_per_plane = tiles_x * tiles_y;
        plane = t / tiles_per_plane;
        t -= plane * tiles_per_plane;
        tile_x = t / tiles_y;
        tile_y = t - tile_x * tiles_y;
    }
    const int oy0 = tile_y * (WY * TH);
    const int ox0 = tile_x * TW;

    // flipped taps -> SGPRs (uniform loads)
    float kf[KH][KW];
#pragma unroll
    for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int j = 0; j < KW; ++j) kf[i][j] = k[(KH - 1 - i) * KW + (KW - 1 - j)];

    const int iy0 = oy0 - pad_y0;
    const int ix0 = ox0 - pad_x0;
    // staging map: wave w owns columns (w % WX) * 64 + lane of the rows w / WX, w / WX + WY, ...; the KW-1 halo columns of the tile
    // go to the first (KW-1) * RH threads
    constexpr int NR = (RH + WY - 1) / WY;
    constexpr int NH = ((KW - 1) * RH + 255) / 256;
    const int tx = (wave % WX) * 64 + lane;
    const int ty = wave / WX;  // (scalar)
    const int ixm = ix0 + tx;
    const unsigned in_voff = (ixm >= 0 && ixm < in_w) ? (unsigned)ixm * 4u : FIR_OOB;
    const unsigned in_row_bytes = (unsigned)in_w * 4u, out_row_bytes = (unsigned)out_w * 4u;
#ifdef MAUA_DEVICE_PASS
    // one descriptor per plane: offsets stay below 2^31 (checked by the launcher), anything past the plane's end is out of range
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x) + (size_t)plane * in_h * in_w, 0, (int)((unsigned)in_h * in_row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(y + (size_t)plane * out_h * out_w, 0, (int)((unsigned)out_h * out_row_bytes), 0x00020000);
#endif
    float v[NR], vh[NH];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int rr = ty + i * WY;  // (scalar)
        const int iy = iy0 + rr;
        (void)iy;
        v[i] = 0.f;
#ifdef MAUA_DEVICE_PASS
        if (rr < RH && iy >= 0 && iy < in_h)  // uniform over the wave: a scalar branch
            v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, in_voff, (unsigned)iy * in_row_bytes, 0));
#endif
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int e = tid + h * 256;
        const int hr = e / (KW - 1), hc = TW + (e - hr * (KW - 1));
        const int iy = iy0 + hr, ix = ix0 + hc;
        const bool ok = e < (KW - 1) * RH && iy >= 0 && iy < in_h && ix >= 0 && ix < in_w;
        (void)ok;
        vh[h] = 0.f;
#ifdef MAUA_DEVICE_PASS
        vh[h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                              x_rsrc, ok ? ((unsigned)iy * (unsigned)in_w + (unsigned)ix) * 4u : FIR_OOB, 0, 0));
#endif
    }
    // compute map: wave (wx, wy), lane = column
    const int wx = wave % WX, wy = wave / WX;
    const int col = wx * 64 + lane;
    const int ox = ox0 + col;
    const int row0 = wy * TH;
    const unsigned out_voff = ox < out_w ? (unsigned)ox * 4u : FIR_OOB;

    // tail operands: this lane's 32 noise values are fetched now, in flight together with the input tile
    float g = 1.f, nw = 0.f, bs = 0.f, post = 1.f;
    float nzv[TAIL ? TH : 1];
    if (TAIL) {
        const int b = plane / tail.channels;
        const int c = plane - b * tail.channels;
        // leaky ReLU * sqrt2 as max(t, 0.2 t) on pre-scaled operands: t = sqrt2 (g v + nw noise + bias)
        g = 1.41421356237309515f * (tail.gain ? tail.gain[plane] : 1.f);
        bs = tail.bias ? tail.bias[c] * 1.41421356237309515f : 0.f;
        if (tail.post_s) post = tail.post_s[(size_t)b * tail.post_stride + c];
#pragma unroll
        for (int o = 0; o < TH; ++o) nzv[o] = 0.f;
        const float* noise = tail.noise;
        int64_t nstride = tail.noise_batch_stride;
        if (tail.src) {  // uniform scalar loads
            nstride = tail.src->noise_stride[tail.noise_slot];
            noise = tail.src->noise[tail.noise_slot];
            if (noise) noise += (int64_t)tail.src->frame0 * nstride;
        }
        if (noise) {
            nw = tail.noise_w[0] * 1.41421356237309515f;
#ifdef MAUA_DEVICE_PASS
            const __amdgpu_buffer_rsrc_t n_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(noise) + (size_t)b * nstride, 0, (int)((unsigned)out_h * out_row_bytes), 0x00020000);
#pragma unroll
            for (int o = 0; o < TH; ++o) {
                const int oy = oy0 + row0 + o;  // (scalar)
                if (oy < out_h)
                    nzv[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(n_rsrc, out_voff, (unsigned)oy * out_row_bytes, 0));
            }
#endif
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int rr = ty + i * WY;
        if (rr < RH) lds[rr * RW + tx] = v[i];
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int e = tid + h * 256;
        const int hr = e / (KW - 1), hc = TW + (e - hr * (KW - 1));
        if (e < (KW - 1) * RH) lds[hr * RW + hc] = vh[h];
    }
    __syncthreads();

    float acc[KH];
#pragma unroll
    for (int i = 0; i < KH; ++i) acc[i] = 0.f;
    // (the row pointer is advanced, not indexed: ds_read2_b32 takes 8-bit dword offsets, and r * RW as an immediate made the compiler
    // rebuild the address with a VALU add for every pair of reads)
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float* lrow = (lds_float*)lds + (row0 * RW + col);
#pragma unroll
    for (int r = 0; r < TH + KH - 1; ++r) {
        float in[KW];
#pragma unroll
        for (int j = 0; j < KW; ++j) in[j] = lrow[j];
        lrow += RW;
        asm volatile("" : "+v"(lrow));  // (one VALU add of the running LDS address per row instead of re-derived addresses)
#pragma unroll
        for (int i = 0; i < KH; ++i) {
            const int o = r - i;  // output row (within the wave's strip) this input row feeds through tap row i
            if (o >= 0 && o < TH) {
#pragma unroll
                for (int j = 0; j < KW; ++j) acc[o % KH] = (i == 0 && j == 0) ? kf[i][j] * in[j] : fmaf(kf[i][j], in[j], acc[o % KH]);
            }
        }
        const int o_done = r - (KH - 1);
        if (o_done >= 0) {
            const int oy = oy0 + row0 + o_done;  // (scalar)
            float val = acc[o_done % KH];
            if (TAIL) {
                const float tt = fmaf(val, g, fmaf(nw, nzv[TAIL ? o_done : 0], bs));
                val = fmaxf(tt, 0.2f * tt) * post;
            }
#ifdef MAUA_DEVICE_PASS
            if (oy < out_h)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), y_rsrc, out_voff, (unsigned)oy * out_row_bytes, 0);
#endif
        }
    }
}

// up = 2, down = 1 (the Upsample module, reference models/stylegan2.py:51-67 -> op/upfirdn2d.py:145-200 with up = 2, pad = (2, 1); the
// reference's own tiled specialisation is op/upfirdn2d_kernel.cu:313-359, mode 3).  HBM-bound on its STORES: 4 output floats per input
// float.  Polyphase: output o reads canvas positions o - pad0 + j, of which only the even ones hold data, so with c = o - pad0 and
// j0 = c & 1 it has two taps per axis, j0 and j0 + 2, on inputs (c + j0) / 2 and the next.  A lane owns the output column PAIR
// (2 lane, 2 lane + 1) of its wave's 128-column strip: the pair's tap phases are uniform over the wave (weights in SGPRs), its inputs
// are three consecutive staged columns, and two consecutive output rows share one pair of input rows — per input row 3 LDS reads and
// two 8-byte stores (512 B per wave instruction), no per-element index arithmetic.  A workgroup is WX x (4 / WX) such strips of TH rows;
// the input tile (TH / 2 + 2 rows per strip) is staged with scalar-row buffer loads, out-of-range rows / columns read 0.
// Taps are zero-extended to 4 x 4: KH, KW <= 4.
template <int WX, int TH>
__global__ __launch_bounds__(256) void fir_up2_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ y,
                                                      int in_h, int in_w, int out_h, int out_w, int kh, int kw, int pad_x0, int pad_y0,
                                                      int tiles_x, int tiles_y) {
    constexpr int WY = 4 / WX;
    constexpr int RW = 64 * WX + 2;           // staged columns
    constexpr int RH = WY * (TH / 2) + 2;     // staged rows
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_per_plane = tiles_x * tiles_y;
    const int plane = t / tiles_per_plane;
    t -= plane * tiles_per_plane;
    const int tile_x = t / tiles_y, tile_y = t - tile_x * tiles_y;
    const int oy0 = tile_y * (WY * TH), ox0 = tile_x * (128 * WX);

    const int jx = (ox0 - pad_x0) & 1;                  // tap phase of the even column of a pair; the odd column has 1 - jx
    const int ix0 = (ox0 - pad_x0 + jx) >> 1;           // first staged input column: the even column's first input at lane 0
    const int dx = 1 - jx;                              // the odd column's inputs start dx columns later
    const int iy0 = (oy0 - pad_y0) >> 1;                // first staged input row
    // flipped, zero-extended taps by column phase (uniform loads): ka[i][t] = tap (i, jx + 2 t) of the even column, kb[i][t] = tap
    // (i, 1 - jx + 2 t) of the odd one
    auto tap = [&](int i, int j) { return (i < kh && j < kw) ? k[(kh - 1 - i) * kw + (kw - 1 - j)] : 0.f; };
    float ka[4][2], kb[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) ka[i][tt] = tap(i, jx + 2 * tt), kb[i][tt] = tap(i, 1 - jx + 2 * tt);
    const unsigned in_row_bytes = (unsigned)in_w * 4u, out_row_bytes = (unsigned)out_w * 4u;
#ifdef MAUA_DEVICE_PASS
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x) + (size_t)plane * in_h * in_w, 0, (int)((unsigned)in_h * in_row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(y + (size_t)plane * out_h * out_w, 0, (int)((unsigned)out_h * out_row_bytes), 0x00020000);
#endif
    // staging: wave w takes rows w, w + 4, ...; a row is WX dense 64-float segments + 2 more columns (lanes 0, 1 of one more load)
    constexpr int NR = (RH + 3) / 4;
    unsigned voff[WX + 1];
#pragma unroll
    for (int sgm = 0; sgm <= WX; ++sgm) {
        const int ix = ix0 + sgm * 64 + lane;
        voff[sgm] = (ix >= 0 && ix < in_w && (sgm < WX || lane < 2)) ? (unsigned)ix * 4u : FIR_OOB;
    }
    float v[NR][WX + 1];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int rr = wave + 4 * i;  // (scalar)
        const int iy = iy0 + rr;
        (void)iy;
#pragma unroll
        for (int sgm = 0; sgm <= WX; ++sgm) {
            v[i][sgm] = 0.f;
#ifdef MAUA_DEVICE_PASS
            if (rr < RH && iy >= 0 && iy < in_h)
                v[i][sgm] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, voff[sgm], (unsigned)iy * in_row_bytes, 0));
#endif
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int rr = wave + 4 * i;
        if (rr < RH) {
#pragma unroll
            for (int sgm = 0; sgm < WX; ++sgm) lds[rr * RW + sgm * 64 + lane] = v[i][sgm];
            if (lane < 2) lds[rr * RW + WX * 64 + lane] = v[i][WX];
        }
    }
    __syncthreads();

    const int wx = wave % WX, wy = wave / WX;
    const int strip_lo = oy0 + wy * TH;
    const int ox = ox0 + wx * 128 + 2 * lane;
    // 8-byte stores; a lone last column (odd out_w) goes out as 4 bytes
    const unsigned out_voff = ox < out_w ? (unsigned)ox * 4u : FIR_OOB;
    const bool pair_ok = ox + 1 < out_w;
    typedef __attribute__((address_space(3))) float lds_float;
    const lds_float* lrow = (const lds_float*)lds + ((wy * (TH / 2)) * RW + wx * 64 + lane);
    float p0 = lrow[0], p1 = lrow[1], p2 = lrow[2];   // input row ra (three consecutive columns)
    const int ra_lo = (strip_lo - pad_y0) >> 1;        // == iy0 + wy * TH / 2 (TH, oy0 even)
#pragma unroll
    for (int r = 0; r <= TH / 2; ++r) {
        lrow += RW;
        const float q0 = lrow[0], q1 = lrow[1], q2 = lrow[2];  // input row ra + 1
        const float pb0 = dx ? p1 : p0, pb1 = dx ? p2 : p1, qb0 = dx ? q1 : q0, qb1 = dx ? q2 : q1;
        // the two output rows that read input rows (ra, ra + 1): 2 ra - 1 + pad_y0 with row taps (1, 3), 2 ra + pad_y0 with (0, 2)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int oy = 2 * (ra_lo + r) - 1 + e + pad_y0;  // (scalar)
            const int i0 = 1 - e;
            if (oy >= strip_lo && oy < strip_lo + TH && oy < out_h) {
                const float a = fmaf(ka[i0][0], p0, fmaf(ka[i0][1], p1, fmaf(ka[i0 + 2][0], q0, ka[i0 + 2][1] * q1)));
                const float b = fmaf(kb[i0][0], pb0, fmaf(kb[i0][1], pb1, fmaf(kb[i0 + 2][0], qb0, kb[i0 + 2][1] * qb1)));
                (void)a, (void)b;
#ifdef MAUA_DEVICE_PASS
                if (pair_ok) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)}, y_rsrc, out_voff,
                                                          (unsigned)oy * out_row_bytes, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), y_rsrc, out_voff, (unsigned)oy * out_row_bytes, 0);
                }
#endif
            }
        }
        p0 = q0, p1 = q1, p2 = q2;
    }
}

// up = 1, down = 2 (the Downsample module and the backward of Upsample; the reference tiles it as op/upfirdn2d_kernel.cu:313-359 modes 5 / 6; not
// on the inference path).  HBM-bound on its LOADS: 4 input floats per output float.  out[oy][ox] = sum kf[i][j] x[2 oy + i - pad_y0][2 ox + j - pad_x0].
// A lane owns one output column: its four input columns 2 c .. 2 c + 3 are two conflict-free ds_read_b64, two new input rows per output row roll
// through a 4 x 4 register window; the input tile (2 TH + 2 rows of 128 WX + 2 columns per strip row) is staged with scalar-row dense loads.
// Taps are zero-extended to 4 x 4: KH, KW <= 4.
template <int WX, int TH>
__global__ __launch_bounds__(256) void fir_down2_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ y,
                                                        int in_h, int in_w, int out_h, int out_w, int kh, int kw, int pad_x0, int pad_y0,
                                                        int tiles_x, int tiles_y) {
    constexpr int WY = 4 / WX;
    constexpr int RW = 128 * WX + 2;        // staged columns (even: rows stay 8-byte aligned)
    constexpr int RH = 2 * WY * TH + 2;     // staged rows
    constexpr int NSEG = 2 * WX + 1;        // dense 64-float loads per staged row (the last one: 2 columns)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_per_plane = tiles_x * tiles_y;
    const int plane = t / tiles_per_plane;
    t -= plane * tiles_per_plane;
    const int tile_x = t / tiles_y, tile_y = t - tile_x * tiles_y;
    const int oy0 = tile_y * (WY * TH), ox0 = tile_x * (64 * WX);
    const int iy0 = 2 * oy0 - pad_y0, ix0 = 2 * ox0 - pad_x0;

    auto tap = [&](int i, int j) { return (i < kh && j < kw) ? k[(kh - 1 - i) * kw + (kw - 1 - j)] : 0.f; };  // flipped, zero-extended (uniform loads)
    float kf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[i][j] = tap(i, j);

    const unsigned in_row_bytes = (unsigned)in_w * 4u, out_row_bytes = (unsigned)out_w * 4u;
#ifdef MAUA_DEVICE_PASS
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x) + (size_t)plane * in_h * in_w, 0, (int)((unsigned)in_h * in_row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(y + (size_t)plane * out_h * out_w, 0, (int)((unsigned)out_h * out_row_bytes), 0x00020000);
#endif
    unsigned voff[NSEG];
#pragma unroll
    for (int sgm = 0; sgm < NSEG; ++sgm) {
        const int ix = ix0 + sgm * 64 + lane;
        voff[sgm] = (ix >= 0 && ix < in_w && (sgm < 2 * WX || lane < 2)) ? (unsigned)ix * 4u : FIR_OOB;
    }
    // staging in two halves (rows wave, wave + 4, ...): bounds the loads in flight per lane
    constexpr int NR = (RH + 3) / 4;
    constexpr int HALF = (NR + 1) / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float v[HALF][NSEG];
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const int rr = wave + 4 * (half * HALF + i);  // (scalar)
            const int iy = iy0 + rr;
            (void)iy;
#pragma unroll
            for (int sgm = 0; sgm < NSEG; ++sgm) {
                v[i][sgm] = 0.f;
#ifdef MAUA_DEVICE_PASS
                if (rr < RH && iy >= 0 && iy < in_h)
                    v[i][sgm] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, voff[sgm], (unsigned)iy * in_row_bytes, 0));
#endif
            }
        }
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const int rr = wave + 4 * (half * HALF + i);
            if (rr < RH) {
#pragma unroll
                for (int sgm = 0; sgm < 2 * WX; ++sgm) lds[rr * RW + sgm * 64 + lane] = v[i][sgm];
                if (lane < 2) lds[rr * RW + 2 * WX * 64 + lane] = v[i][2 * WX];
            }
        }
    }
    __syncthreads();

    const int wx = wave % WX, wy = wave / WX;
    const int strip_lo = oy0 + wy * TH;
    const int ox = ox0 + wx * 64 + lane;
    const unsigned out_voff = ox < out_w ? (unsigned)ox * 4u : FIR_OOB;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
    const lds_f32x2* lrow = (const lds_f32x2*)((const __attribute__((address_space(3))) float*)lds + ((2 * wy * TH) * RW + wx * 128 + 2 * lane));
    float w[4][4];  // rolling window: input rows 2 o .. 2 o + 3, the lane's four columns
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const f32x2 a = lrow[0], b = lrow[1];
        w[r + 2][0] = a.x, w[r + 2][1] = a.y, w[r + 2][2] = b.x, w[r + 2][3] = b.y;
        lrow += RW / 2;
    }
#pragma unroll
    for (int o = 0; o < TH; ++o) {
#pragma unroll
        for (int c = 0; c < 4; ++c) w[0][c] = w[2][c], w[1][c] = w[3][c];
#pragma unroll
        for (int r = 2; r < 4; ++r) {
            const f32x2 a = lrow[0], b = lrow[1];
            w[r][0] = a.x, w[r][1] = a.y, w[r][2] = b.x, w[r][3] = b.y;
            lrow += RW / 2;
        }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = fmaf(kf[i][j], w[i][j], acc);
        const int oy = strip_lo + o;  // (scalar)
        (void)oy, (void)acc;
#ifdef MAUA_DEVICE_PASS
        if (oy < out_h) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc), y_rsrc, out_voff, (unsigned)oy * out_row_bytes, 0);
#endif
    }
}

__global__ __launch_bounds__(256) void fir_generic_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                          float* __restrict__ y, int major, int in_h, int in_w,
                                                          int minor, int kh, int kw, int up_x, int up_y, int down_x,
                                                          int down_y, int pad_x0, int pad_y0, int out_h, int out_w,
                                                          int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t rest = idx;
        const int mi = (int)(rest % minor);
        rest /= minor;
        const int ox = (int)(rest % out_w);
        rest /= out_w;
        const int oy = (int)(rest % out_h);
        const int64_t m = rest / out_h;
        float acc = 0.f;
        for (int i = 0; i < kh; ++i) {
            const int cy = oy * down_y + i - pad_y0;  // row in the zero-stuffed (unpadded) canvas
            if (cy < 0 || cy % up_y) continue;
            const int iy = cy / up_y;
            if (iy >= in_h) continue;
            for (int j = 0; j < kw; ++j) {
                const int cx = ox * down_x + j - pad_x0;
                if (cx < 0 || cx % up_x) continue;
                const int ix = cx / up_x;
                if (ix >= in_w) continue;
                acc = fmaf(k[(kh - 1 - i) * kw + (kw - 1 - j)], x[((m * in_h + iy) * in_w + ix) * minor + mi], acc);
            }
        }
        y[idx] = acc;
    }
}

// half / double instantiations of the reference's dtype dispatch (op/upfirdn2d_kernel.cu:313-359): the generic gather in the
// tensor's type (double) or with fp32 accumulation (half); taps come in the tensor's dtype like the reference's `kernel`.
template <typename T, typename A>
__global__ __launch_bounds__(256) void fir_generic_typed_kernel(const T* __restrict__ x, const T* __restrict__ k, T* __restrict__ y,
                                                                int in_h, int in_w, int minor, int kh, int kw, int up_x, int up_y,
                                                                int down_x, int down_y, int pad_x0, int pad_y0, int out_h,
                                                                int out_w, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t rest = idx;
        const int mi = (int)(rest % minor);
        rest /= minor;
        const int ox = (int)(rest % out_w);
        rest /= out_w;
        const int oy = (int)(rest % out_h);
        const int64_t m = rest / out_h;
        A acc = (A)0;
        for (int i = 0; i < kh; ++i) {
            const int cy = oy * down_y + i - pad_y0;
            if (cy < 0 || cy % up_y) continue;
            const int iy = cy / up_y;
            if (iy >= in_h) continue;
            for (int j = 0; j < kw; ++j) {
                const int cx = ox * down_x + j - pad_x0;
                if (cx < 0 || cx % up_x) continue;
                const int ix = cx / up_x;
                if (ix >= in_w) continue;
                acc += (A)k[(kh - 1 - i) * kw + (kw - 1 - j)] * (A)x[((m * in_h + iy) * in_w + ix) * minor + mi];
            }
        }
        y[idx] = (T)acc;
    }
}

template <typename T, typename A>
int launch_fir_typed(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x,
                     int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
    if (!x || !k || !y || major <= 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0 || up_x <= 0 || up_y <= 0 ||
        down_x <= 0 || down_y <= 0)
        return MAUA_EINVAL;
    const int num_h = in_h * up_y + pad_y0 + pad_y1 - kh, num_w = in_w * up_x + pad_x0 + pad_x1 - kw;
    if (num_h < 0 || num_w < 0) return MAUA_EINVAL;
    const int out_h = num_h / down_y + 1, out_w = num_w / down_x + 1;
    const int64_t total = (int64_t)major * out_h * out_w * minor;
    const int64_t blocks = ceil_div64(total, 256);
    hipLaunchKernelGGL((fir_generic_typed_kernel<T, A>), dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)x, (const T*)k, (T*)y, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y,
                       pad_x0, pad_y0, out_h, out_w, total);
    MAUA_LAUNCH_CHECK();
    return 0;
}

template <int KH, int KW, bool TAIL>
int launch_fir_tile(const float* x, const float* k, float* y, int planes, int in_h, int in_w, int out_h, int out_w,
                    int pad_x0, int pad_y0, const FirTail& tail, hipStream_t st) {
    auto go = [&](auto wx_tag, auto th_tag) -> int {
        constexpr int WX = decltype(wx_tag)::value, TH = decltype(th_tag)::value;
        constexpr int WY = 4 / WX;
        constexpr int RH = WY * TH + KH - 1, RW = 64 * WX + KW - 1;
        const int tiles_x = ceil_div(out_w, 64 * WX), tiles_y = ceil_div(out_h, WY * TH);
        const int64_t nblocks = (int64_t)planes * tiles_x * tiles_y;
        if (nblocks <= 0) return 0;
        if (nblocks > 0x7fffffff) return MAUA_EINVAL;
        const size_t lds_bytes = (size_t)RH * RW * sizeof(float);
        hipLaunchKernelGGL((fir_tile_kernel<KH, KW, WX, TAIL, TH>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, x, k, y, planes,
                           in_h, in_w, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, tail);
        MAUA_LAUNCH_CHECK();
        return 0;
    };
    auto go_wx = [&](auto wx_tag) -> int {
        if constexpr (!TAIL) {  // (constexpr: the strip heights of the other form are never instantiated)
            return go(wx_tag, std::integral_constant<int, FIR_ROWS_PLAIN>{});
        } else {
            if (out_h <= FIR_TAIL_SMALL_MAX_H) return go(wx_tag, std::integral_constant<int, FIR_ROWS_TAIL_SMALL>{});
            return go(wx_tag, std::integral_constant<int, FIR_ROWS_TAIL>{});
        }
    };
    // (the kernel addresses a plane through 32-bit buffer offsets)
    if ((int64_t)in_h * in_w * 4 >= 0x7fffffffLL || (int64_t)out_h * out_w * 4 >= 0x7fffffffLL) return MAUA_ENOSYS;
    if (out_w <= 64) return go_wx(std::integral_constant<int, 1>{});
    if (out_w <= 128) return go_wx(std::integral_constant<int, 2>{});
    return go_wx(std::integral_constant<int, 4>{});
}

template <bool TAIL>
int dispatch_fir_tile(const float* x, const float* k, float* y, int planes, int in_h, int in_w, int out_h, int out_w,
                      int kh, int kw, int pad_x0, int pad_y0, const FirTail& tail, hipStream_t st) {
    if (kh == 4 && kw == 4) return launch_fir_tile<4, 4, TAIL>(x, k, y, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, tail, st);
    if (kh == 3 && kw == 3) return launch_fir_tile<3, 3, TAIL>(x, k, y, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, tail, st);
    if (kh == 2 && kw == 2) return launch_fir_tile<2, 2, TAIL>(x, k, y, planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, tail, st);
    return MAUA_ENOSYS;
}

// (up, down) = (2, 1), taps up to 4 x 4: the tiled polyphase kernel
int launch_fir_up2(const float* x, const float* k, float* y, int planes, int in_h, int in_w, int out_h, int out_w, int kh, int kw,
                   int pad_x0, int pad_y0, hipStream_t st) {
    if ((int64_t)in_h * in_w * 4 >= 0x7fffffffLL || (int64_t)out_h * out_w * 4 >= 0x7fffffffLL) return MAUA_ENOSYS;
    auto go = [&](auto wx_tag) -> int {
        constexpr int WX = decltype(wx_tag)::value, TH = FIR_UP2_ROWS, WY = 4 / WX;
        const int tiles_x = ceil_div(out_w, 128 * WX), tiles_y = ceil_div(out_h, WY * TH);
        const int64_t nblocks = (int64_t)planes * tiles_x * tiles_y;
        if (nblocks <= 0) return 0;
        if (nblocks > 0x7fffffff) return MAUA_EINVAL;
        const size_t lds_bytes = sizeof(float) * (size_t)(WY * (TH / 2) + 2) * (64 * WX + 2);
        hipLaunchKernelGGL((fir_up2_kernel<WX, TH>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, x, k, y, in_h, in_w, out_h, out_w,
                           kh, kw, pad_x0, pad_y0, tiles_x, tiles_y);
        MAUA_LAUNCH_CHECK();
        return 0;
    };
    if (out_w <= 128) return go(std::integral_constant<int, 1>{});
    if (out_w <= 256) return go(std::integral_constant<int, 2>{});
    return go(std::integral_constant<int, 4>{});
}

// (up, down) = (1, 2), taps up to 4 x 4
int launch_fir_down2(const float* x, const float* k, float* y, int planes, int in_h, int in_w, int out_h, int out_w, int kh, int kw,
                     int pad_x0, int pad_y0, hipStream_t st) {
    if ((int64_t)in_h * in_w * 4 >= 0x7fffffffLL || (int64_t)out_h * out_w * 4 >= 0x7fffffffLL) return MAUA_ENOSYS;
    auto go = [&](auto wx_tag) -> int {
        constexpr int WX = decltype(wx_tag)::value, TH = FIR_DOWN2_ROWS, WY = 4 / WX;
        const int tiles_x = ceil_div(out_w, 64 * WX), tiles_y = ceil_div(out_h, WY * TH);
        const int64_t nblocks = (int64_t)planes * tiles_x * tiles_y;
        if (nblocks <= 0) return 0;
        if (nblocks > 0x7fffffff) return MAUA_EINVAL;
        const size_t lds_bytes = sizeof(float) * (size_t)(2 * WY * TH + 2) * (128 * WX + 2);
        static_assert(sizeof(float) * (2 * WY * TH + 2) * (128 * WX + 2) <= 64 * 1024, "dynamic LDS beyond 64 KB needs the launch attribute");
        hipLaunchKernelGGL((fir_down2_kernel<WX, TH>), dim3((unsigned)nblocks), dim3(256), lds_bytes, st, x, k, y, in_h, in_w, out_h, out_w,
                           kh, kw, pad_x0, pad_y0, tiles_x, tiles_y);
        MAUA_LAUNCH_CHECK();
        return 0;
    };
    if (out_w <= 64) return go(std::integral_constant<int, 1>{});
    if (out_w <= 128) return go(std::integral_constant<int, 2>{});
    return go(std::integral_constant<int, 4>{});
}

}  // namespace

extern "C" int maua_upfirdn2d_f32(const float* x, const float* k, float* y, int major, int in_h, int in_w, int minor,
                                  int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                  int pad_y0, int pad_y1, void* stream) {
    if (!x || !k || !y || major < 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0 || up_x <= 0 ||
        up_y <= 0 || down_x <= 0 || down_y <= 0)
        return MAUA_EINVAL;
    const int span_h = in_h * up_y + pad_y0 + pad_y1 - kh, span_w = in_w * up_x + pad_x0 + pad_x1 - kw;
    if (span_h < 0 || span_w < 0) return MAUA_EINVAL;  // (C division would round a negative span toward one output row)
    const int out_h = span_h / down_y + 1, out_w = span_w / down_x + 1;
    if (major == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (minor == 1 && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == kw && kh >= 2 && kh <= 4) {
        FirTail none{};
        const int rc = dispatch_fir_tile<false>(x, k, y, major, in_h, in_w, out_h, out_w, kh, kw, pad_x0, pad_y0, none, st);
        if (rc != MAUA_ENOSYS) return rc;  // (planes of 2 GiB and more take the generic gather below)
    }
    if (minor == 1 && up_x == 1 && up_y == 1 && down_x == 2 && down_y == 2 && kh <= 4 && kw <= 4) {
        const int rc = launch_fir_down2(x, k, y, major, in_h, in_w, out_h, out_w, kh, kw, pad_x0, pad_y0, st);
        if (rc != MAUA_ENOSYS) return rc;
    }
    if (minor == 1 && up_x == 2 && up_y == 2 && down_x == 1 && down_y == 1 && kh <= 4 && kw <= 4) {
        const int rc = launch_fir_up2(x, k, y, major, in_h, in_w, out_h, out_w, kh, kw, pad_x0, pad_y0, st);
        if (rc != MAUA_ENOSYS) return rc;
    }
    const int64_t total = (int64_t)major * out_h * out_w * minor;
    const int64_t blocks = ceil_div64(total, 256);
    const unsigned grid = (unsigned)(blocks < 256 * 32 ? blocks : 256 * 32);
    hipLaunchKernelGGL(fir_generic_kernel, dim3(grid), dim3(256), 0, st, x, k, y, major, in_h, in_w, minor, kh, kw,
                       up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w, total);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_blur_noise_act_f32(const float* x, const float* k, float* y, int batch, int channels, int in_h,
                                       int in_w, int kh, int kw, int pad0, int pad1, const float* gain,
                                       const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                       const float* bias, const maua_frame_source_t* src, int noise_slot, const float* post_s,
                                       int post_stride, void* stream) {
    if (!x || !k || !y || batch <= 0 || channels <= 0 || in_h <= 0 || in_w <= 0) return MAUA_EINVAL;
    if ((noise || src) && !noise_w) return MAUA_EINVAL;
    if (src && (noise_slot < 0 || noise_slot >= MAUA_MAX_NOISE_SLOTS)) return MAUA_EINVAL;
    const int out_h = in_h + pad0 + pad1 - kh + 1, out_w = in_w + pad0 + pad1 - kw + 1;
    if (out_h <= 0 || out_w <= 0) return MAUA_EINVAL;
    // Block order.  Channel-fastest (planes that share a noise tile back to back: the tile stays in one XCD's L2) pays on the 1024-row maps,
    // whose noise maps (4 MB per frame) exceed an L2; up to 512 rows the batch's noise maps fit and plane-by-plane order — sequential HBM
    // rows, as the plain op walks them — is faster: 256^2 x 128 channels 0.147 -> 0.122 ms, 512^2 0.235 -> 0.228, 1024^2 unchanged
    // (alternating A/B in bench.py, profiles/r04_probes.md)
#ifndef MAUA_FIR_PLANE_MAJOR_MAX_H
#define MAUA_FIR_PLANE_MAJOR_MAX_H 512
#endif
    FirTail tail{gain, noise, noise_w, bias, noise_batch_stride, channels, src, noise_slot, out_h <= MAUA_FIR_PLANE_MAJOR_MAX_H ? 1 : 0,
                 post_s, post_stride};
    return dispatch_fir_tile<true>(x, k, y, batch * channels, in_h, in_w, out_h, out_w, kh, kw, pad0, pad0, tail,
                                   (hipStream_t)stream);
}

extern "C" int maua_upfirdn2d_f16(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor, int kh, int kw,
                                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                  void* stream) {
    return launch_fir_typed<__half, float>(x, k, y, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                           pad_y0, pad_y1, stream);
}

extern "C" int maua_upfirdn2d_f64(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor, int kh, int kw,
                                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                  void* stream) {
    return launch_fir_typed<double, double>(x, k, y, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                            pad_y0, pad_y1, stream);
}

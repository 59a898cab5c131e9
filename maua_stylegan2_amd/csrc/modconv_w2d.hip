// Plain 3x3 modulated convolution through 2-D Winograd F(2x4, 3x3) on the fp32 matrix cores (mode 5 of
// maua_modconv3x3_f32 / maua_styledconv_torgb_f32).
//
// Behavioural contract: /root/reference/models/stylegan2.py:217-254 (plain branch :248-252) + StyledConv tail :338-343
// (+ ToRGB :356-365 when fused) — the same input-scale -> shared-weight contraction -> output-demod formulation as
// modconv.hip.  What changes is the contraction: F(2,3) along y on top of F(4,3) along x,
//
//     Y(2x4 outputs) = A_y^T [ sum_c (G_y g_c G_x^T) .* (B_y^T d_c B_x) ] A_x          d_c = 4x6 input window of channel c
//
// 24 products per 8 outputs per (cin, cout) pair = 3 MAC per output instead of 9 (direct) or 4.5 (F(4,3) along x only):
// a third fewer matrix-core cycles than mode 3 on layers that are MFMA-bound.  fp32 error: F(2,3) only adds {1, 1/2}
// constants on top of F(4,3)'s, measured ~3e-5 of the output scale (tests/test_layers_gpu.py).
//
// Work decomposition (what makes 24 accumulator tiles fit): a workgroup owns BM = 16 TM output channels x 16 TN positions
// (a position = one 2x4 output block; an n-tile = 8 x 2 positions = 32 x 4 pixels; n-tiles stack along y), and its FOUR WAVES
// SPLIT THE FOUR y-FREQUENCIES: wave fy accumulates the six x-frequencies of its own y-frequency for the whole BM x 16 TN
// tile on v_mfma_f32_16x16x4_f32 (K = 4 input channels per step, 6 TM TN accumulator tiles of 4 registers).  Consequences:
//   * the B operand (B_y^T d B_x of the staged raw patch) is never formed twice inside a workgroup: wave fy needs only row
//     combination fy of the window (one fused multiply-add per column), then the 6-point x transform — 24 VALU per
//     (position, channel) per wave for 24 TM MFMAs;
//   * every wave streams its own quarter of the weight tile from LDS, the patch is shared;
//   * the inverse y transform crosses waves: after the main loop every wave applies A_x to its accumulators and the four
//     partial 1x4 rows meet in LDS (16 channels per pass), where the tail (demod, noise, bias, leaky ReLU), the fused ToRGB
//     reduction and the 16-byte stores are done by all 256 threads.  The channel a thread handles in a step of a pass is
//     uniform over its wave: per-channel constants are one broadcast LDS read and the store is a buffer store with a scalar
//     channel offset; no selects (row 1 of A_y^T = row 0 with a sign, leaky ReLU = max(t, slope t)).
// Operands reach LDS by MUBUF `buffer_load ... lds` DMA, double buffered, one barrier per 4-channel K step; the packed weight
// (maua_pack_weight_wino2d_f32) is laid out in HBM exactly as the LDS tile image, so its DMA is a linear copy.
#include "common.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__)
#define MAUA_DEVICE_PASS 1
#endif

// Compile-time ablation mask, experiment builds only (tools/build_exp.sh <name> -DMAUA_W2D_ABL=<mask>; results are wrong by construction,
// the timing is an upper bound of what optimising that part could return): 1 no MFMA, 2 no DMA after the first chunk, 4 no feature
// stores, 8 no epilogue, 16 no K-step barrier, 32 no input transform (raw window values feed the MFMAs), 64 no window reads, 128 half the weight DMA pieces, 1024 no patch DMA,
// 256 no ToRGB tail after the passes, 512 no combine phase (barriers and the A_x^T / exchange writes stay, so the accumulators
// remain live); wave-complete kernel's epilogue split (round 6, profiles/r06_w2dw_epilogue_split.md): 2048 no inverse transform (one accumulator per
// output instead of A_x^T / A_y^T), 4096 no ToRGB products / butterfly / skip / frames, 8192 no tail (gain, noise, bias, leaky ReLU).
// 0 in the product, which carries no run-time switch of any kind.
#if !defined(MAUA_EXPERIMENTS)
#undef MAUA_W2D_ABL  // the product ignores the mask even when somebody passes it
#endif
#ifndef MAUA_W2D_ABL
#define MAUA_W2D_ABL 0
#endif
// W2D_ABL(bits): true only in an experiments build whose mask has one of `bits` (a compile-time constant: `if (W2D_ABL(..))` folds away)
#define W2D_ABL(bits) ((MAUA_W2D_ABL & (bits)) != 0)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Staged patch: rows of ten 16-byte segments = image columns tx0-4 .. tx0+35 (16-byte aligned in HBM, so that every segment
// lies wholly inside or wholly outside the image and the zero padding comes from the DMA's out-of-range rule); a position's
// 6-float window starts at row float 4 jx + 3.  LDS row stride 48 floats = 12 segment slots: two position rows (2 patch rows
// apart = 96 floats = 32 banks) land on opposite halves of the 64 banks for the 8-byte window reads.
constexpr int W2D_SEGS = 10;
constexpr int W2D_PWS = 48;
constexpr int W2D_ROWS_PER_DMA = 5;  // 64 lanes x 16 bytes = 5 1/3 LDS rows: a DMA instruction covers 5 rows (+ 4 slots of the 6th)
constexpr int W2D_CC = 4;     // input channels per K step = K of v_mfma_f32_16x16x4_f32

// LDS reads of the main loop are issued as single `ds_read_b64` instructions through inline assembly: left to itself the
// compiler pairs neighbouring 8-byte reads into ds_read2_b64 / ds_read2st64_b64, which are serviced in 16-lane groups on a
// 32-bank modulus at half the bytes per clock (MI355X_MICROARCH.md, LDS table) — measured 30 % of the LDS-active cycles of this
// kernel as bank conflicts for a layout that is conflict-free under ds_read_b64's rule (32-lane groups, 64 banks).  The
// compiler does not count inline-assembly LDS operations, so the waits are explicit as well; a wait "produces" the values it
// guards (tied operands), which keeps their consumers behind it.
template <int OFF>
__device__ __forceinline__ f32x2 lds_read64(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ float lds_read32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x2& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x2& a, f32x2& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
// 16-byte MUBUF store whose channel offset is a SCALAR (soffset SGPR), followed by its own wait states.  Round 6 finding, measured on the
// MI355X (tools/dbg/w2dw_val.py: every thread's value correct in its register, the fourth dword of the stored row not): the compiler
// treats a buffer store of more than 64 bits as hazard-free when its soffset is a register (LLVM GCNHazardRecognizer::createsVALUHazard:
// "this hazard only exists if the instruction is not using a register in the soffset field") and lets the very next instructions
// overwrite the data registers — it had emitted `buffer_store_dwordx4 v[190:193], .., s67 offen` followed at once by four v_mov into
// v190..v193 for the next row, and on gfx950 the row-0 store then carried the NEXT row's last dword in a quarter of its lanes.  With an
// immediate soffset it keeps 2 wait states (and those builds were right).  The store therefore goes out as one inline-assembly blob
// with `s_nop` behind it: no instruction the compiler schedules can reach the data registers earlier than 4 wait states after the issue.
__device__ __forceinline__ void buffer_store_b128_sgpr_offset(u32x4 data, const float* base, unsigned voffset_bytes, unsigned soffset_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const uint64_t a = (uint64_t)(uintptr_t)base;
    // raw buffer descriptor (stride 0, 2^31 - 1 records, DATA_FORMAT = 32 as __builtin_amdgcn_make_buffer_rsrc(.., 0, 0x7fffffff, 0x00020000))
    i32x4 rsrc = i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff), 0x7fffffff, 0x00020000};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(data), "v"(voffset_bytes), "s"(rsrc), "s"(soffset_bytes) : "memory");
#else
    (void)data, (void)base, (void)voffset_bytes, (void)soffset_bytes;
#endif
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct W2dArgs {
    const float* x;
    const float* wq;
    const float* s;
    const float* d;
    const float* noise;
    const float* noise_w;
    const float* bias;
    float* y;
    const float* rgb_w;
    const float* rgb_s;
    const float* rgb_bias;
    const float* rgb_skip;
    const float* rgb_k4;
    float* rgb_out;
    uint8_t* rgb_u8;  // when set: uint8 NHWC frames [B, H, W, 3] (render.py:40-43); rgb_out may then be NULL
    // the style fold (round 6).  Producer side: post_s [B, s_stride] = the styles of the layer that CONSUMES y — the stored feature map is
    // act(..) * post_s[b, o] (the fused ToRGB reads the un-scaled value), so that the consumer's K loop drops its per-window style multiplies
    // (models/stylegan2.py:220-221 reassociated: conv(W, x * s) with the multiply moved into the producer's epilogue, once per element
    // instead of once per output-channel tile).  Consumer side: s == NULL (kernel instances with PRE = true) = x arrives pre-scaled.
    const float* post_s;
    int B, Cin, Cout, H, W;
    int s_stride;
    float wscale;
    int fuse_act;
    int64_t noise_batch_stride;
    int tiles_x, tiles_y, m_tiles, n_chunks;
    int rgb;  // 0 off, 1 on, 2 on and the feature map itself is not stored, 3 partial: this m-tile's share of the ToRGB sum goes to
              // rgb_out [B, 3 m_tiles, H, W] (no bias, no skip), the feature map is stored
    float rgb_wscale;
    const maua_frame_source_t* src;  // frame source (include/maua_hip.h): noise from src->noise[noise_slot] at frame src->frame0
    int noise_slot;
};

__host__ __device__ constexpr int w2d_dma_per_channel(int tn) { return (4 * tn + 2 + W2D_ROWS_PER_DMA - 1) / W2D_ROWS_PER_DMA; }
__host__ __device__ constexpr int w2d_pstride(int tn) {
    // floats per channel of the staged patch: whole DMA instructions (the last one runs 16 floats past its fifth row)
    return w2d_dma_per_channel(tn) * W2D_ROWS_PER_DMA * W2D_PWS + 16;
}
// Odd channel planes of a chunk sit W2D_ODD_SHIFT floats further: the window of a position is read as 8-byte pairs that start at an
// even float of its row (floats 4 jx + 2 ..), i.e. on banks = 2, 3 (mod 4) whatever 16-byte-aligned offset the plane has, and a
// 32-lane group of ds_read_b64 holds TWO K lanes (channels kq, kq + 1): 64 eight-byte accesses onto the 32 banks = 2, 3 (mod 4), a
// 2-way conflict on every window read (round 3: 30 % of the LDS-active cycles; removing the window reads alone is worth 0.10 ms of
// a 0.50 ms launch, profiles/r04_w2d_ablation.md).  With the odd planes two floats off their pairs fall on banks = 0, 1 (mod 4):
// conflict-free.  The LDS-DMA takes an 8-byte-aligned LDS destination (tools/dma_align_probe.hip, measured on the MI355X), the
// global side stays 16-byte aligned, so the zero padding still comes from whole out-of-range segments.
#ifndef MAUA_W2D_ODD_SHIFT
#define MAUA_W2D_ODD_SHIFT 2
#endif
constexpr int W2D_ODD_SHIFT = MAUA_W2D_ODD_SHIFT;
__host__ __device__ constexpr int w2d_pbuf(int tn) { return W2D_CC * w2d_pstride(tn) + 4; }  // (+4: the last plane's shifted overrun)

// physical column of (m-tile mt, row i16) inside a weight row of BM = 16 TM floats: m-tile pairs interleaved so that one
// 8-byte LDS read feeds two MFMAs; for BM = 64 odd K lanes are rotated by half a row (their 256-byte row stride would put
// them on the banks of the even ones)
__host__ __device__ inline int w2d_col(int tm, int mt, int i16, int kq) {
    const int col = (mt >> 1) * 32 + i16 * 2 + (mt & 1);
    return tm == 4 ? ((col + 32 * (kq & 1)) & 63) : col;
}

// Build switches (defaults = the product; tools/build_exp.sh builds the alternatives for A/B runs):
//   MAUA_W2D_TN32 / MAUA_W2D_MINB32  n-tiles per workgroup and workgroups per CU of the 32-output-channel layers.  Round 3: 2 n-tiles
//       (156 VGPRs, 44 KB LDS) at three workgroups per CU instead of 4 n-tiles at two: the layer is neither matrix- nor HBM-bound, its
//       eight-step K loop leaves the prologue DMA wait, the per-step barrier and the LDS exchange of the epilogue exposed, and a third
//       resident workgroup covers more of them (convs.15 0.79 -> 0.73 ms, convs.13 unchanged, +1-2 % frames/s under three lanes).
//   MAUA_W2D_NMAJOR  32-channel layers: per n-tile transform -> next window read -> the tile's MFMAs (1) or all transforms first (0)
//   MAUA_W2D_PKT     input transform on register pairs (1) or scalar (0)
//   MAUA_W2D_READ_FIRST  the window reads of a K step go out BEFORE the DMA of the next chunk is issued (1) or behind it (0)
#ifndef MAUA_W2D_READ_FIRST
#define MAUA_W2D_READ_FIRST 0
#endif
#ifndef MAUA_W2D_NMAJOR
#define MAUA_W2D_NMAJOR 1
#endif
#ifndef MAUA_W2D_PKT
#define MAUA_W2D_PKT 1
#endif
#ifndef MAUA_W2D_TN32
#define MAUA_W2D_TN32 2
#endif
#ifndef MAUA_W2D_MINB32
#define MAUA_W2D_MINB32 3
#endif
//   MAUA_W2D_TN64 / MAUA_W2D_MINB64  the same for the layers of 64 and more output channels (TM = 4).  2 n-tiles at two workgroups per CU
//       (247 VGPRs); the round-5 experiment is 4 n-tiles at ONE workgroup per CU = one wave per SIMD with a 512-register budget, twice the
//       outputs per transformed window and per weight piece (VERDICT r4 item 4 i; profiles/r05_w2d_one_wave_per_simd.md)
#ifndef MAUA_W2D_TN64
#define MAUA_W2D_TN64 2
#endif
#ifndef MAUA_W2D_MINB64
#define MAUA_W2D_MINB64 2
#endif

template <int TM, int TN, int MINB = 2, bool PRE = false>
__global__ __launch_bounds__(256, MINB) void modconv_w2d_kernel(W2dArgs p) {
    constexpr int BM = 16 * TM;
    constexpr int NPOS = 16 * TN;
    constexpr int TH = 4 * TN;  // output rows per tile
    constexpr int PH = TH + 2;
    constexpr int PSTRIDE = w2d_pstride(TN);
    constexpr int NQ = w2d_dma_per_channel(TN);  // patch DMA instructions per channel
    constexpr int A_FLOATS = 24 * W2D_CC * BM;
    constexpr int A_INSTR = A_FLOATS / 256;  // 1-KiB DMA instructions per weight tile
    constexpr int A_PER_WAVE = A_INSTR / 4;
    constexpr int PBUF = w2d_pbuf(TN);
    static_assert(A_INSTR % 4 == 0, "weight tile must split evenly over the four waves");
    (void)A_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    // per-channel constants of the epilogue live behind whatever is larger, the main loop's buffers or the epilogue's exchange buffers
    constexpr int EPI_FLOATS = 4 * 16 * NPOS * 4 + (256 / (2 * NPOS)) * 2 * NPOS * 12;
    const int main_floats = 2 * A_FLOATS + 2 * PBUF + p.Cin;
    const int e_off = main_floats > EPI_FLOATS ? main_floats : EPI_FLOATS;
    float* lds = lds_all;
    float* As = lds;                      // [2][A_FLOATS]
    (void)As;
    float* Ps = lds + 2 * A_FLOATS;       // [2][PBUF]
    float* Ss = Ps + 2 * PBUF;            // [Cin] styles of this image

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int fy = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's y-frequency
    const int j = lane & 15, kq = lane >> 4;
    const int jx = j & 7, jy = j >> 3;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int mt_id = t % p.m_tiles;
    t /= p.m_tiles;
    const int tile_x = t % p.tiles_x;
    t /= p.tiles_x;
    const int tile_y = t % p.tiles_y;
    const int b0 = t / p.tiles_y;
    const int ty0 = tile_y * TH, tx0 = tile_x * 32;
    const int m0 = mt_id * BM;
    const size_t plane = (size_t)p.H * p.W;

    // ---- patch DMA of this lane (decoded once).  The 4 NQ instructions of a chunk (channel c, row group q) are dealt to the
    // waves round-robin; NQ divides 4 or is a multiple of it, so a wave always draws the same q: one source offset per lane.
    // Lane l of instruction (c, q) fills 16-byte slot l of the group = patch row 5 q + l / 12, segment l % 12 (slots 60..63 are
    // the first four of the next row: the same bytes the next instruction writes there).  Segments 10, 11, rows past the patch
    // and everything outside the image get an offset beyond the buffer descriptor's range, for which a raw buffer load returns
    // 0: the DMA itself writes the convolution's zero padding — no exec masks, no pre-zeroing.
    static_assert(NQ == 1 || NQ == 2 || NQ % 4 == 0, "a wave must keep one row group");
    unsigned rel_bytes[(NQ + 3) / 4];
#pragma unroll
    for (int g = 0; g < (NQ + 3) / 4; ++g) {
        const int q = (fy + 4 * g) % NQ;
        const int pr = W2D_ROWS_PER_DMA * q + lane / 12, sg = lane % 12;
        const int yy = ty0 + pr - 1, xx = tx0 - 4 + 4 * sg;
        const bool ok = sg < W2D_SEGS && pr < PH && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        rel_bytes[g] = ok ? (unsigned)(yy * p.W + xx) * 4u : 0x80000000u;
    }
    if constexpr (!PRE)
        for (int e = tid; e < p.Cin; e += 256) Ss[e] = p.s[(size_t)b0 * p.s_stride + e];
    const char* ximg = reinterpret_cast<const char*>(p.x + (size_t)b0 * p.Cin * plane);
    const size_t plane_bytes = plane * sizeof(float);
    (void)ximg, (void)plane_bytes, (void)Ss;
#ifdef MAUA_DEVICE_PASS
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ximg), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wq), 0, 0x7fffffff, 0x00020000);
#endif
    auto issue = [&](int chunk, int buf) {
#ifdef MAUA_DEVICE_PASS
        // weight tile: linear copy of A_FLOATS floats, 1 KiB per wave instruction
        const int wbase = (int)(((size_t)mt_id * p.n_chunks + chunk) * A_FLOATS * sizeof(float));
#pragma unroll
        for (int k = 0; k < A_PER_WAVE; ++k) {
            const int i = fy + 4 * k;
            if (W2D_ABL(128) && (k & 1) && chunk) continue;  // (ablation: half the weight pieces)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(As + buf * A_FLOATS + i * 256),
                                                     16, (i * 256 + lane * 4) * 4, wbase, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < (4 * NQ + 3) / 4; ++k) {
            const int id = fy + 4 * k;  // (scalar) instruction id = channel * NQ + row group
            if (id < W2D_CC * NQ && !(W2D_ABL(1024) && chunk)) {  // (ablation 1024: no patch DMA after the first chunk)
                const int c = id / NQ, q = id % NQ;
                float* dst = Ps + buf * PBUF + c * PSTRIDE + (c & 1) * W2D_ODD_SHIFT + q * (W2D_ROWS_PER_DMA * W2D_PWS);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)dst, 16,
                                                         (int)rel_bytes[(NQ % 4 == 0) ? (k % (NQ / 4 > 0 ? NQ / 4 : 1)) : 0],
                                                         (int)((size_t)(chunk * W2D_CC + c) * plane_bytes), 0, 0);
            }
        }
#else
        (void)chunk, (void)buf;
#endif
    };

    // per-output-channel constants of the epilogue, [BM][8] = gain, bias, the three modulated ToRGB weights: fetched here, under the
    // first DMA wait, into LDS that neither the main loop nor the exchange buffers of the epilogue touch (loaded after the main loop
    // their ~1 us round trip was exposed in every workgroup)
    float* E = lds + e_off;
    const bool act = p.fuse_act != 0;
    const float act_gain = act ? 1.41421356237309515f : 1.f;
    const float* noise_base = p.noise;
    int64_t noise_bstride = p.noise_batch_stride;
    if (p.src) {  // uniform scalar loads: base of this launch's first frame inside the HBM-resident sequence
        noise_bstride = p.src->noise_stride[p.noise_slot];
        noise_base = p.src->noise[p.noise_slot];
        if (noise_base) noise_base += (int64_t)p.src->frame0 * noise_bstride;
    }
    const float nw = (act && noise_base) ? p.noise_w[0] * act_gain : 0.f;
    for (int i = tid; i < BM; i += 256) {
        const int o = m0 + i;
        float gain = p.wscale * act_gain;
        if (p.d) gain *= p.d[(size_t)b0 * p.Cout + o];
        f32x4 e = f32x4{gain, (act && p.bias) ? p.bias[o] * act_gain : 0.f, 0.f, 0.f};
        float r2 = 0.f;
        if (p.rgb) {
            const float ms = p.rgb_wscale * p.rgb_s[(size_t)b0 * p.s_stride + o];
            e[2] = ms * p.rgb_w[0 * p.Cout + o], e[3] = ms * p.rgb_w[1 * p.Cout + o], r2 = ms * p.rgb_w[2 * p.Cout + o];
        }
        *reinterpret_cast<f32x4*>(E + 8 * i) = e;
        E[8 * i + 4] = r2;
        E[8 * i + 5] = p.post_s ? p.post_s[(size_t)b0 * p.s_stride + o] : 1.f;  // scale of the STORED feature value (the consumer's style)
    }

    // ---- accumulators: [x-frequency][m-tile][n-tile]
    // (not zeroed: the first K step runs as a peeled copy of the loop whose matrix instructions take C = 0 — 48 TM TN / 8 register
    // moves per wave that the VALU, which shares its datapath with the fp32 matrix instructions, does not have to issue)
    f32x4 acc[6][TM][TN];

    // window rows combined by y-frequency fy (F(2,3) B^T): fy 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = fy == 0 ? 0 : (fy == 2 ? 2 : 1);
    const int rb = fy == 0 ? 2 : (fy == 1 ? 2 : (fy == 2 ? 1 : 3));
    const float sgn = fy == 1 ? 1.f : -1.f;
    const f32x2 sgn2 = f32x2{sgn, sgn};
    float m5 = -5.f;
    asm("" : "+v"(m5));  // (a register operand of the hand-written transform)
    (void)sgn2;
    // LDS byte addresses (buffer 0) of this lane's operands; the second buffer is a constant distance away
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    // window rows ra / rb of n-tile 0, channel kq; n-tile n lies a constant 4 patch rows further (immediate offset)
    unsigned b_addr[2];
    {
        const int top = kq * PSTRIDE + (kq & 1) * W2D_ODD_SHIFT + (2 * jy) * W2D_PWS + 4 * jx + 3;  // the window starts one column left of the position
        b_addr[0] = lds0 + (unsigned)(2 * A_FLOATS + top + ra * W2D_PWS) * 4u;
        b_addr[1] = lds0 + (unsigned)(2 * A_FLOATS + top + rb * W2D_PWS) * 4u;
    }
    constexpr int NT_BYTES = 4 * W2D_PWS * 4;
    // weight rows of this wave: ((fy * 6 + xf) * 4 + kq) * BM + physical column of (m-tile pair h, row j)
    unsigned a_addr[TM / 2];
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) a_addr[h] = lds0 + (unsigned)((fy * 24 + kq) * BM + w2d_col(TM, 2 * h, j, kq)) * 4u;
    unsigned s_addr = lds0 + (unsigned)(2 * A_FLOATS + 2 * PBUF + kq) * 4u;
    constexpr unsigned A_BUF_BYTES = A_FLOATS * 4u, P_BUF_BYTES = PBUF * 4u;
    constexpr int XF_BYTES = W2D_CC * BM * 4;  // distance between x-frequencies in the weight tile

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int phase = 0; phase < 2; ++phase)  // phase 0 = the first K step (C = 0), phase 1 = the others
    for (int chunk = phase; chunk < (phase ? p.n_chunks : 1); ++chunk) {
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#if !MAUA_W2D_READ_FIRST
        if (chunk + 1 < p.n_chunks && !W2D_ABL(2)) issue(chunk + 1, cur ^ 1);
#endif
        // ---- operand reads of this chunk: style, raw window rows, first weight row
        const unsigned a_off = cur ? A_BUF_BYTES : 0u, p_off = cur ? P_BUF_BYTES : 0u;
        float sc = 1.f;
        if constexpr (!PRE) {
            sc = lds_read32(s_addr);
            s_addr += W2D_CC * 4u;
        }
        unsigned ap[TM / 2];
#pragma unroll
        for (int h = 0; h < TM / 2; ++h) ap[h] = a_addr[h] + a_off;
        f32x2 a2[2][TM / 2];
        constexpr bool NMAJOR = TM == 2 && MAUA_W2D_NMAJOR;
        float bv[NMAJOR ? 1 : TN][6];
        // windows are read two n-tiles ahead of their transform (12 registers each: at most two are live), the first weight row
        // goes out behind the last window; LDS returns in order, so "at most k operations outstanding" identifies what landed
        // (with four n-tiles per wave the register file only has room for one window in flight)
        constexpr int WSLOTS = TN > 2 ? 1 : 2;
        // the 6-float window (row floats 4 jx + 3 .. 4 jx + 8) is read as four aligned 8-byte pairs, columns -1|0, 1|2, 3|4, 5|6:
        // 4-byte reads of the odd-aligned window edges are 4-way bank conflicts (4 lanes per bank in a 32-lane group), the 8-byte
        // ones 2-way at worst (round 3: -2 % over the plain-layer family, 45 % -> ~25 % of the LDS-active cycles in conflicts)
        f32x2 wa0[WSLOTS], wa5[WSLOTS], wb0[WSLOTS], wb5[WSLOTS];
        f32x2 wa[WSLOTS][2], wb[WSLOTS][2];
        const unsigned pa = b_addr[0] + p_off - 4u, pb = b_addr[1] + p_off - 4u;
        auto read_window = [&](auto n_c, int slot) {
            constexpr int o = decltype(n_c)::value * NT_BYTES;
            if constexpr (W2D_ABL(64)) {  // (ablation: the registers keep whatever they held)
                asm volatile("" : "=v"(wa0[slot]), "=v"(wa5[slot]), "=v"(wb0[slot]), "=v"(wb5[slot]), "=v"(wa[slot][0]), "=v"(wa[slot][1]),
                             "=v"(wb[slot][0]), "=v"(wb[slot][1]));
                return;
            }
            wa0[slot] = lds_read64<o>(pa), wa[slot][0] = lds_read64<o + 8>(pa), wa[slot][1] = lds_read64<o + 16>(pa);
            wa5[slot] = lds_read64<o + 24>(pa);
            wb0[slot] = lds_read64<o>(pb), wb[slot][0] = lds_read64<o + 8>(pb), wb[slot][1] = lds_read64<o + 16>(pb);
            wb5[slot] = lds_read64<o + 24>(pb);
        };
        auto transform = [&](int n, int slot) {
            if constexpr (W2D_ABL(32)) {  // (ablation: raw window values feed the matrix instructions)
                bv[n][0] = wa0[slot].y, bv[n][1] = wa[slot][0].x, bv[n][2] = wa[slot][0].y, bv[n][3] = wb[slot][1].x, bv[n][4] = wb[slot][1].y;
                bv[n][5] = wb5[slot].x + sc;
                return;
            }
#if MAUA_W2D_PKT
            const f32x2 sc2 = f32x2{sc, sc};
            // row combination and B_x^T of F(4,3) on register pairs (v_pk_*_f32 = two fp32 operations per issue slot; VALU and the
            // fp32 matrix instructions share the datapath, so every transform instruction saved is matrix time won):
            //   D12 = (d1, d2), D34 = (d3, d4);  (b, a) = D34 - 4 D12;  (e, c) = D34 - D12;
            //   (bv1, bv2) = (a + b, a - b);  (bv3, bv4) = (c + 2 e, c - 2 e)   -- the two cross-lane forms through op_sel / neg_hi
            // Written instruction by instruction (16 per window): left to the compiler the same arithmetic comes out at ~20 with the
            // register shuffles its vectoriser adds around the two scalar chains.
            f32x2 D12, D34, ba, ec, b12, b34;
            float d0, d5, t0, t5, u0, u5;
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(D12) : "v"(wb[slot][0]), "v"(sgn2), "v"(wa[slot][0]));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(D34) : "v"(wb[slot][1]), "v"(sgn2), "v"(wa[slot][1]));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d0) : "v"(wb0[slot].y), "v"(sgn), "v"(wa0[slot].y));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d5) : "v"(wb5[slot].x), "v"(sgn), "v"(wa5[slot].x));
            asm("v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]" : "=v"(ba) : "v"(D12), "v"(D34));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(ec) : "v"(D34), "v"(D12));
            asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(b12) : "v"(ba));
            asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(b34) : "v"(ec));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(m5), "v"(D12.y), "v"(D34.y));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t5) : "v"(m5), "v"(D34.x), "v"(d5));
            asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(u0) : "v"(d0), "v"(t0));
            asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(u5) : "v"(D12.x), "v"(t5));
            if constexpr (!PRE) {  // (PRE: the producer's epilogue multiplied the map by this layer's styles: 12 instead of 16 per window)
                asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(b12) : "v"(b12), "v"(sc2));
                asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(b34) : "v"(b34), "v"(sc2));
                u0 *= sc, u5 *= sc;
            }
            bv[n][0] = u0, bv[n][5] = u5;
            bv[n][1] = b12.x, bv[n][2] = b12.y, bv[n][3] = b34.x, bv[n][4] = b34.y;
#else
            const float d0 = fmaf(sgn, wb0[slot].y, wa0[slot].y), d1 = fmaf(sgn, wb[slot][0].x, wa[slot][0].x);
            const float d2 = fmaf(sgn, wb[slot][0].y, wa[slot][0].y), d3 = fmaf(sgn, wb[slot][1].x, wa[slot][1].x);
            const float d4 = fmaf(sgn, wb[slot][1].y, wa[slot][1].y), d5 = fmaf(sgn, wb5[slot].x, wa5[slot].x);
            // B_x^T for F(4,3) (interpolation points 0, +-1, +-2, inf), as in modconv.hip's mode 3
            const float a_ = fmaf(-4.f, d2, d4), b_ = fmaf(-4.f, d1, d3);
            const float c_ = d4 - d2, e_ = d3 - d1;
            static_assert(!PRE, "the pre-scaled instances use the packed transform");
            bv[n][0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4)) * sc;
            bv[n][1] = (a_ + b_) * sc;
            bv[n][2] = (a_ - b_) * sc;
            bv[n][3] = fmaf(2.f, e_, c_) * sc;
            bv[n][4] = fmaf(-2.f, e_, c_) * sc;
            bv[n][5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5)) * sc;
#endif
        };
        if constexpr (NMAJOR) {
            // n-major order (32-channel layers, TM = 2): the six weight rows of the K step are read once up front, then per n-tile
            // transform -> window read of the next n-tile -> the tile's 6 TM MFMAs, so that the LDS latency of every window but the
            // first hides behind 12 matrix instructions and only one transformed window (6 registers, not 6 TN) is live
            f32x2 aw[6][TM / 2];
            static_for<0, 6>([&](auto xf_c) {
                constexpr int xf = decltype(xf_c)::value;
#pragma unroll
                for (int h = 0; h < TM / 2; ++h) aw[xf][h] = lds_read64<xf * XF_BYTES>(ap[h]);
            });
            read_window(std::integral_constant<int, 0>{}, 0);
            if constexpr (TN > 1 && WSLOTS > 1) read_window(std::integral_constant<int, 1>{}, 1);
#if MAUA_W2D_READ_FIRST
            if (chunk + 1 < p.n_chunks && !W2D_ABL(2)) issue(chunk + 1, cur ^ 1);
#endif
            static_for<0, TN>([&](auto n_c) {
                constexpr int n = decltype(n_c)::value;
                constexpr int slot = n % WSLOTS;
                constexpr int behind = (n + 1 < TN && WSLOTS > 1) ? 8 : 0;
                if constexpr (PRE)
                    asm volatile("s_waitcnt lgkmcnt(%8)"
                                 : "+v"(wa[slot][0]), "+v"(wa[slot][1]), "+v"(wb[slot][0]), "+v"(wb[slot][1]), "+v"(wa0[slot]),
                                   "+v"(wa5[slot]), "+v"(wb0[slot]), "+v"(wb5[slot])
                                 : "n"(behind));
                else
                asm volatile("s_waitcnt lgkmcnt(%9)"
                             : "+v"(sc), "+v"(wa[slot][0]), "+v"(wa[slot][1]), "+v"(wb[slot][0]), "+v"(wb[slot][1]), "+v"(wa0[slot]),
                               "+v"(wa5[slot]), "+v"(wb0[slot]), "+v"(wb5[slot])
                             : "n"(behind));
                if constexpr (n == 0) {  // the weight rows went out first: landed whenever window 0 has
#pragma unroll
                    for (int xf = 0; xf < 6; ++xf)
#pragma unroll
                        for (int h = 0; h < TM / 2; ++h) asm volatile("" : "+v"(aw[xf][h]));
                }
                transform(0, slot);
                if constexpr (n + WSLOTS < TN) read_window(std::integral_constant<int, n + WSLOTS>{}, slot);
                __builtin_amdgcn_sched_barrier(0);
                if (!W2D_ABL(1)) {
#pragma unroll
                    for (int xf = 0; xf < 6; ++xf)
#pragma unroll
                        for (int h = 0; h < TM / 2; ++h) {
                            acc[xf][2 * h][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[xf][h].x, bv[0][xf], phase ? acc[xf][2 * h][n] : zero4, 0, 0, 0);
                            acc[xf][2 * h + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[xf][h].y, bv[0][xf], phase ? acc[xf][2 * h + 1][n] : zero4, 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
        read_window(std::integral_constant<int, 0>{}, 0);
        if constexpr (TN > 1 && WSLOTS > 1) read_window(std::integral_constant<int, 1>{}, 1);
#if MAUA_W2D_READ_FIRST
        if (chunk + 1 < p.n_chunks && !W2D_ABL(2)) issue(chunk + 1, cur ^ 1);
#endif
        static_for<0, TN>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            constexpr int slot = n % WSLOTS;
            if constexpr (n == TN - 1) {  // behind the last window: the first weight row
#pragma unroll
                for (int h = 0; h < TM / 2; ++h) a2[0][h] = lds_read64<0>(ap[h]);
            }
            // outstanding behind window n: window n+1 (8 reads) unless n is the last, plus the weight reads when they are out
            constexpr int behind = ((n + 1 < TN && WSLOTS > 1) ? 8 : 0) + (n == TN - 1 ? TM / 2 : 0);
            if constexpr (PRE)
                asm volatile("s_waitcnt lgkmcnt(%8)"
                             : "+v"(wa[slot][0]), "+v"(wa[slot][1]), "+v"(wb[slot][0]), "+v"(wb[slot][1]), "+v"(wa0[slot]),
                               "+v"(wa5[slot]), "+v"(wb0[slot]), "+v"(wb5[slot])
                             : "n"(behind));
            else
            asm volatile("s_waitcnt lgkmcnt(%9)"
                         : "+v"(sc), "+v"(wa[slot][0]), "+v"(wa[slot][1]), "+v"(wb[slot][0]), "+v"(wb[slot][1]), "+v"(wa0[slot]),
                           "+v"(wa5[slot]), "+v"(wb0[slot]), "+v"(wb5[slot])
                         : "n"(behind));
            transform(n, slot);
            if constexpr (n + WSLOTS < TN) read_window(std::integral_constant<int, n + WSLOTS>{}, slot);
        });
        // ---- MFMA phase: the weight row of the next x-frequency is read one step ahead
        static_for<0, 6>([&](auto xf_c) {
            constexpr int xf = decltype(xf_c)::value;
            if constexpr (xf < 5) {
#pragma unroll
                for (int h = 0; h < TM / 2; ++h) a2[(xf + 1) & 1][h] = lds_read64<(xf + 1) * XF_BYTES>(ap[h]);
            }
            constexpr int pending = xf < 5 ? TM / 2 : 0;
            if constexpr (TM == 4) lds_wait<pending>(a2[xf & 1][0], a2[xf & 1][1]);
            else lds_wait<pending>(a2[xf & 1][0]);
            if (W2D_ABL(1) && !phase) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n) acc[xf][m][n] = zero4;
            }
            if (!W2D_ABL(1)) {
#pragma unroll
                for (int h = 0; h < TM / 2; ++h)
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        acc[xf][2 * h][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[xf & 1][h].x, bv[n][xf], phase ? acc[xf][2 * h][n] : zero4, 0, 0, 0);
                        acc[xf][2 * h + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[xf & 1][h].y, bv[n][xf], phase ? acc[xf][2 * h + 1][n] : zero4, 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!W2D_ABL(16)) __syncthreads();
        cur ^= 1;
    }

    if (W2D_ABL(8)) {
        if (p.B < 0) {  // never true: keeps the accumulators alive in builds without the epilogue
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n) sum += acc[a][m][n];
            *reinterpret_cast<f32x4*>(p.y + tid * 4) = sum;
        }
        return;
    }
    // ---- epilogue
    // A_x^T m in registers: y0 = m0+m1+m2+m3+m4, y1 = (m1-m2) + 2(m3-m4), y2 = (m1+m2) + 4(m3+m4), y3 = (m1-m2) + 8(m3-m4) + m5
    // then, 16 output channels (one m-tile) per pass, the four waves' 1x4 partial rows meet in LDS:
    //   output row 0 = Z0 + Z1 + Z2,  row 1 = Z1 - Z2 - Z3     (A_y^T of F(2,3))
    float* Z = lds;                               // [4 fy][16 ch][NPOS][4]
    float* Rr = lds + 4 * 16 * NPOS * 4;          // [CG][2 * NPOS][12] ToRGB partial sums
    // combine-phase role of this thread: position cp, output row cr of the 2-row block, channel group cg of every pass
    constexpr int CG = 256 / (2 * NPOS);        // channel groups (4 for TN = 2, 2 for TN = 4)
    static_assert(2 * NPOS >= 64, "the combine phase reads its channel index as a wave-uniform scalar: a wave must not span channel groups");
    constexpr int CPG = 16 / CG;                // channels per thread per pass
    const int cp = tid % NPOS, cr = (tid / NPOS) & 1, cg = tid / (2 * NPOS);
    const int cjx = cp & 7, cpy = cp >> 3;      // position column / position row inside the tile (cpy = 2 nt + jy)
    const int oy = ty0 + 2 * cpy + cr, ox = tx0 + 4 * cjx;
    f32x4 nz = f32x4{0.f, 0.f, 0.f, 0.f};  // raw noise of this thread's four pixels (scaled by nw where it is used: no wait here)
    if (nw != 0.f) nz = *reinterpret_cast<const f32x4*>(noise_base + (size_t)b0 * noise_bstride + (size_t)oy * p.W + ox);
    f32x4 rgbv[3];  // this thread's share of the ToRGB sums, [rgb channel][pixel]
#pragma unroll
    for (int c = 0; c < 3; ++c) rgbv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool store_feat = p.rgb != 2 && !W2D_ABL(4);
    const unsigned pix_off = (unsigned)oy * (unsigned)p.W + (unsigned)ox;
    // the channel a thread combines in step q of a pass is uniform over its wave (cg = tid / (2 NPOS), 2 NPOS >= 64): per-channel
    // constants come from one broadcast LDS read, and the feature store is a buffer store whose channel offset is a scalar
    const int cg_s = __builtin_amdgcn_readfirstlane(cg);
    const float row_sign = cr ? -1.f : 1.f;      // A_y^T of F(2,3): row 0 = Z1 + (Z2 + Z0), row 1 = Z1 - (Z2 + Z3)
    const float slope = act ? 0.2f : 1.f;        // leaky ReLU as max(t, slope t); slope 1 = no activation
    const unsigned plane_b = (unsigned)plane * 4u;
    const float* y_base = p.y + ((size_t)b0 * p.Cout + m0) * plane;  // (uniform: the store builds its buffer descriptor from it)
    (void)y_base;
    const float* zbase = Z + cp * 4;
    const int zo_off = (cr ? 3 : 0) * 16 * NPOS * 4;

    // 2x FIR-upsampled skip image of the fused ToRGB: upfirdn2d(skip, k4, up=2, pad=(2,1)) at (oy, x) has two live source rows / columns,
    // iy0 = floor((oy-1)/2), iy0+1 with taps k4[3]/k4[1] for even oy and k4[2]/k4[0] for odd (models/stylegan2.py:34-52,
    // op/upfirdn2d.py:159-200).  All 24 source values of the 4 output pixels (2 live rows x 4 live columns x 3 channels) are fetched
    // unconditionally from clamped addresses, in flight together; positions outside the skip image are masked through their tap
    // weight (per-pixel conditional loads serialise ~48 dependent L2 round trips behind each other: measured 0.5 ms of the 1024^2
    // layer).  Output column x = ox + px reads source columns (x-1)>>1 and +1: px 0 -> k 0,1; 1, 2 -> k 1,2; 3 -> k 2,3 of
    // k = (ox>>1) - 1 + {0..3}, with taps k4[.][3], k4[.][1] for even x and k4[.][2], k4[.][0] for odd x.  The loads go out before
    // the combine of the LAST pass (the accumulators are dead by then), so that their round trip hides behind it.
    const int sh = p.H >> 1, sw = p.W >> 1;
    const bool want_skip = (p.rgb == 1 || p.rgb == 2) && p.rgb_skip && cg == 0;
    float sv[3][2][4], wy[2], wx[4];
    auto load_skip = [&]() {
        const float* skip_img = p.rgb_skip + (size_t)b0 * 3 * sh * sw;
        const int iy0 = (oy - 1) >> 1;
        int rowc[2], colc[4];
#pragma unroll
        for (int qy = 0; qy < 2; ++qy) {
            const int yy = iy0 + qy;
            wy[qy] = (yy >= 0 && yy < sh) ? 1.f : 0.f;
            rowc[qy] = min(max(yy, 0), sh - 1);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = (ox >> 1) - 1 + k;
            wx[k] = (xx >= 0 && xx < sw) ? 1.f : 0.f;
            colc[k] = min(max(xx, 0), sw - 1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[c][qy][k] = skip_img[(unsigned)((c * sh + rowc[qy]) * sw + colc[k])];
    };

#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        __syncthreads();  // previous pass's reads (or the main loop's LDS use) are done
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float m0_ = acc[0][mt][n][v], m1_ = acc[1][mt][n][v], m2_ = acc[2][mt][n][v];
                const float m3_ = acc[3][mt][n][v], m4_ = acc[4][mt][n][v], m5_ = acc[5][mt][n][v];
                const float s12 = m1_ + m2_, d12 = m1_ - m2_, s34 = m3_ + m4_, d34 = m3_ - m4_;
                const f32x4 o4 = f32x4{(m0_ + s12) + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m5_};
                const int ch16 = 4 * kq + v;  // row of the 16x16 result tile held in register v
                *reinterpret_cast<f32x4*>(Z + (((fy * 16 + ch16) * NPOS) + n * 16 + j) * 4) = o4;
            }
        __syncthreads();
        if (mt == TM - 1 && want_skip) load_skip();
        if constexpr (!W2D_ABL(512))
#pragma unroll
        for (int q = 0; q < CPG; ++q) {
            const int ch16 = cg_s * CPG + q;
            const int ol = mt * 16 + ch16;
            const float* zp = zbase + ch16 * NPOS * 4;
            const f32x4 z1 = *reinterpret_cast<const f32x4*>(zp + 1 * 16 * NPOS * 4);
            const f32x4 z2 = *reinterpret_cast<const f32x4*>(zp + 2 * 16 * NPOS * 4);
            const f32x4 zo = *reinterpret_cast<const f32x4*>(zp + zo_off);
            const f32x4 e = *reinterpret_cast<const f32x4*>(E + 8 * ol);  // gain, bias, ToRGB weights 0, 1
            const f32x2 r2ps = *reinterpret_cast<const f32x2*>(E + 8 * ol + 4);  // third ToRGB weight, post scale
            const float r2 = r2ps.x;
            const f32x4 raw = (z2 + zo) * row_sign + z1;
            const f32x4 tt = raw * e[0] + (nz * nw + e[1]);
            const f32x4 v4 = __builtin_elementwise_max(tt, tt * slope);
            if (p.rgb) {
                rgbv[0] = v4 * e[2] + rgbv[0];
                rgbv[1] = v4 * e[3] + rgbv[1];
                rgbv[2] = v4 * r2 + rgbv[2];
            }
#ifdef MAUA_DEVICE_PASS
            if (store_feat)
                buffer_store_b128_sgpr_offset(__builtin_bit_cast(u32x4, v4 * r2ps.y), y_base, pix_off * 4u, (unsigned)ol * plane_b);
#endif
        }
    }
    if (!p.rgb || W2D_ABL(256)) return;
    // ---- fused ToRGB: sum the channel groups through LDS, add bias and the 2x FIR-upsampled skip image, store
    __syncthreads();
    {
        float* rp = Rr + ((cg * 2 + cr) * NPOS + cp) * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x4*>(rp + 4 * c) = rgbv[c];
    }
    __syncthreads();
    if (cg != 0) return;
#pragma unroll
    for (int g2 = 1; g2 < CG; ++g2) {
        const float* rp = Rr + ((g2 * 2 + cr) * NPOS + cp) * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgbv[c] += *reinterpret_cast<const f32x4*>(rp + 4 * c);
    }
    if (p.rgb == 3) {  // several m-tiles per pixel: leave this tile's partial sums; maua_torgb_f32 adds them up (+ bias, skip)
        float* part = p.rgb_out + ((size_t)b0 * 3 * p.m_tiles + 3 * mt_id) * plane + pix_off;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<f32x4*>(part + (size_t)c * plane) = rgbv[c];
        return;
    }
    const size_t rgb_plane = plane;
    float* rgb_img = p.rgb_out + (size_t)b0 * 3 * rgb_plane;
    f32x4 outc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) outc[c] = rgbv[c] + p.rgb_bias[c];
    if (p.rgb_skip) {
        const int ty_ = (oy & 1) ? 2 : 3;
        float kt[2][4];  // the two live tap rows
#pragma unroll
        for (int qy = 0; qy < 2; ++qy)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) kt[qy][t4] = p.rgb_k4[(ty_ - 2 * qy) * 4 + t4] * wy[qy];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int k0 = (px + 1) >> 1;              // first live source column of this pixel
            const int t0 = (px & 1) ? 2 : 3;           // its tap; the second live column uses tap t0 - 2
#pragma unroll
            for (int qy = 0; qy < 2; ++qy) {
                const float w0 = kt[qy][t0] * wx[k0], w1 = kt[qy][t0 - 2] * wx[k0 + 1];
#pragma unroll
                for (int c = 0; c < 3; ++c) outc[c][px] = fmaf(w0, sv[c][qy][k0], fmaf(w1, sv[c][qy][k0 + 1], outc[c][px]));
            }
        }
    }
    if (p.rgb_u8) {  // fused frame epilogue: clamp(-1, 1), (x + 1) * 127.5, truncating cast; 12 bytes = three dword stores
        uint32_t pix[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            pix[px] = 0u;
#pragma unroll
            for (int c = 0; c < 3; ++c) pix[px] |= (uint32_t)((fminf(fmaxf(outc[c][px], -1.f), 1.f) + 1.f) * 127.5f) << (8 * c);
        }
        uint32_t* fw = reinterpret_cast<uint32_t*>(p.rgb_u8 + ((size_t)b0 * rgb_plane + pix_off) * 3);
        fw[0] = pix[0] | (pix[1] << 24);
        fw[1] = (pix[1] >> 8) | (pix[2] << 16);
        fw[2] = (pix[2] >> 16) | (pix[3] << 8);
        if (!p.rgb_out) return;  // (both given: the fp32 planes are written as well — the parity tests' tap)
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x4*>(rgb_img + (size_t)c * rgb_plane + pix_off) = outc[c];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 32-output-channel workgroups in which every wave keeps ALL 24 frequencies of its own 16 positions (modconv_w2dw_kernel).
//
// The y-frequency split above buys its small accumulator footprint with an epilogue in which the four waves' partial rows meet in
// LDS (two barriers and an exchange pass per 16 channels) and every thread then handles 4 pixels of 16 / CG channels: on the 32-channel
// layers (8 K steps) that epilogue costs as many VALU issue cycles as the whole K loop costs matrix cycles, and fp32 VALU and fp32
// matrix instructions share the datapath (MFMA-busy 45 %).  Here a workgroup is 32 channels x 4 n-tiles (16 rows x 32 pixels) and WAVE w
// OWNS n-tile w: 4 x 6 frequencies x 2 m-tiles = 48 accumulator tiles (192 registers, two workgroups per CU).  Per K step a wave forms
// all four row combinations of its window itself (64 VALU for 48 MFMAs: the same 1.33 per MFMA as the split kernel's 16 for 12) and reads
// the whole 12 KB weight tile; the window rows are fetched lazily into two register slots (rows 0, 2 -> f0; row 1 -> f1, f2; row 3 ->
// f3), each fetch hidden behind the previous frequency's 12 MFMAs.  The epilogue runs entirely in registers: A_x^T and A_y^T per lane,
// tail, the ToRGB products of the lane's 8 channels, a two-step butterfly over the four K lane groups, then bias / up-sampled skip /
// uint8 frames by the lanes of groups 0 and 1 (one output row of the position each) — no barrier after the K loop.  Same packed weight
// (TM = 2 layout), same staged patch and DMA scheme as modconv_w2d_kernel<2, 4>.
constexpr int WW_TN = 4, WW_BM = 32;
constexpr int WW_A_FLOATS = 24 * W2D_CC * WW_BM;

template <bool PLUS, bool PRE = false>
__device__ __forceinline__ void ww_transform(const f32x2 (&a)[4], const f32x2 (&b)[4], float sc, float m5, float (&bv)[6]) {
    // D = a +- b (the F(2,3) row combination), then B_x^T of F(4,3) on register pairs exactly as in modconv_w2d_kernel
    f32x2 D12, D34, ba, ec, b12, b34;
    float d0, d5, t0, t5, u0, u5;
    const f32x2 sc2 = f32x2{sc, sc};
    if constexpr (PLUS) {
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(D12) : "v"(a[1]), "v"(b[1]));
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(D34) : "v"(a[2]), "v"(b[2]));
        d0 = a[0].y + b[0].y, d5 = a[3].x + b[3].x;
    } else {
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(D12) : "v"(a[1]), "v"(b[1]));
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(D34) : "v"(a[2]), "v"(b[2]));
        d0 = a[0].y - b[0].y, d5 = a[3].x - b[3].x;
    }
    asm("v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]" : "=v"(ba) : "v"(D12), "v"(D34));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(ec) : "v"(D34), "v"(D12));
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(b12) : "v"(ba));
    asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(b34) : "v"(ec));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(m5), "v"(D12.y), "v"(D34.y));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t5) : "v"(m5), "v"(D34.x), "v"(d5));
    asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(u0) : "v"(d0), "v"(t0));
    asm("v_fma_f32 %0, 4.0, %1, %2" : "=v"(u5) : "v"(D12.x), "v"(t5));
    if constexpr (!PRE) {  // (PRE: x arrives multiplied by this layer's styles)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(b12) : "v"(b12), "v"(sc2));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(b34) : "v"(b34), "v"(sc2));
        u0 *= sc, u5 *= sc;
    }
    bv[0] = u0, bv[5] = u5;
    bv[1] = b12.x, bv[2] = b12.y, bv[3] = b34.x, bv[4] = b34.y;
}

template <bool PRE = false>
__global__ __launch_bounds__(256, 2) void modconv_w2dw_kernel(W2dArgs p) {
    constexpr int TN = WW_TN, BM = WW_BM;
    constexpr int TH = 4 * TN, PH = TH + 2;
    constexpr int PSTRIDE = w2d_pstride(TN);
    constexpr int NQ = w2d_dma_per_channel(TN);
    constexpr int A_FLOATS = WW_A_FLOATS;
    constexpr int A_PER_WAVE = A_FLOATS / 256 / 4;
    constexpr int PBUF = w2d_pbuf(TN);
    static_assert(NQ == 4 && A_FLOATS % 1024 == 0, "one patch row group and whole weight pieces per wave");
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    float* lds = lds_all;
    float* Ps = lds + 2 * A_FLOATS;       // [2][PBUF]
    float* Ss = Ps + 2 * PBUF;            // [Cin] styles of this image
    float* E = Ss + ((p.Cin + 3) & ~3);   // [BM / 2][12] epilogue constants of channel pairs

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's n-tile
    const int j = lane & 15, kq = lane >> 4;
    const int jx = j & 7, jy = j >> 3;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int mt_id = t % p.m_tiles;
    t /= p.m_tiles;
    const int tile_x = t % p.tiles_x;
    t /= p.tiles_x;
    const int tile_y = t % p.tiles_y;
    const int b0 = t / p.tiles_y;
    const int ty0 = tile_y * TH, tx0 = tile_x * 32;
    const int m0 = mt_id * BM;
    const size_t plane = (size_t)p.H * p.W;

    // patch DMA of this lane: wave w issues row group w of every channel (see modconv_w2d_kernel)
    unsigned rel_bytes;
    {
        const int pr = W2D_ROWS_PER_DMA * wv + lane / 12, sg = lane % 12;
        const int yy = ty0 + pr - 1, xx = tx0 - 4 + 4 * sg;
        const bool ok = sg < W2D_SEGS && pr < PH && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        rel_bytes = ok ? (unsigned)(yy * p.W + xx) * 4u : 0x80000000u;
    }
    const char* ximg = reinterpret_cast<const char*>(p.x + (size_t)b0 * p.Cin * plane);
    const size_t plane_bytes = plane * sizeof(float);
    (void)ximg, (void)plane_bytes, (void)rel_bytes;
#ifdef MAUA_DEVICE_PASS
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ximg), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wq), 0, 0x7fffffff, 0x00020000);
#endif
    auto issue = [&](int chunk, int buf) {
#ifdef MAUA_DEVICE_PASS
        const int wbase = (int)(((size_t)mt_id * p.n_chunks + chunk) * A_FLOATS * sizeof(float));
#pragma unroll
        for (int k = 0; k < A_PER_WAVE; ++k) {
            const int i = wv + 4 * k;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(lds + buf * A_FLOATS + i * 256), 16,
                                                     (i * 256 + lane * 4) * 4, wbase, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < W2D_CC; ++c) {
            float* dst = Ps + buf * PBUF + c * PSTRIDE + (c & 1) * W2D_ODD_SHIFT + wv * (W2D_ROWS_PER_DMA * W2D_PWS);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)rel_bytes,
                                                     (int)((size_t)(chunk * W2D_CC + c) * plane_bytes), 0, 0);
        }
#else
        (void)chunk, (void)buf;
#endif
    };

    // the first chunk's operands are requested BEFORE the per-image tables below are fetched: their global round trip (styles, demod,
    // bias, ToRGB weights) then runs under the DMA's instead of in front of it
    issue(0, 0);
    // (styles: the first 256 channels' load goes out here and is written below, next to the table — one round trip for both)
    const float* dummy = p.wq;  // (always valid)
    const float s_first = *((!PRE && tid < p.Cin) ? p.s + (size_t)b0 * p.s_stride + tid : dummy);
    const bool act = p.fuse_act != 0;
    const float act_gain = act ? 1.41421356237309515f : 1.f;
    const float* noise_base = p.noise;
    int64_t noise_bstride = p.noise_batch_stride;
    if (p.src) {
        noise_bstride = p.src->noise_stride[p.noise_slot];
        noise_base = p.src->noise[p.noise_slot];
        if (noise_base) noise_base += (int64_t)p.src->frame0 * noise_bstride;
    }
    const float nw = (act && noise_base) ? p.noise_w[0] * act_gain : 0.f;
    if (tid < BM) {
        // ALL table loads go out together, unconditionally: absent operands (no demodulation, no bias, no ToRGB) read a valid dummy
        // address and are masked by a wave-uniform select.  Written as `if (p.d) gain *= p.d[..]` etc. the compiler emitted a branch and
        // a full wait per operand: four dependent global round trips (~3 us) in front of every workgroup's K loop.
        const int i = tid, o = m0 + i;
        const float dv = *(p.d ? p.d + (size_t)b0 * p.Cout + o : dummy);
        const float bv_ = *((act && p.bias) ? p.bias + o : dummy);
        const float sv_ = *(p.rgb ? p.rgb_s + (size_t)b0 * p.s_stride + o : dummy);
        const float w0 = *(p.rgb ? p.rgb_w + 0 * p.Cout + o : dummy), w1 = *(p.rgb ? p.rgb_w + 1 * p.Cout + o : dummy);
        const float w2 = *(p.rgb ? p.rgb_w + 2 * p.Cout + o : dummy);
        const float pv_ = *(p.post_s ? p.post_s + (size_t)b0 * p.s_stride + o : dummy);
        const float gain = p.wscale * act_gain * (p.d ? dv : 1.f);
        const float ms = p.rgb ? p.rgb_wscale * sv_ : 0.f;
        // channel PAIRS side by side (the epilogue works on register pairs): [pair][gain x2 | bias x2 | rgb0 x2 | rgb1 x2 | rgb2 x2 | post scale x2]
        float* ep = E + 12 * (i >> 1) + (i & 1);
        ep[0] = gain, ep[2] = (act && p.bias) ? bv_ * act_gain : 0.f, ep[4] = ms * w0, ep[6] = ms * w1, ep[8] = ms * w2;
        ep[10] = p.post_s ? pv_ : 1.f;
    }
    if constexpr (!PRE) {
        if (tid < p.Cin) Ss[tid] = s_first;
        for (int e = tid + 256; e < p.Cin; e += 256) Ss[e] = p.s[(size_t)b0 * p.s_stride + e];
    }
    (void)s_first;

    f32x4 acc[4][6][2];  // [y-frequency][x-frequency][m-tile]; the first K step runs with C = 0
    float m5 = -5.f;
    asm("" : "+v"(m5));
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    // window of position (2 wv + jy, jx), channel kq: patch rows 2 (2 wv + jy) .. + 3, read as aligned 8-byte pairs from row float 4 jx + 2
    const unsigned w_addr = lds0 + (unsigned)(2 * A_FLOATS + kq * PSTRIDE + (kq & 1) * W2D_ODD_SHIFT + (2 * (2 * wv + jy)) * W2D_PWS + 4 * jx + 2) * 4u;
    const unsigned a_addr = lds0 + (unsigned)(kq * BM + 2 * j) * 4u;  // weight rows: ((fy * 6 + xf) * 4 + kq) * BM + (i16 * 2 + m-tile)
    unsigned s_addr = lds0 + (unsigned)(2 * A_FLOATS + 2 * PBUF + kq) * 4u;
    constexpr unsigned A_BUF_BYTES = A_FLOATS * 4u, P_BUF_BYTES = PBUF * 4u;
    constexpr int XF_BYTES = W2D_CC * BM * 4;
    constexpr int ROW_BYTES = W2D_PWS * 4;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int phase = 0; phase < 2; ++phase)
    for (int chunk = phase; chunk < (phase ? p.n_chunks : 1); ++chunk) {
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (chunk + 1 < p.n_chunks && !W2D_ABL(2)) issue(chunk + 1, cur ^ 1);
        const unsigned ap = a_addr + (cur ? A_BUF_BYTES : 0u), pw = w_addr + (cur ? P_BUF_BYTES : 0u);
        float sc = 1.f;
        if constexpr (!PRE) {
            sc = lds_read32(s_addr);
            s_addr += W2D_CC * 4u;
        }
        f32x2 SA[4], SB[4], a2[2];
        float bv[6];
        auto read_row = [&](f32x2(&dst)[4], auto r_c) {
            constexpr int o = decltype(r_c)::value * ROW_BYTES;
            dst[0] = lds_read64<o>(pw), dst[1] = lds_read64<o + 8>(pw), dst[2] = lds_read64<o + 16>(pw), dst[3] = lds_read64<o + 24>(pw);
        };
        // the 12 matrix instructions of y-frequency f; the weight row of the next (f, xf) is read one step ahead.  FIRST_BEHIND: LDS
        // operations issued behind this block's first weight row before its wait (4 when a window row went out in between)
        auto block = [&](auto f_c, auto first_c) {
            constexpr int f = decltype(f_c)::value, first_behind = decltype(first_c)::value;
            static_for<0, 6>([&](auto xf_c) {
                constexpr int xf = decltype(xf_c)::value, idx = f * 6 + xf;
                if constexpr (idx + 1 < 24) a2[(idx + 1) & 1] = lds_read64<(idx + 1) * XF_BYTES>(ap);
                constexpr int behind = (idx + 1 < 24 ? 1 : 0) + (xf == 0 ? first_behind : 0);
                lds_wait<behind>(a2[idx & 1]);
                if constexpr (W2D_ABL(1)) {
                    if (!phase) acc[f][xf][0] = acc[f][xf][1] = zero4;
                    asm volatile("" : "+v"(acc[f][xf][0]), "+v"(acc[f][xf][1]) : "v"(a2[idx & 1]), "v"(bv[xf]));
                } else {
                    acc[f][xf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[idx & 1].x, bv[xf], phase ? acc[f][xf][0] : zero4, 0, 0, 0);
                    acc[f][xf][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[idx & 1].y, bv[xf], phase ? acc[f][xf][1] : zero4, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using I4 = std::integral_constant<int, 4>;
        read_row(SA, I0{});
        read_row(SB, I2{});
        a2[0] = lds_read64<0>(ap);
        if constexpr (PRE)
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(SA[0]), "+v"(SA[1]), "+v"(SA[2]), "+v"(SA[3]), "+v"(SB[0]), "+v"(SB[1]), "+v"(SB[2]), "+v"(SB[3]));
        else
        asm volatile("s_waitcnt lgkmcnt(1)"
                     : "+v"(sc), "+v"(SA[0]), "+v"(SA[1]), "+v"(SA[2]), "+v"(SA[3]), "+v"(SB[0]), "+v"(SB[1]), "+v"(SB[2]), "+v"(SB[3]));
        ww_transform<false, PRE>(SA, SB, sc, m5, bv);  // f0: d0 - d2
        __builtin_amdgcn_sched_barrier(0);
        read_row(SA, I1{});                        // (row 0 is dead)
        block(I0{}, I4{});
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(SA[0]), "+v"(SA[1]), "+v"(SA[2]), "+v"(SA[3]));
        ww_transform<true, PRE>(SA, SB, sc, m5, bv);   // f1: d1 + d2
        __builtin_amdgcn_sched_barrier(0);
        block(I1{}, I0{});
        ww_transform<false, PRE>(SB, SA, sc, m5, bv);  // f2: d2 - d1
        __builtin_amdgcn_sched_barrier(0);
        read_row(SB, I3{});                        // (row 2 is dead)
        block(I2{}, I4{});
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(SB[0]), "+v"(SB[1]), "+v"(SB[2]), "+v"(SB[3]));
        ww_transform<false, PRE>(SA, SB, sc, m5, bv);  // f3: d1 - d3
        __builtin_amdgcn_sched_barrier(0);
        block(I3{}, I0{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    if (W2D_ABL(8)) {
        if (p.B < 0) {  // never true: keeps the accumulators alive in builds without the epilogue
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int a = 0; a < 6; ++a) sum += acc[f][a][0] + acc[f][a][1];
            *reinterpret_cast<f32x4*>(p.y + tid * 4) = sum;
        }
        return;
    }
    // ---- epilogue, in registers.  This lane: position (2 wv + jy, jx) = output rows oy0, oy0 + 1, columns ox .. ox + 3; channels
    // 16 m + 4 kq + v of the workgroup's 32
    const int oy0 = ty0 + 2 * (2 * wv + jy), ox = tx0 + 4 * jx;
    const unsigned pix0 = (unsigned)oy0 * (unsigned)p.W + (unsigned)ox;
    f32x4 nzw[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (nw != 0.f) {
        const float* nb = noise_base + (size_t)b0 * noise_bstride + pix0;
        nzw[0] = *reinterpret_cast<const f32x4*>(nb) * nw;
        nzw[1] = *reinterpret_cast<const f32x4*>(nb + p.W) * nw;
    }
    const float slope = act ? 0.2f : 1.f;
    const bool store_feat = p.rgb != 2;
    const unsigned plane_b = (unsigned)plane * 4u;
    const float* y_base = p.y + ((size_t)b0 * p.Cout + m0) * plane;  // (uniform: the store builds its buffer descriptor from it)
    (void)y_base;
    const unsigned y_voff = pix0 * 4u + (unsigned)(4 * kq) * plane_b;
    (void)y_voff;
    const float* Elane = E + 12 * (2 * kq);
    // The arithmetic runs on CHANNEL PAIRS (v = 2 vp, 2 vp + 1 of an accumulator tile are neighbouring registers): v_pk_* instructions do
    // two channels per issue slot.  ToRGB partial sums per pair member; the members are added before the butterfly.
    f32x2 nzp[2][4];  // noise * weight of this lane's 2 x 4 pixels, duplicated into pairs
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int px = 0; px < 4; ++px) nzp[r][px] = f32x2{nzw[r][px], nzw[r][px]};
    f32x2 rgb2[3][2][4];  // [colour][row][pixel], first written by the first channel pair

    // 2x FIR-upsampled skip image (see modconv_w2d_kernel): fetched by the lanes that finish the pixels (K lane groups 0 and 1: output
    // row oy0 + kq), once the first m-tile's accumulators are dead
    const int cr = kq & 1;
    const int oy = oy0 + cr;
    const int sh = p.H >> 1, sw = p.W >> 1;
    const bool finisher = kq < 2;
    const bool want_skip = (p.rgb == 1 || p.rgb == 2) && p.rgb_skip && finisher;
    float sv[3][2][4], wy[2], wx[4];
    auto load_skip = [&]() {
        const float* skip_img = p.rgb_skip + (size_t)b0 * 3 * sh * sw;
        const int iy0 = (oy - 1) >> 1;
        int rowc[2], colc[4];
#pragma unroll
        for (int qy = 0; qy < 2; ++qy) {
            const int yy = iy0 + qy;
            wy[qy] = (yy >= 0 && yy < sh) ? 1.f : 0.f;
            rowc[qy] = min(max(yy, 0), sh - 1);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = (ox >> 1) - 1 + k;
            wx[k] = (xx >= 0 && xx < sw) ? 1.f : 0.f;
            colc[k] = min(max(xx, 0), sw - 1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[c][qy][k] = skip_img[(unsigned)((c * sh + rowc[qy]) * sw + colc[k])];
    };

    const f32x2 slope2 = f32x2{slope, slope};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m == 1 && want_skip) load_skip();
#pragma unroll
        for (int vp = 0; vp < 2; ++vp) {
            auto pair_of = [&](const f32x4& a) { return vp ? f32x2{a[2], a[3]} : f32x2{a[0], a[1]}; };
            // A_x^T per y-frequency (y0 = m0+m1+m2+m3+m4, y1 = (m1-m2) + 2(m3-m4), y2 = (m1+m2) + 4(m3+m4), y3 = (m1-m2) + 8(m3-m4) + m5)
            // accumulated straight into A_y^T of F(2,3): row 0 = Z0 + Z1 + Z2, row 1 = Z1 - Z2 - Z3
            f32x2 raw0[4], raw1[4];
            if constexpr (W2D_ABL(2048)) {  // (ablation: no inverse transform — every accumulator is still read once (a dead accumulator would take its
                // matrix instructions out of the K loop): 16 plain adds instead of the 56 operations of A_x^T and A_y^T)
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    raw0[px] = (pair_of(acc[0][px][m]) + pair_of(acc[2][px][m])) + pair_of(acc[0][4 + (px & 1)][m]);
                    raw1[px] = (pair_of(acc[1][px][m]) + pair_of(acc[3][px][m])) + pair_of(acc[1][4 + (px & 1)][m]);
                }
#pragma unroll
                for (int px = 0; px < 2; ++px) raw0[px] += pair_of(acc[2][4 + px][m]), raw1[px] += pair_of(acc[3][4 + px][m]);
            } else
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const f32x2 M0 = pair_of(acc[f][0][m]), M1 = pair_of(acc[f][1][m]), M2 = pair_of(acc[f][2][m]);
                const f32x2 M3 = pair_of(acc[f][3][m]), M4 = pair_of(acc[f][4][m]), M5 = pair_of(acc[f][5][m]);
                const f32x2 s12 = M1 + M2, d12 = M1 - M2, s34 = M3 + M4, d34 = M3 - M4;
                const f32x2 Zf[4] = {(M0 + s12) + s34, d34 * 2.f + d12, s34 * 4.f + s12, (d34 * 8.f + d12) + M5};
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    if (f == 0) raw0[px] = Zf[px];
                    if (f == 1) raw0[px] += Zf[px], raw1[px] = Zf[px];
                    if (f == 2) raw0[px] += Zf[px], raw1[px] -= Zf[px];
                    if (f == 3) raw1[px] -= Zf[px];
                }
            }
            const float* ep = Elane + 12 * (8 * m + vp);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(ep), r01 = *reinterpret_cast<const f32x4*>(ep + 4);
            const f32x4 r2ps = *reinterpret_cast<const f32x4*>(ep + 8);
            const f32x2 r2p = f32x2{r2ps[0], r2ps[1]}, post2 = f32x2{r2ps[2], r2ps[3]};
            const f32x2 gain2 = f32x2{gb[0], gb[1]}, bias2 = f32x2{gb[2], gb[3]};
            const f32x2 w0 = f32x2{r01[0], r01[1]}, w1 = f32x2{r01[2], r01[3]};
            f32x2 val[2][4];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                if constexpr (W2D_ABL(8192)) {  // (ablation: no tail)
                    val[0][px] = raw0[px], val[1][px] = raw1[px];
                    continue;
                }
                const f32x2 t0 = raw0[px] * gain2 + (nzp[0][px] + bias2), t1 = raw1[px] * gain2 + (nzp[1][px] + bias2);
                val[0][px] = __builtin_elementwise_max(t0, t0 * slope2);
                val[1][px] = __builtin_elementwise_max(t1, t1 * slope2);
            }
            if constexpr (W2D_ABL(4096)) {  // (ablation: the values stay live through one add each instead of three products)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int px = 0; px < 4; ++px) rgb2[0][r][px] = (m == 0 && vp == 0) ? val[r][px] : rgb2[0][r][px] + val[r][px];
            }
            if (p.rgb && !W2D_ABL(4096)) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        if (m == 0 && vp == 0) {
                            rgb2[0][r][px] = val[r][px] * w0, rgb2[1][r][px] = val[r][px] * w1, rgb2[2][r][px] = val[r][px] * r2p;
                        } else {
                            rgb2[0][r][px] = val[r][px] * w0 + rgb2[0][r][px];
                            rgb2[1][r][px] = val[r][px] * w1 + rgb2[1][r][px];
                            rgb2[2][r][px] = val[r][px] * r2p + rgb2[2][r][px];
                        }
                    }
            }
#ifdef MAUA_DEVICE_PASS
            if (store_feat) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ol = 16 * m + 2 * vp + h;  // + 4 kq through y_voff
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const f32x4 row = f32x4{val[r][0][h], val[r][1][h], val[r][2][h], val[r][3][h]} * post2[h];
                        buffer_store_b128_sgpr_offset(__builtin_bit_cast(u32x4, row), y_base, y_voff + (unsigned)(r * p.W) * 4u, (unsigned)ol * plane_b);
                    }
                }
            }
#endif
        }
    }
    if constexpr (W2D_ABL(4096)) {
        if (p.B < 0) {  // never true
            f32x2 sum = f32x2{0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int px = 0; px < 4; ++px) sum += rgb2[0][r][px];
            *reinterpret_cast<f32x2*>(p.y + tid * 2) = sum;
        }
        return;
    }
    if (!p.rgb) return;
    // ---- fused ToRGB: lanes j, j + 16, j + 32, j + 48 hold disjoint channels of one position.  Step 1 (xor 32) sums both rows of the
    // pairs (kq, kq ^ 2); step 2 (xor 16) is a reduce-scatter: a lane keeps the row it will finish (row kq & 1) and sends the other
    f32x4 outc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            float a = rgb2[c][0][px].x + rgb2[c][0][px].y, b = rgb2[c][1][px].x + rgb2[c][1][px].y;
            a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 32);
            const float mine = cr ? b : a, other = cr ? a : b;
            outc[c][px] = mine + __shfl_xor(other, 16) + p.rgb_bias[c];
        }
    if (!finisher) return;
    const unsigned pix_off = pix0 + (unsigned)cr * (unsigned)p.W;
    const size_t rgb_plane = plane;
    if (p.rgb_skip) {
        const int ty_ = (oy & 1) ? 2 : 3;
        float kt[2][4];
#pragma unroll
        for (int qy = 0; qy < 2; ++qy)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) kt[qy][t4] = p.rgb_k4[(ty_ - 2 * qy) * 4 + t4] * wy[qy];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const int k0 = (px + 1) >> 1;
            const int t0 = (px & 1) ? 2 : 3;
#pragma unroll
            for (int qy = 0; qy < 2; ++qy) {
                const float w0 = kt[qy][t0] * wx[k0], w1 = kt[qy][t0 - 2] * wx[k0 + 1];
#pragma unroll
                for (int c = 0; c < 3; ++c) outc[c][px] = fmaf(w0, sv[c][qy][k0], fmaf(w1, sv[c][qy][k0 + 1], outc[c][px]));
            }
        }
    }
    if (p.rgb_u8) {
        uint32_t pix[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            pix[px] = 0u;
#pragma unroll
            for (int c = 0; c < 3; ++c) pix[px] |= (uint32_t)((fminf(fmaxf(outc[c][px], -1.f), 1.f) + 1.f) * 127.5f) << (8 * c);
        }
        uint32_t* fw = reinterpret_cast<uint32_t*>(p.rgb_u8 + ((size_t)b0 * rgb_plane + pix_off) * 3);
        fw[0] = pix[0] | (pix[1] << 24);
        fw[1] = (pix[1] >> 8) | (pix[2] << 16);
        fw[2] = (pix[2] >> 16) | (pix[3] << 8);
        if (!p.rgb_out) return;
    }
    float* rgb_img = p.rgb_out + (size_t)b0 * 3 * rgb_plane;
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<f32x4*>(rgb_img + (size_t)c * rgb_plane + pix_off) = outc[c];
}

size_t w2dw_lds_bytes(int cin) {
    return sizeof(float) * ((size_t)2 * WW_A_FLOATS + (size_t)2 * w2d_pbuf(WW_TN) + (size_t)((cin + 3) & ~3) + (size_t)8 * WW_BM);
}

// wq (the LDS tile image): [m_tile][chunk][fy 4][xf 6][kq 4][BM physical column]
__global__ __launch_bounds__(256) void pack_weight_wino2d_kernel(const float* __restrict__ w, float* __restrict__ wq, int cout,
                                                                 int cin, int tm) {
    const int bm = 16 * tm;
    const int n_chunks = cin / W2D_CC;
    const int64_t total = (int64_t)(cout / bm) * n_chunks * W2D_CC * bm;  // one thread per (m_tile, chunk, kq, logical column)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx;
        const int ol = (int)(r % bm);
        r /= bm;
        const int kq = (int)(r % W2D_CC);
        r /= W2D_CC;
        const int chunk = (int)(r % n_chunks);
        const int mtile = (int)(r / n_chunks);
        const int o = mtile * bm + ol, i = chunk * W2D_CC + kq;
        const float* g = w + ((size_t)o * cin + i) * 9;
        // G_y (F(2,3)): rows {g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2};  G_x (F(4,3)): {g0/4, -(g0+g1+g2)/6, -(g0-g1+g2)/6,
        // (g0+2g1+4g2)/24, (g0-2g1+4g2)/24, g2} — the constants of maua_pack_weight_wino_f32 / _wino43_f32
        float gy[4][3];
        for (int kx = 0; kx < 3; ++kx) {
            const float g0 = g[0 * 3 + kx], g1 = g[1 * 3 + kx], g2 = g[2 * 3 + kx];
            gy[0][kx] = g0, gy[1][kx] = 0.5f * (g0 + g1 + g2), gy[2][kx] = 0.5f * (g0 - g1 + g2), gy[3][kx] = g2;
        }
        const int col = w2d_col(tm, ol / 16, ol % 16, kq);
        float* dst = wq + ((size_t)mtile * n_chunks + chunk) * (24 * W2D_CC * bm);
        for (int f = 0; f < 4; ++f) {
            const float g0 = gy[f][0], g1 = gy[f][1], g2 = gy[f][2];
            const float u[6] = {g0 * 0.25f,
                                -(g0 + g1 + g2) * (1.f / 6.f),
                                -(g0 - g1 + g2) * (1.f / 6.f),
                                (g0 + 2.f * g1 + 4.f * g2) * (1.f / 24.f),
                                (g0 - 2.f * g1 + 4.f * g2) * (1.f / 24.f),
                                g2};
            for (int xf = 0; xf < 6; ++xf) dst[((f * 6 + xf) * W2D_CC + kq) * bm + col] = u[xf];
        }
    }
}

size_t w2d_lds_bytes(int tm, int tn, int cin) {
    const int bm = 16 * tm, npos = 16 * tn;
    const size_t main_loop = (size_t)2 * 24 * W2D_CC * bm + (size_t)2 * w2d_pbuf(tn) + (size_t)cin;
    const int cg = 256 / (2 * npos);
    const size_t epilogue = (size_t)4 * 16 * npos * 4 + (size_t)cg * 2 * npos * 12;
    return sizeof(float) * (((main_loop > epilogue ? main_loop : epilogue) + (size_t)8 * bm + 3) & ~(size_t)3);  // + the channel-constant table
}

char g_w2d_instance[64] = "";

template <int TM, int TN, int MINB = 2, bool PRE = false>
int w2d_launch_t(const W2dArgs& a, hipStream_t st) {
    auto kern = modconv_w2d_kernel<TM, TN, MINB, PRE>;
    static unsigned long long lds_ok = 0;  // per launcher: devices on which the attribute has been set (common.h)
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(kern), &lds_ok, 160 * 1024)) return rc;
    snprintf(g_w2d_instance, sizeof(g_w2d_instance), PRE ? "modconv_w2d_kernel<%d, %d, %d, true>" : "modconv_w2d_kernel<%d, %d, %d, false>", TM, TN, MINB);
    const int64_t blocks = (int64_t)a.B * a.tiles_y * a.tiles_x * a.m_tiles;
#ifdef MAUA_EXPERIMENTS  // occupancy probe (MAUA_W2D_LDS_PAD with an experiments build): extra dynamic LDS so that a CU holds one workgroup instead of two
    static const size_t lds_pad = getenv("MAUA_W2D_LDS_PAD") ? (size_t)atoi(getenv("MAUA_W2D_LDS_PAD")) : 0;
#else
    constexpr size_t lds_pad = 0;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), w2d_lds_bytes(TM, TN, a.Cin) + lds_pad, st, a);
    MAUA_LAUNCH_CHECK();
    return 0;
}

template <bool PRE>
int w2dw_launch(const W2dArgs& a, hipStream_t st) {
    static unsigned long long lds_ok = 0;  // per launcher: devices on which the attribute has been set (common.h)
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_w2dw_kernel<PRE>), &lds_ok, 160 * 1024)) return rc;
    snprintf(g_w2d_instance, sizeof(g_w2d_instance), PRE ? "modconv_w2dw_kernel<true>" : "modconv_w2dw_kernel<false>");
    const int64_t blocks = (int64_t)a.B * a.tiles_y * a.tiles_x * a.m_tiles;
    hipLaunchKernelGGL(modconv_w2dw_kernel<PRE>, dim3((unsigned)blocks), dim3(256), w2dw_lds_bytes(a.Cin), st, a);
    MAUA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Tile shape for a layer: TM (16-channel m-tiles per workgroup) and TN (16-position n-tiles); 0 = the layer does not qualify.
int maua_w2d_tiles(int cin, int cout, int h, int w, int* tm, int* tn) {
    if (cin % W2D_CC || w % 32 || cin <= 0 || cout <= 0) return 0;
    int m, n;
    // (the tile shape depends on the channel counts only: the packed weight of a layer serves every map size)
#ifdef MAUA_W2D_TM2_64  // (experiment: 64-channel layers as two 32-channel workgroups per tile on the wave-complete kernel)
    if (cout == 64) m = 2, n = MAUA_W2D_TN32;
    else
#endif
    if (cout % 64 == 0) m = 4, n = MAUA_W2D_TN64;
    else if (cout == 32) m = 2, n = MAUA_W2D_TN32;
    else return 0;
    if (h % (4 * n)) return 0;
    if (tm) *tm = m;
    if (tn) *tn = n;
    return 1;
}

const char* maua_w2d_last_instance() { return g_w2d_instance; }

int maua_w2d_launch(const float* x, const float* wq, const float* s, int s_stride, const float* d, float* y, int batch, int cin,
                    int cout, int h, int w, float wscale, int fuse_act, const float* noise, int64_t noise_batch_stride,
                    const float* noise_w, const float* bias, const float* rgb_w, const float* rgb_s, float rgb_wscale,
                    const float* rgb_bias, const float* rgb_skip, const float* rgb_k4, float* rgb_out, uint8_t* rgb_u8,
                    int rgb_mode, const maua_frame_source_t* src, int noise_slot, const float* post_s, void* stream) {
    int tm = 0, tn = 0;
    if (!maua_w2d_tiles(cin, cout, h, w, &tm, &tn)) return MAUA_EINVAL;
    if ((int64_t)cin * h * w * 4 > 0x7fffffffLL || (int64_t)24 * cin * cout * 4 > 0x7fffffffLL) return MAUA_EINVAL;  // descriptor ranges
    W2dArgs a{};
    a.x = x, a.wq = wq, a.s = s, a.d = d, a.noise = noise, a.noise_w = noise_w, a.bias = bias, a.y = y;
    a.rgb_w = rgb_w, a.rgb_s = rgb_s, a.rgb_bias = rgb_bias, a.rgb_skip = rgb_skip, a.rgb_k4 = rgb_k4, a.rgb_out = rgb_out;
    a.rgb_u8 = rgb_u8, a.post_s = post_s;
    a.B = batch, a.Cin = cin, a.Cout = cout, a.H = h, a.W = w, a.s_stride = s_stride, a.wscale = wscale, a.fuse_act = fuse_act;
    a.noise_batch_stride = noise_batch_stride;
    a.src = src, a.noise_slot = noise_slot;
    a.tiles_x = w / 32, a.tiles_y = h / (4 * tn), a.m_tiles = cout / (16 * tm), a.n_chunks = cin / W2D_CC;
    a.rgb = rgb_mode, a.rgb_wscale = rgb_wscale;
    if (rgb_mode == 3) {
        if (!fuse_act || !rgb_w || !rgb_s || !rgb_out) return MAUA_EINVAL;
    } else if (rgb_mode && (a.m_tiles != 1 || !fuse_act || !rgb_w || !rgb_s || !rgb_bias || (!rgb_out && !rgb_u8) || (rgb_skip && (!rgb_k4 || (h & 1) || (w & 1)))))
        return MAUA_ENOSYS;
    hipStream_t st = (hipStream_t)stream;
#ifndef MAUA_W2D_NO_WW
    if (tm == 2 && h % (4 * WW_TN) == 0 && rgb_mode != 3) {  // 32 output channels: the wave-complete kernel (16-row tiles)
        a.tiles_y = h / (4 * WW_TN);
        return s ? w2dw_launch<false>(a, st) : w2dw_launch<true>(a, st);
    }
#endif
    if (!s)  // x arrives multiplied by this layer's styles (the producer's post_s)
        return tm == 4 ? w2d_launch_t<4, MAUA_W2D_TN64, MAUA_W2D_MINB64, true>(a, st) : w2d_launch_t<2, MAUA_W2D_TN32, MAUA_W2D_MINB32, true>(a, st);
    return tm == 4 ? w2d_launch_t<4, MAUA_W2D_TN64, MAUA_W2D_MINB64>(a, st) : w2d_launch_t<2, MAUA_W2D_TN32, MAUA_W2D_MINB32>(a, st);
}

extern "C" int maua_pack_weight_wino2d_f32(const float* w, float* wq, int cout, int cin, void* stream) {
    if (!w || !wq || cout <= 0 || cin <= 0) return MAUA_EINVAL;
    int tm = 0, tn = 0;
    if (!maua_w2d_tiles(cin, cout, 32, 32, &tm, &tn)) return MAUA_EINVAL;
    const int64_t total = (int64_t)cout * cin;
    const int64_t blocks = ceil_div64(total, 256);
    hipLaunchKernelGGL(pack_weight_wino2d_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       w, wq, cout, cin, tm);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_modconv_w2d_ok(int cin, int cout, int h, int w) { return maua_w2d_tiles(cin, cout, h, w, nullptr, nullptr); }

extern "C" int maua_modconv_w2d_mtiles(int cin, int cout, int h, int w) {
    int tm = 0, tn = 0;
    return maua_w2d_tiles(cin, cout, h, w, &tm, &tn) ? cout / (16 * tm) : 0;
}


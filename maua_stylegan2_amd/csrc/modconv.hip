// ModulatedConv2d 3x3 on the gfx950 matrix cores (fp32-in / fp32-accumulate MFMA, exact fp32).
//
// Behavioural contract: /root/reference/models/stylegan2.py:217-254 (+ StyledConv tail :338-343).
// The reference materialises per-sample weights [B,Cout,Cin,3,3] and runs a grouped cuDNN conv.  Here:
//
//     y[b,o] = d[b,o] * sum_{i,tap} (wscale * W[o,i,tap]) * (s[b,i] * x[b,i, . + tap])
//
// i.e. input-scale -> SHARED-weight implicit GEMM -> output-demod, so one weight panel
// serves the whole batch of frames and stays L2-resident across pixel tiles.
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32:  M = Cout, N = pixels, K = Cin x taps.
//   A (weights)  : tap-major repack wp[tap][cin][cout_pad] (maua_pack_weight_f32) -> LDS As[tap][c][BM], lanes read
//                  consecutive cout -> conflict-free ds_read_b32 with immediate offsets.
//   B (features) : the workgroup stages ONE halo patch per 8-channel chunk (scaled by s[b,i] on the way in);
//                  the 9 taps are 9 shifted views of the same patch — no im2col replication, in HBM or in LDS.
//   C            : 64 accumulator VGPRs per wave; epilogue applies demod, noise, bias, leaky-ReLU*sqrt2 in-register
//                  and stores 128-byte row segments (lane = pixel column).
// Transposed (stride-2) convolution (:229-237) is evaluated polyphase: the 9 taps split into output parities
// (4/2/2/1 taps), every workgroup owns a tile of input positions and accumulates all four parities from the same
// staged patch — 9 MACs per input pixel per channel pair, exactly the reference's FLOPs, no zero-stuffing.
// Small feature maps (4^2..32^2) are weight-streaming bound: K is split across workgroups (deterministic two-pass
// split-K through a caller-owned workspace, reduced by reduce_tail_kernel together with the tail).
//
// Kernel modes (template MODE = the `up` argument of maua_modconv3x3_f32), all on the same tile / pipeline skeleton:
//   0  direct 3x3                      9 weight rows per channel, 1 accumulator tile per position group
//   1  transposed, polyphase           9 rows, 4 tiles (output parities); tiles = flat runs of the (H+1)x(W+1) grid
//   2  Winograd F(2,3) along x         12 rows (3 ky x 4 frequencies), positions = output pairs, 4 tiles
//   3  Winograd F(4,3) along x         18 rows (3 ky x 6 frequencies), positions = output quads, 6 tiles
//   4  transposed + F(2,2) on the even x-phase   12 rows, positions = position pairs, 10 tiles
// Both operands reach LDS by global_load_lds DMA, double buffered, one barrier per K chunk; the style scale of the
// input channel is applied when the B operand is read (a register-staged, pre-scaled patch remains for tiles that hold
// several images and for the 32-channel F(2,3) config).  Measured rule of this chip that shapes everything above: VALU /
// SALU instructions of co-resident waves do NOT hide under the 64-cycle MFMAs (profiles/r01_pmc_modconv.md) — fewer
// MFMAs per output (Winograd) and fewer non-MFMA instructions are what pay, not occupancy or prefetch depth.
#include "common.h"

#include <cstdio>
#include <type_traits>

// the buffer-descriptor builtins (MUBUF `buffer_load ... lds`) only exist in the device pass
#if defined(__HIP_DEVICE_COMPILE__)
#define MAUA_DEVICE_PASS 1
#endif

// Ablation / A-B switches only exist in -DMAUA_EXPERIMENTS builds (tools/build_exp.sh): the product kernels carry no debug
// branches and the product library exports no tuning entry.
#ifdef MAUA_EXPERIMENTS
#define MAUA_DBG(mask) ((g.debug & (mask)) != 0)
#define MAUA_CFG(mask) ((g_conv_cfg & (mask)) != 0)
#else
#define MAUA_DBG(mask) false
#define MAUA_CFG(mask) false
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x2 f32x2u __attribute__((aligned(4)));
typedef f32x4 f32x4u __attribute__((aligned(4)));

// channels per K chunk: the weight tile As[rows][CC][BM] is kept at <= 18..24 KB so that the double buffer fits 2-3x per CU
constexpr int chunk_channels(int bm, int bn = 0, int mode = 0) {
    if (mode == 3) return bm >= 128 ? 2 : 4;  // F(4,3): 18 weight rows per channel
    return (bm >= 128 || bn >= 512) ? 4 : 8;
}

struct ConvGeom {
    int B, Cin, Cout, CoutPad, H, W;  // input feature map
    int GH, GW;                       // per-image grid of tile positions (plain: H,W   up: H+1,W+1)
    int OH, OW;                       // output plane
    int lsw, lsh;                     // log2 sub-tile (one MFMA N=32 group) width / height, 2^(lsw+lsh) = 32
    int lnsx, lnsy;                   // log2 sub-tiles per tile along x / y
    int lni;                          // log2 images per tile
    int tiles_x, tiles_y, img_groups;
    int PH, PW, PWS, PSTRIDE;         // staged patch: rows, valid cols, LDS row stride, floats per channel (all images)
    int m_tiles, n_tiles;
    int splits, chunks_per_split, n_chunks;
    int s_stride;
    float wscale;
    int fuse_act;
    int64_t noise_batch_stride;
    int64_t ws_slab;  // floats per split slab
    int flat;         // transposed mode: tiles are runs of BN consecutive positions of the row-major (H+1)x(W+1) grid
    int rgb;          // fused ToRGB epilogue: 0 off, 1 on, 2 on and the feature map itself is not stored
    float rgb_wscale;
    int force_ws;     // the partial sums go to the workspace slabs also when K is not split (the low-resolution entries reduce them themselves)
#ifdef MAUA_EXPERIMENTS
    int debug;        // (maua_tuning_set key 1) 1 skip stores, 2 skip MFMA, 4 skip loads, 32 skip weight DMA, 64 skip patch loads,
                      // 128 fold the patch loads onto 4 KB per channel (always cache hits; wrong results).  Not a member of the product's struct.
#endif
};

struct ConvPtrs {
    const float* x;
    const float* wp;
    const float* s;
    const float* d;
    const float* noise;
    const float* noise_w;
    const float* bias;
    float* y;
    float* ws;
    // fused ToRGB (models/stylegan2.py:346-365) — only for single-M-tile plain configs
    const float* rgb_w;     // [3, Cout]
    const float* rgb_s;     // styles of the ToRGB layer, [B, s_stride] (already offset to the layer's slice)
    const float* rgb_bias;  // [3]
    const float* rgb_skip;  // [B, 3, H/2, W/2] or null
    const float* rgb_k4;    // 4x4 upsample taps
    float* rgb_out;         // [B, 3, H, W]
    uint8_t* rgb_u8;        // when set: the image leaves as uint8 NHWC frames [B, H, W, 3] (render.py:40-43) instead of rgb_out
    // frame source (include/maua_hip.h): when set, the noise maps are read from src->noise[noise_slot] at frame src->frame0
    const maua_frame_source_t* src;
    int noise_slot;
};

// UP = false : plain 3x3, pad 1.           per wave: TM x TN MFMA tiles.
// UP = true  : stride-2 transposed 3x3.    per wave: TM x (TN position groups x 4 output parities).
// FAST (Cin a multiple of the K chunk and the padded weight covers whole BM tiles — every layer of a real generator): both
// operands come in by LDS DMA (or, on the register-staged patch path, by unconditional loads masked by multiplication at
// LDS-write time), addressed as uniform base + 32-bit lane offset, which removes the exec-mask / 64-bit-address scalar work
// that dominated the short-K (32/64-channel) layers.  Everything else (ragged channel counts) takes the generic path.
// MODE 2 (WINO): plain 3x3 through Winograd F(2,3) along x — two adjacent output columns share four "frequency" products
//   m0 = (d0-d2) g0, m1 = (d1+d2)(g0+g1+g2)/2, m2 = (d2-d1)(g0-g1+g2)/2, m3 = (d1-d3) g2;  y0 = m0+m1+m2, y1 = m1-m2-m3
// so a pair of outputs costs 4 MFMA K-steps per (channel, ky) instead of 6: 1.5x fewer matrix-core cycles on the
// MFMA-bound >= 128-channel layers.  A "position" is an output PAIR; the four frequencies take the place of the four
// parities of the transposed mode (same accumulator layout), weights are pre-transformed (pack_weight_wino_kernel),
// the B operand is formed from two 8-byte LDS reads of the ordinary patch, the epilogue undoes the transform in
// registers and stores 8 bytes per lane.  fp32 F(2,3) has transform constants {1, 1/2}: error stays at the 1e-6 level.
// MODE 3 / MODE 4: see the comments at their mfma_chunk branches (F(4,3) on output quads; transposed conv with F(2,2)).
template <int BM, int BN, int WM, int MODE, bool MULTI, bool FAST, int MAXP>
__global__ __launch_bounds__(256, ((MODE == 3 && BM >= 64) || MODE == 4) ? 2
                                            : ((BM / WM / 32) * (BN / (4 / WM) / 32) * (MODE ? 4 : 1) >= 8 || BM * BN > 8192 ? 2 : 3))
void modconv_mfma_kernel(ConvGeom g, ConvPtrs p) {
    constexpr bool UP = MODE == 1 || MODE == 4;
    constexpr bool UW = MODE == 4;  // transposed conv with Winograd F(2,2) on the even x-phase: positions are position PAIRS
    constexpr bool W23 = MODE == 2;             // Winograd F(2,3): positions are output pairs, 4 frequencies
    constexpr bool W43 = MODE == 3;             // Winograd F(4,3): positions are output quads, 6 frequencies
    constexpr bool WINO = W23 || W43;
    constexpr int WX = W43 ? 4 : ((W23 || UW) ? 2 : 1);  // patch columns advanced per position
    constexpr int NFREQ = W43 ? 6 : 4;
    constexpr int NTAPS = WINO ? 3 * NFREQ : (UW ? 12 : 9);  // weight rows per channel: 9 taps, or 3 ky x NFREQ frequencies
    constexpr int CC = chunk_channels(BM, BN, MODE);
    constexpr int WN = 4 / WM;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int NPH = UW ? 10 : (UP ? 4 : (WINO ? NFREQ : 1));
    constexpr int A_FLOATS = NTAPS * CC * BM;
    constexpr int A_VEC_ITERS = (A_FLOATS / 4 + 255) / 256;
    constexpr int MAX_POS = MAXP;  // patch positions per thread (PSTRIDE <= 256 * MAX_POS)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: As[2][A_FLOATS] | Ps[2][CC * PSTRIDE]   (the generic path only uses buffer 0 of each)
    float* As = lds;
    float* Ps = lds + 2 * A_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;

    // ---- which tile
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int nt_id = t % g.n_tiles;
    t /= g.n_tiles;
    const int mt_id = t % g.m_tiles;
    const int split = t / g.m_tiles;
    int r = nt_id;
    const int tile_x = r % g.tiles_x;
    r /= g.tiles_x;
    const int tile_y = r % g.tiles_y;
    const int img_group = r / g.tiles_y;

    const int SW = 1 << g.lsw, SH = 1 << g.lsh;
    const int TWd = SW << g.lnsx, THt = SH << g.lnsy;
    const int NI = 1 << g.lni;
    const int ty0 = tile_y * THt, tx0 = tile_x * TWd, b0 = img_group * NI;
    const int m0 = mt_id * BM;
    const size_t plane_in = (size_t)g.H * g.W;

    // ---- per-thread patch positions (decoded once)
    int src_off[MAX_POS];    // offset of (b, ch 0, y, x) in x; -1 (slow path) / 0 (FAST) when outside the image
    float src_mask[MAX_POS]; // 1 inside the image, 0 outside (FAST path multiplies instead of branching)
    int sb_off[MAX_POS];     // b * s_stride
#pragma unroll
    for (int i = 0; i < MAX_POS; ++i) {
        const int pp = tid + i * 256;
        src_off[i] = FAST ? 0 : -1;
        src_mask[i] = 0.f;
        sb_off[i] = 0;
        if (pp < g.PSTRIDE) {
            const int per_img = g.PH * g.PWS;
            const int img = pp / per_img;
            const int rem = pp - img * per_img;
            const int pr = rem / g.PWS, pc = rem - pr * g.PWS;
            const int b = b0 + img;
            int yy = ty0 + pr - 1, xx = WX * tx0 + pc - 1;
            if (UP && g.flat) {
                // flat run: patch row 1 holds positions p0-1 .. p0+BN-1 of the pitch-(W+1) flattened input (column W and
                // row H are the zero padding), patch row 0 the same run one input row up (p - GW)
                // (UW: g.GW counts position pairs per row, the flattened input pitch is 2 * g.GW = W + 2)
                const int pitch = UW ? 2 * g.GW : g.GW;
                const int fi = WX * tx0 + pc - 1 - (1 - pr) * pitch;
                yy = fi >= 0 ? fi / pitch : -1;
                xx = fi - yy * pitch;
            }
            if (pc < g.PW && img < NI && b < g.B && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) {
                src_off[i] = ((b * g.Cin * g.H + yy) * g.W + xx);  // < 2^31: checked on the host
                src_mask[i] = 1.f;
                sb_off[i] = b * g.s_stride;
            }
        }
    }

    if (MAUA_DBG(128)) {
#pragma unroll
        for (int i = 0; i < MAX_POS; ++i) src_off[i] = src_off[i] < 0 ? src_off[i] : (src_off[i] & 0x3ff);
    }

    // ---- per-lane B-fragment base offsets (top-left of the 3x3 window of this lane's pixel, channel parity hi)
    int boff[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int sub = wn * TN + n;
        const int jx = l31 & (SW - 1), jy = l31 >> g.lsw;
        const int sx = sub & ((1 << g.lnsx) - 1);
        const int sy = (sub >> g.lnsx) & ((1 << g.lnsy) - 1);
        const int img = sub >> (g.lnsx + g.lnsy);
        const int tyy = sy * SH + jy, txx = sx * SW + jx;
        boff[n] = hi * g.PSTRIDE + (img * g.PH + tyy) * g.PWS + WX * txx;
    }
    const int aoff = hi * BM + wm * (TM * 32) + l31;
    const int aoff2 = hi * BM + wm * (TM * 32) + 2 * l31;  // interleaved [lane][mt] layout of the Winograd weight packs

    f32x16 acc[TM][TN * NPH];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int n = 0; n < TN * NPH; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][n][e] = 0.f;

    const int chunk_begin = split * g.chunks_per_split;
    int chunk_end = chunk_begin + g.chunks_per_split;
    if (chunk_end > g.n_chunks) chunk_end = g.n_chunks;

    // Software pipeline (issue-early / write-late): the global loads of chunk c+1 are issued right after the barrier
    // that publishes chunk c and stay in flight underneath chunk c's 36*TM*TN*NPH' MFMAs; they are only waited for
    // when their registers are written to LDS at the top of the next iteration.
    f32x4 av[A_VEC_ITERS];
    float pv[MAX_POS][CC];
    constexpr bool one_image = !MULTI;  // every patch position belongs to image b0: styles are block-uniform
    auto issue_loads = [&](int chunk) {  // generic path only (the FAST path below has its own DMA pipeline)
        const int c0 = chunk * CC;
        {
#pragma unroll
        for (int it = 0; it < A_VEC_ITERS; ++it) {
            const int f = tid + it * 256;  // float4 index in As
            const int row = f / (BM / 4);  // tap*CC + c
            const int col = (f - row * (BM / 4)) * 4;
            const int tap = row / CC, c = row - tap * CC;
            const int ch = c0 + c, o = m0 + col;
            av[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (f < A_FLOATS / 4 && ch < g.Cin && o < g.CoutPad)
                av[it] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)tap * g.Cin + ch) * g.CoutPad + o);
        }
#pragma unroll
        for (int i = 0; i < MAX_POS; ++i)
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const int ch = c0 + c;
                float v = 0.f;
                if (src_off[i] >= 0 && ch < g.Cin) {
                    v = p.x[(size_t)src_off[i] + ch * plane_in];
                    if (!one_image) v *= p.s[sb_off[i] + ch];
                }
                pv[i][c] = v;
            }
        }
    };
    auto write_lds = [&](int chunk) {
        const int c0 = chunk * CC;
        float sc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c)  // uniform address -> scalar loads
            sc[c] = (one_image && c0 + c < g.Cin && b0 < g.B) ? p.s[b0 * g.s_stride + c0 + c] : 1.f;
#pragma unroll
        for (int it = 0; it < A_VEC_ITERS; ++it) {
            const int f = tid + it * 256;
            if (f < A_FLOATS / 4) reinterpret_cast<f32x4*>(As)[f] = av[it];
        }
#pragma unroll
        for (int i = 0; i < MAX_POS; ++i) {
            const int pp = tid + i * 256;
            if (pp < g.PSTRIDE) {
#pragma unroll
                for (int c = 0; c < CC; ++c) Ps[c * g.PSTRIDE + pp] = pv[i][c] * sc[c];
            }
        }
    };

    // ---- MFMA phase over one staged chunk
    // Sc != nullptr: the staged patch holds RAW features (DMA path) and the style of channel 2q+hi is applied to the B
    // operand here, one VALU multiply per operand next to a 64-cycle MFMA.
    auto mfma_chunk = [&](const float* __restrict__ Ac, const float* __restrict__ Pc, const float* __restrict__ Sc) {
        if (UW) {
            // Polyphase transposed conv, x direction through F(2,2): for the position pair (p, p+1) with inputs
            // d0 = x[p-1], d1 = x[p], d2 = x[p+1] of one input row and the kernel row (g0, g1, g2):
            //   even columns  e_p = g0 d1 + g2 d0,  e_p+1 = g0 d2 + g2 d1   =  (m0 + m1, m1 + m2) with
            //                 m0 = g2 (d0 - d1),  m1 = (g0 + g2) d1,  m2 = g0 (d2 - d1)            (3 products instead of 4)
            //   odd columns   o_p = g1 d1,  o_p+1 = g1 d2
            // 5 MFMA K-steps per (kernel row, channel pair) instead of 6.  Kernel row 0 / 1 read input row y (output row
            // parity 0 / 1), kernel row 2 reads input row y-1 (parity 0): accumulator slots [parity][m0, m1, m2, o_p, o_p+1].
            // Weight rows per kernel row (maua_pack_weight_upwino_f32): g2, g0 + g2, g0, g1.
#pragma unroll
            for (int q = 0; q < CC / 2; ++q) {
                float tb[2][TN][4];  // [patch row][n][d0-d1, d1, d2-d1, d2]
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        const float* dp = Pc + 2 * q * g.PSTRIDE + boff[n] + r * g.PWS;  // even float offset
                        const f32x2 d01 = *reinterpret_cast<const f32x2*>(dp);
                        const float d2 = dp[2];
                        tb[r][n][0] = d01.x - d01.y;
                        tb[r][n][1] = d01.y;
                        tb[r][n][2] = d2 - d01.y;
                        tb[r][n][3] = d2;
                        if (Sc) {
                            const float sc = Sc[2 * q + hi];
#pragma unroll
                            for (int k = 0; k < 4; ++k) tb[r][n][k] *= sc;
                        }
                    }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = ky == 2 ? 0 : 1;       // patch row: 1 = input row y, 0 = input row y-1
                    const int base = ky == 1 ? 5 : 0;    // accumulator slots of the output-row parity
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int wrow = j < 3 ? j : 3;              // weight row within the kernel row
                        const int bk = j < 3 ? j : (j == 3 ? 1 : 3);  // which transformed input
                        float a[TM];
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt) a[mt] = Ac[((ky * 4 + wrow) * CC + 2 * q) * BM + mt * 32 + aoff];
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                            for (int n = 0; n < TN; ++n)
                                acc[mt][n * NPH + base + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    a[mt], tb[r][n][bk], acc[mt][n * NPH + base + j], 0, 0, 0);
                    }
                }
            }
            return;
        }
        if (W43) {
            // B^T d for F(4,3) (interpolation points 0, +-1, +-2, inf) on six consecutive patch floats:
            //   t0 = 4 d0 - 5 d2 + d4          t1 = (d4 - 4 d2) + (d3 - 4 d1)     t2 = (d4 - 4 d2) - (d3 - 4 d1)
            //   t5 = 4 d1 - 5 d3 + d5          t3 = (d4 - d2) + 2 (d3 - d1)       t4 = (d4 - d2) - 2 (d3 - d1)
            // Explicit software pipeline over the (ky, channel pair) groups of a chunk.  A naive schedule starts every group
            // with "B reads -> wait -> 13 VALU -> first MFMA" and every frequency with "A read -> wait -> MFMA": two LDS round
            // trips and ~80 VALU cycles with at most two MFMAs (128 cycles) in flight (profiles/r01_isa_modconv.md).  Here the
            // raw window of group g+1 is read under the first MFMAs of group g, transformed under its middle ones, and the
            // weight row of the next frequency is read one step ahead; sched_barrier(0) pins that order.  Measured against the
            // naive schedule (profiles/r02_ab_variants.md): -5..8 % per launch on the 64..512-channel layers.
            constexpr int NG = 3 * (CC / 2);
            float bvp[2][TN][6];
            f32x2 raw[TN][3];
            auto read_raw = [&](int gq) {
                const int ky = gq / (CC / 2), q = gq % (CC / 2);
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const float* dp = Pc + 2 * q * g.PSTRIDE + boff[n] + ky * g.PWS;
                    raw[n][0] = *reinterpret_cast<const f32x2*>(dp);
                    raw[n][1] = *reinterpret_cast<const f32x2*>(dp + 2);
                    raw[n][2] = *reinterpret_cast<const f32x2*>(dp + 4);
                }
            };
            auto transform = [&](int gq, float (&bv)[TN][6]) {
                const int q = gq % (CC / 2);
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const f32x2 d01 = raw[n][0], d23 = raw[n][1], d45 = raw[n][2];
                    float a_ = fmaf(-4.f, d23.x, d45.x);
                    asm volatile("" : "+v"(a_));
                    const float b_ = fmaf(-4.f, d01.y, d23.y);
                    const float c_ = d45.x - d23.x, e_ = d23.y - d01.y;
                    bv[n][0] = fmaf(4.f, d01.x, fmaf(-5.f, d23.x, d45.x));
                    bv[n][1] = a_ + b_;
                    bv[n][2] = a_ - b_;
                    bv[n][3] = fmaf(2.f, e_, c_);
                    bv[n][4] = fmaf(-2.f, e_, c_);
                    bv[n][5] = fmaf(4.f, d01.y, fmaf(-5.f, d23.y, d45.y));
                    if (Sc) {
                        const float sc = Sc[2 * q + hi];
#pragma unroll
                        for (int k = 0; k < 6; ++k) bv[n][k] *= sc;
                    }
                }
            };
            auto read_a = [&](int step, float (&a)[TM]) {  // step = gq * 6 + xi
                const int gq = step / 6, xi = step % 6;
                const int ky = gq / (CC / 2), q = gq % (CC / 2);
                if (TM == 2) {
                    const f32x2 a2 = *reinterpret_cast<const f32x2*>(Ac + ((ky * 6 + xi) * CC + 2 * q) * BM + aoff2);
                    a[0] = a2.x, a[TM - 1] = a2.y;
                } else {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) a[mt] = Ac[((ky * 6 + xi) * CC + 2 * q) * BM + mt * 32 + aoff];
                }
            };
            float a_cur[TM], a_nxt[TM];
            read_raw(0);
            read_a(0, a_cur);
            transform(0, bvp[0]);
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                const int ky = gq / (CC / 2);
                (void)ky;
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) {
                    const int step = gq * 6 + xi;
                    if (step + 1 < NG * 6) read_a(step + 1, a_nxt);
                    if (xi == 0 && gq + 1 < NG) read_raw(gq + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                        for (int n = 0; n < TN; ++n)
                            acc[mt][n * NPH + xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt], bvp[gq & 1][n][xi],
                                                                                         acc[mt][n * NPH + xi], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (xi == 2 && gq + 1 < NG) {
                        transform(gq + 1, bvp[(gq + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) a_cur[mt] = a_nxt[mt];
                }
            }
            return;
        }
        if (W23) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int q = 0; q < CC / 2; ++q) {
                    float bv[TN][4];
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        const float* dp = Pc + 2 * q * g.PSTRIDE + boff[n] + ky * g.PWS;  // even float offset: 8-byte aligned
                        const f32x2 d01 = *reinterpret_cast<const f32x2*>(dp);
                        const f32x2 d23 = *reinterpret_cast<const f32x2*>(dp + 2);
                        bv[n][0] = d01.x - d23.x;
                        bv[n][1] = d01.y + d23.x;
                        bv[n][2] = d23.x - d01.y;
                        bv[n][3] = d01.y - d23.y;
                        if (Sc) {
                            const float sc = Sc[2 * q + hi];
#pragma unroll
                            for (int k = 0; k < 4; ++k) bv[n][k] *= sc;
                        }
                    }
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        float a[TM];
                        if (TM == 2) {  // weight rows are packed [lane][mt] per 64 output channels: one 8-byte read, immediate offset
                            const f32x2 a2 = *reinterpret_cast<const f32x2*>(Ac + ((ky * 4 + xi) * CC + 2 * q) * BM + aoff2);
                            a[0] = a2.x, a[TM - 1] = a2.y;
                        } else {
#pragma unroll
                            for (int mt = 0; mt < TM; ++mt) a[mt] = Ac[((ky * 4 + xi) * CC + 2 * q) * BM + mt * 32 + aoff];
                        }
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                            for (int n = 0; n < TN; ++n)
                                acc[mt][n * NPH + xi] =
                                    __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bv[n][xi], acc[mt][n * NPH + xi], 0, 0, 0);
                    }
                }
            }
            return;
        }
        {
            // direct / polyphase branch: one-step look-ahead of both operands over the 9 x CC/2 (tap, channel pair) steps; the
            // style multiply of the next B operand runs behind the current MFMAs (-5..7 % on the 512-channel layers).
            constexpr int NS = 9 * (CC / 2);
            float a_c[TM], b_c[TN], a_n[TM], b_n[TN];
            auto fetch = [&](int st, float (&a)[TM], float (&b)[TN]) {
                const int tap = st / (CC / 2), q = st % (CC / 2);
                const int ky = tap / 3, kx = tap % 3;
                const int dy = UP ? (ky == 2 ? 0 : 1) : ky;
                const int dx = UP ? (kx == 2 ? 0 : 1) : kx;
                const int tapoff = dy * g.PWS + dx;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) a[mt] = Ac[(tap * CC + 2 * q) * BM + mt * 32 + aoff];
#pragma unroll
                for (int n = 0; n < TN; ++n) b[n] = Pc[2 * q * g.PSTRIDE + boff[n] + tapoff];
            };
            auto scale = [&](int st, float (&b)[TN]) {
                if (Sc) {
                    const float sc = Sc[2 * (st % (CC / 2)) + hi];
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] *= sc;
                }
            };
            fetch(0, a_c, b_c);
            scale(0, b_c);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int tap = st / (CC / 2);
                const int ky = tap / 3, kx = tap % 3;
                const int ph = UP ? ((ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0)) : 0;
                if (st + 1 < NS) fetch(st + 1, a_n, b_n);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[mt][n * NPH + ph] =
                            __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[mt], b_c[n], acc[mt][n * NPH + ph], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < NS) scale(st + 1, b_n);
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) a_c[mt] = a_n[mt];
#pragma unroll
                for (int n = 0; n < TN; ++n) b_c[n] = b_n[n];
            }
            return;
        }
    };

    if (FAST) {
        // Double-buffered pipeline, ONE barrier per chunk.  The weight tile goes HBM/L2 -> LDS by DMA
        // (global_load_lds, 1 KiB per wave instruction, no staging registers, no ds_write); the feature patch goes
        // through registers because it is scaled by the style and masked at the image border on the way in.
        constexpr int RPI = 256 / BM;              // weight rows (BM floats) per 1-KiB DMA instruction
        constexpr int LPR = 64 / RPI;              // lanes per row
        constexpr int A_INSTR = (NTAPS * CC + RPI - 1) / RPI;  // DMA instructions per tile (the last one may be partial)
        constexpr int A_PER_WAVE = (A_INSTR + 3) / 4;
        int a_goff[A_PER_WAVE];
#pragma unroll
        for (int k = 0; k < A_PER_WAVE; ++k) {
            const int row = (wave + 4 * k) * RPI + lane / LPR;
            const int col = (lane % LPR) * 4;
            const int tap = row / CC, c = row - tap * CC;
            a_goff[k] = row < NTAPS * CC ? (tap * g.Cin + c) * g.CoutPad + col : -1;  // -1: lane past the tile, masked off
        }
        // Both DMAs are MUBUF `buffer_load ... lds` through raw buffer descriptors: while a FLAT-encoded global_load_lds is
        // in flight the compiler's wait insertion treats the LGKM counter as out of order and turns EVERY LDS wait of the
        // chunk into lgkmcnt(0) (profiles/r01_isa_modconv.md); with the buffer form it emits partial counts, which is what
        // the look-ahead LDS reads of the pipelined MFMA phases need.
#ifdef MAUA_DEVICE_PASS
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, 0x7fffffff, 0x00020000);
#endif
        auto issue_dma = [&](int chunk, int buf) {
            float* dst = As + buf * A_FLOATS;
            (void)dst;
#pragma unroll
            for (int k = 0; k < A_PER_WAVE; ++k) {
                const int i = wave + 4 * k;  // scalar
                constexpr bool RAGGED = (NTAPS * CC) % RPI != 0;  // only then can lanes of the last instruction be masked
                if (i < A_INSTR && (!RAGGED || a_goff[k] >= 0))
#ifdef MAUA_DEVICE_PASS
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        w_rsrc, (__attribute__((address_space(3))) void*)(dst + i * 256), 16, a_goff[k] * 4,
                        (int)(((size_t)chunk * CC * g.CoutPad + m0) * 4), 0, 0);
#else
                    (void)a_goff[k];
#endif
            }
        };
        auto load_patch = [&](int chunk) {
            const int c0 = chunk * CC;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float* __restrict__ xbase = p.x + (size_t)(c0 + c) * plane_in;  // uniform
#pragma unroll
                for (int i = 0; i < MAX_POS; ++i)
                    if (i == 0 || tid + i * 256 < g.PSTRIDE) pv[i][c] = xbase[(unsigned)src_off[i]];
            }
            if (MULTI) {
#pragma unroll
                for (int i = 0; i < MAX_POS; ++i)
#pragma unroll
                    for (int c = 0; c < CC; ++c) pv[i][c] *= p.s[sb_off[i] + c0 + c];
            }
        };
        auto write_patch = [&](int chunk, int buf) {
            const int c0 = chunk * CC;
            float* Pd = Ps + buf * (CC * g.PSTRIDE);
            float sc[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) sc[c] = (one_image && b0 < g.B) ? p.s[b0 * g.s_stride + c0 + c] : 1.f;
#pragma unroll
            for (int i = 0; i < MAX_POS; ++i) {
                const int pp = tid + i * 256;
                if (pp < g.PSTRIDE) {
#pragma unroll
                    for (int c = 0; c < CC; ++c) Pd[c * g.PSTRIDE + pp] = pv[i][c] * (sc[c] * src_mask[i]);
                }
            }
        };
        int cur = 0;
        // measured (30-launch averages): -6..7 % on the transposed 64-row configs, -4 % on the transposed 32-row config,
        // -3 % on the 64/128-row Winograd configs, +1 % on the 32-channel 1024^2 Winograd layer (its features stream from
        // HBM; the register-staged patch tolerates that latency better), which keeps the register path
        constexpr bool PDMA = !MULTI && !(W23 && BM == 32);
        if (PDMA) {
            // Feature patch by DMA as well: raw features go L2/HBM -> LDS without staging registers or ds_write, double
            // buffered like the weight tile.  Out-of-image patch elements are never written: their lanes are masked out of
            // the DMA and their LDS slots are zeroed once by the owning thread.  The style scale moves to the B-operand read (mfma_chunk Sc).
            // (A three-deep patch ring with a two-chunk prefetch distance measured the same and was dropped.)
            const int PBUF = CC * g.PSTRIDE;
            float* Ss = Ps + 2 * PBUF;
            bool pvalid[MAX_POS];
#pragma unroll
            for (int i = 0; i < MAX_POS; ++i) {
                pvalid[i] = src_mask[i] != 0.f;
                const int pp = tid + i * 256;
                if (!pvalid[i] && pp < g.PSTRIDE) {  // border / padding element of this thread: zero in both buffers, for good
#pragma unroll
                    for (int c = 0; c < CC; ++c) Ps[c * g.PSTRIDE + pp] = 0.f, Ps[PBUF + c * g.PSTRIDE + pp] = 0.f;
                }
            }
            const int nK = (chunk_end - chunk_begin) * CC;
            for (int e = tid; e < nK; e += 256) Ss[e] = p.s[b0 * g.s_stride + chunk_begin * CC + e];
            // lane offsets relative to the image (bytes, < 4 MiB) on a scalar running channel pointer: the DMA address is
            // SGPR base + 32-bit VGPR offset, two scalar adds per channel
            unsigned rel_bytes[MAX_POS];
#pragma unroll
            for (int i = 0; i < MAX_POS; ++i)
                rel_bytes[i] = pvalid[i] ? (unsigned)(src_off[i] - b0 * g.Cin * (int)plane_in) * 4u : 0u;
            const char* ximg = reinterpret_cast<const char*>(p.x + (size_t)b0 * g.Cin * plane_in);
            const size_t plane_bytes = plane_in * sizeof(float);
            (void)ximg, (void)plane_bytes;  // (only the device pass builds the descriptor)
#ifdef MAUA_DEVICE_PASS
            // one image's features (Cin planes, < 2 GiB: checked on the host) behind a raw buffer descriptor
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(ximg), 0, 0x7fffffff, 0x00020000);
#endif
            auto issue_patch = [&](int chunk, int buf) {
                float* dst = Ps + buf * PBUF + wave * 64;
                (void)dst;
#pragma unroll
                for (int c = 0; c < CC; ++c) {
#pragma unroll
                    for (int i = 0; i < MAX_POS; ++i)
                        if (pvalid[i]) {
#ifdef MAUA_DEVICE_PASS
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                x_rsrc, (__attribute__((address_space(3))) void*)(dst + c * g.PSTRIDE + i * 256), 4,
                                (int)rel_bytes[i], (int)((size_t)(chunk * CC + c) * plane_bytes), 0, 0);
#endif
                        }
                }
            };
            if (chunk_begin < chunk_end) {
                issue_dma(chunk_begin, 0);
                issue_patch(chunk_begin, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
                if (chunk + 1 < chunk_end) {
                    if (!MAUA_DBG(32)) issue_dma(chunk + 1, cur ^ 1);
                    if (!MAUA_DBG(64)) issue_patch(chunk + 1, cur ^ 1);
                }
                if (!MAUA_DBG(2)) mfma_chunk(As + cur * A_FLOATS, Ps + cur * PBUF, Ss + (chunk - chunk_begin) * CC);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur ^= 1;
            }
        } else {
        if (chunk_begin < chunk_end) {
            issue_dma(chunk_begin, 0);
            load_patch(chunk_begin);
            write_patch(chunk_begin, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
            const bool more = chunk + 1 < chunk_end;
            if (more && !MAUA_DBG(4)) {
                if (!MAUA_DBG(32)) issue_dma(chunk + 1, cur ^ 1);
                if (!MAUA_DBG(64)) load_patch(chunk + 1);
            }
            if (!MAUA_DBG(2)) mfma_chunk(As + cur * A_FLOATS, Ps + cur * (CC * g.PSTRIDE), nullptr);
            if (more && !MAUA_DBG(4 | 64)) write_patch(chunk + 1, cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
        }
    } else {
    if (chunk_begin < chunk_end) issue_loads(chunk_begin);
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        write_lds(chunk);
        __syncthreads();
        if (chunk + 1 < chunk_end) issue_loads(chunk + 1);

        mfma_chunk(As, Ps, nullptr);
        __syncthreads();
    }
    }

    // ---- epilogue
    // Per-channel epilogue operands (wscale * demod, bias) go through LDS: fetched once per workgroup with all loads in
    // flight together.  (Loading them per accumulator element from global memory serialises ~64 dependent L2 round
    // trips behind the stores — that was ~45 % of the lifetime of a 32-channel 1024^2 workgroup.)
    const bool to_ws = g.splits > 1 || g.force_ws;
    float* outp = to_ws ? (p.ws + (size_t)split * g.ws_slab) : p.y;
    const float* noise_base = p.noise;
    int64_t noise_bstride = g.noise_batch_stride;
    if (p.src) {  // uniform scalar loads: base of this launch's first frame inside the HBM-resident sequence
        noise_bstride = p.src->noise_stride[p.noise_slot];
        noise_base = p.src->noise[p.noise_slot];
        if (noise_base) noise_base += (int64_t)p.src->frame0 * noise_bstride;
    }
    const float nw = (!to_ws && g.fuse_act && noise_base) ? p.noise_w[0] : 0.f;  // (scaled by act_gain where it is applied)
    const size_t plane_out = (size_t)g.OH * g.OW;
    float* Eg = lds;        // [BM] gain
    float* Eb = lds + BM;   // [BM] bias
    // leaky ReLU is positively homogeneous: its sqrt(2) gain is folded into gain, bias and noise once, and the activation
    // itself is max(t, 0.2 t) — 4 instead of 7 VALU operations per output value
    const float act_gain = (!to_ws && g.fuse_act) ? 1.41421356237309515f : 1.f;
    if (!MULTI) {
        for (int i = tid; i < BM; i += 256) {
            const int o = m0 + i;
            float gain = g.wscale, bias = 0.f;
            if (!to_ws && o < g.Cout && b0 < g.B) {
                if (p.d) gain *= p.d[b0 * g.Cout + o];
                if (g.fuse_act && p.bias) bias = p.bias[o];
            }
            Eg[i] = gain * act_gain;
            Eb[i] = bias * act_gain;
            if (!UP && WM == 1 && g.rgb) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    lds[(2 + c) * BM + i] = (o < g.Cout && b0 < g.B)
                                                ? g.rgb_wscale * p.rgb_w[c * g.Cout + o] * p.rgb_s[b0 * g.s_stride + o]
                                                : 0.f;
            }
        }
        __syncthreads();
    }
    // every noise value this lane needs, fetched before the first store (loads cannot be hoisted above stores later)
    float nz_all[TN][NPH];
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int sub = wn * TN + n;
        const int jx = l31 & (SW - 1), jy = l31 >> g.lsw;
        const int sx = sub & ((1 << g.lnsx) - 1);
        const int sy = (sub >> g.lnsx) & ((1 << g.lnsy) - 1);
        const int b = b0 + (sub >> (g.lnsx + g.lnsy));
        int gy = ty0 + sy * SH + jy, gx = tx0 + sx * SW + jx;
        if (UP && g.flat) gy = gx / g.GW, gx -= gy * g.GW;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            const int oy = UP ? 2 * gy + ((ph >> 1) & 1) : gy, ox = UP ? 2 * gx + (ph & 1) : (WINO ? WX * gx + (ph < WX ? ph : 0) : gx);
            nz_all[n][ph] = 0.f;
            if (WINO && ph >= WX) continue;
            if (nw != 0.f && b < g.B && gy < g.GH && gx < g.GW && oy < g.OH && ox < g.OW)
                nz_all[n][ph] = nw * act_gain * (MULTI ? noise_base[(size_t)b * noise_bstride + (size_t)oy * g.OW + ox]
                                                       : (noise_base + (size_t)b0 * noise_bstride)[(unsigned)(oy * g.OW + ox)]);
        }
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int sub = wn * TN + n;
        const int jx = l31 & (SW - 1), jy = l31 >> g.lsw;
        const int sx = sub & ((1 << g.lnsx) - 1);
        const int sy = (sub >> g.lnsx) & ((1 << g.lnsy) - 1);
        const int img = sub >> (g.lnsx + g.lnsy);
        const int b = b0 + img;
        int gy = ty0 + sy * SH + jy, gx = tx0 + sx * SW + jx;
        if (UP && g.flat) gy = gx / g.GW, gx -= gy * g.GW;
        const bool pos_ok = (b < g.B) && (gy < g.GH) && (gx < g.GW);
        // transposed conv: the two x-parities of a position are adjacent in memory -> one 8-byte store per lane
        // (rows of the (2W+1)-wide plane are only 4-byte aligned: f32x2u is an align-4 vector type)
        constexpr int PXN = UW ? 4 : (UP ? 2 : WX);  // outputs per position along x (pairs / quads in the Winograd modes)
#pragma unroll
        for (int py = 0; py < (UP ? 2 : 1); ++py) {
            const int oy = UP ? 2 * gy + py : gy;
            const int ox = UW ? 4 * gx : (UP ? 2 * gx : WX * gx);
            const bool ok0 = pos_ok && oy < g.OH && ox < g.OW;
            const bool ok1 = (UP || WINO) && ok0 && (ox + PXN - 1 < g.OW);  // the whole pair / quad is inside the row
            float nzv[PXN];
#pragma unroll
            for (int px = 0; px < PXN; ++px) nzv[px] = UW ? 0.f : nz_all[n][UP ? py * 2 + px : (WINO ? px : 0)];
            // row pointer of this wave's first output channel (block-uniform unless several images share a tile) + a
            // 32-bit lane offset: the per-element address is scalar base + VGPR offset, no 64-bit vector arithmetic
            float* blk = outp + ((size_t)(MULTI ? b : b0) * g.Cout + m0 + wm * (TM * 32)) * plane_out;
            const unsigned lane_off = (unsigned)oy * (unsigned)g.OW + (unsigned)ox + (unsigned)(4 * hi) * (unsigned)plane_out;
            float rgbp[PXN][3];
#pragma unroll
            for (int px = 0; px < PXN; ++px) rgbp[px][0] = rgbp[px][1] = rgbp[px][2] = 0.f;
            // Everything that does not depend on the accumulator element is decided once per (n, py): whether the tail is
            // applied, whether the feature map is stored, and the lane's store class.  Tiles that lie completely inside the
            // image and the channel range (every tile of the power-of-two plain layers) take the INTERIOR instance of the
            // element loop, which has no per-lane predication at all — on the 32/64-channel 1024^2/512^2 layers (4-8 K-chunks
            // per workgroup) the epilogue's instruction count is a first-order cost.
            const bool apply_act = !to_ws && g.fuse_act;
            const bool do_rgb = !UP && WM == 1 && !MULTI && g.rgb;
            const bool store_feat = !(do_rgb && g.rgb == 2) && !MAUA_DBG(1);
            const int n_ok = g.Cout - m0 - wm * (TM * 32) - 4 * hi;  // this lane's channel rows oe < n_ok exist
            const unsigned lane_bytes = lane_off * 4u;
            auto elements = [&](auto interior_tag) {
                constexpr bool IN = decltype(interior_tag)::value;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int oe = mt * 32 + (e & 3) + 8 * (e >> 2);
                        const int ol = wm * (TM * 32) + oe + 4 * hi;
                        const int o = m0 + ol;
                        float gain, bias;
                        if (MULTI) {
                            gain = g.wscale, bias = 0.f;
                            if (!to_ws && ok0 && o < g.Cout) {
                                if (p.d) gain *= p.d[b * g.Cout + o];
                                if (g.fuse_act && p.bias) bias = p.bias[o];
                            }
                            gain *= act_gain, bias *= act_gain;
                        } else {
                            gain = Eg[ol], bias = Eb[ol];
                        }
                        float v[PXN];
#pragma unroll
                        for (int px = 0; px < PXN; ++px) {
                            float raw;
                            if (UW) {  // columns 4 gx .. 4 gx + 3 = e_p, o_p, e_p+1, o_p+1
                                const int sb = n * NPH + py * 5;
                                raw = px == 0 ? acc[mt][sb + 0][e] + acc[mt][sb + 1][e]
                                    : px == 1 ? acc[mt][sb + 3][e]
                                    : px == 2 ? acc[mt][sb + 1][e] + acc[mt][sb + 2][e] : acc[mt][sb + 4][e];
                            } else if (W43) {
                                // A^T m: y0 = m0+m1+m2+m3+m4, y1 = (m1-m2) + 2(m3-m4), y2 = (m1+m2) + 4(m3+m4),
                                //        y3 = (m1-m2) + 8(m3-m4) + m5
                                const float m0_ = acc[mt][n * NPH + 0][e], m1_ = acc[mt][n * NPH + 1][e];
                                const float m2_ = acc[mt][n * NPH + 2][e], m3_ = acc[mt][n * NPH + 3][e];
                                const float m4_ = acc[mt][n * NPH + 4][e], m5_ = acc[mt][n * NPH + 5][e];
                                const float s12 = m1_ + m2_, d12 = m1_ - m2_, s34 = m3_ + m4_, d34 = m3_ - m4_;
                                raw = px == 0 ? (m0_ + s12) + s34 : px == 1 ? fmaf(2.f, d34, d12)
                                    : px == 2 ? fmaf(4.f, s34, s12) : fmaf(8.f, d34, d12) + m5_;
                            } else if (W23) {  // inverse transform: y0 = m0 + m1 + m2, y1 = m1 - m2 - m3
                                const float m0_ = acc[mt][n * NPH + 0][e], m1_ = acc[mt][n * NPH + 1][e];
                                const float m2_ = acc[mt][n * NPH + 2][e], m3_ = acc[mt][n * NPH + 3][e];
                                raw = px == 0 ? (m0_ + m1_) + m2_ : (m1_ - m2_) - m3_;
                            } else {
                                raw = acc[mt][n * NPH + (UP ? py * 2 + px : 0)][e];
                            }
                            const float t = fmaf(raw, gain, nzv[px] + bias);  // bias and noise are 0 unless the tail is fused
                            v[px] = apply_act ? fmaxf(t, 0.2f * t) : t;
                        }
                        if (do_rgb) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const float rw = lds[(2 + c) * BM + ol];
#pragma unroll
                                for (int px = 0; px < PXN; ++px) rgbp[px][c] = fmaf(rw, v[px], rgbp[px][c]);
                            }
                        }
                        if (!store_feat) continue;
                        // scalar row base + 32-bit lane byte offset
                        char* rowb = reinterpret_cast<char*>(blk + (size_t)oe * plane_out);
                        float* dst = reinterpret_cast<float*>(rowb + lane_bytes);
                        auto store_all = [&]() {
                            if (PXN == 4) *reinterpret_cast<f32x4u*>(dst) = f32x4{v[0], v[1 % PXN], v[2 % PXN], v[3 % PXN]};
                            else *reinterpret_cast<f32x2u*>(dst) = f32x2{v[0], v[PXN - 1]};
                        };
                        if (IN) {
                            if (UP || WINO) store_all();
                            else dst[0] = v[0];
                        } else if (oe < n_ok) {
                            if ((UP || WINO) && ok1) store_all();
                            else if (ok0) dst[0] = v[0];
                        }
                    }
                }
            };
            const bool interior = !UP && !MULTI && (m0 + BM <= g.Cout) && (ty0 + THt <= g.GH) && (tx0 + TWd <= g.GW) &&
                                  (!WINO || g.OW % WX == 0);
            // (the 64-row F(4,3) config is at the register limit: with two instances of the element loop the compiler
            // spills 42 accumulator registers, with one it spills none and runs 4 % faster)
            if (interior && !(W43 && TM == 2 && WM == 1)) elements(std::true_type{});
            else elements(std::false_type{});
            if (!UP && WM == 1 && !MULTI && g.rgb) {
                // all channels of a pixel live in one wave: lanes l and l+32 hold the two halves of the channel set
#pragma unroll
                for (int px = 0; px < PXN; ++px)
#pragma unroll
                    for (int c = 0; c < 3; ++c) rgbp[px][c] += __shfl_xor(rgbp[px][c], 32, 64);
                if (hi == 0 && ok0) {
                    // upfirdn2d(skip, k4, up=2, pad=(2,1)) at (oy, x): exactly two source rows / columns are live,
                    // iy0 = floor((oy-1)/2), iy0+1 (taps k4[1]/k4[3] for even oy, k4[0]/k4[2] for odd) — all loads
                    // (3 channels x 2 x 2 per pixel) are unconditional (clamped) and in flight together, masked by weight 0.
                    const int sh = g.H >> 1, sw = g.W >> 1;
                    const int iy0 = (oy - 1) >> 1;
                    const int ty_ = (oy & 1) ? 2 : 3;  // tap index of the FIRST live row
                    float wy[2];
                    int ry[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int yy = iy0 + q;
                        ry[q] = min(max(yy, 0), sh - 1);
                        wy[q] = (yy >= 0 && yy < sh) ? 1.f : 0.f;
                    }
                    float wgt[PXN][2][2];
                    float sv[PXN][3][2][2];
                    // (this branch is only compiled for single-image tiles: b == b0, bases are scalar, offsets 32-bit)
                    const float* __restrict__ skip_img = p.rgb_skip ? p.rgb_skip + (size_t)b0 * 3 * sh * sw : nullptr;
#pragma unroll
                    for (int px = 0; px < PXN; ++px) {
                        const int x = ox + px;
                        const int ix0 = (x - 1) >> 1;
                        const int tx_ = (x & 1) ? 2 : 3;
                        float wx[2];
                        int rx[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int xx = ix0 + q;
                            rx[q] = min(max(xx, 0), sw - 1);
                            wx[q] = (xx >= 0 && xx < sw) ? 1.f : 0.f;
                        }
#pragma unroll
                        for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                            for (int qx = 0; qx < 2; ++qx) {
                                wgt[px][qy][qx] = 0.f;
#pragma unroll
                                for (int c = 0; c < 3; ++c) sv[px][c][qy][qx] = 0.f;
                            }
                        if (skip_img) {
#pragma unroll
                            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                                for (int qx = 0; qx < 2; ++qx) {
                                    wgt[px][qy][qx] = p.rgb_k4[(ty_ - 2 * qy) * 4 + (tx_ - 2 * qx)] * wy[qy] * wx[qx];
#pragma unroll
                                    for (int c = 0; c < 3; ++c)
                                        sv[px][c][qy][qx] = skip_img[(unsigned)((c * sh + ry[qy]) * sw + rx[qx])];
                                }
                        }
                    }
                    float* __restrict__ rgb_img = p.rgb_out + (size_t)b0 * 3 * plane_out;
                    const unsigned rgb_off = (unsigned)(oy * g.OW + ox);
                    uint32_t pix[PXN];  // packed frame bytes (R | G << 8 | B << 16) when the frame epilogue is fused
#pragma unroll
                    for (int px = 0; px < PXN; ++px) pix[px] = 0u;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float val[PXN];
#pragma unroll
                        for (int px = 0; px < PXN; ++px) {
                            val[px] = rgbp[px][c] + p.rgb_bias[c];
#pragma unroll
                            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                                for (int qx = 0; qx < 2; ++qx) val[px] = fmaf(wgt[px][qy][qx], sv[px][c][qy][qx], val[px]);
                        }
                        if (p.rgb_u8) {  // render.py:40-43: clamp(-1, 1), (x + 1) * 127.5, truncating cast
#pragma unroll
                            for (int px = 0; px < PXN; ++px)
                                pix[px] |= (uint32_t)((fminf(fmaxf(val[px], -1.f), 1.f) + 1.f) * 127.5f) << (8 * c);
                            if (!p.rgb_out) continue;  // (both given: the fp32 planes are written as well — the parity tests' tap)
                        }
                        float* ro = rgb_img + (size_t)c * plane_out + rgb_off;
                        if (PXN == 4 && ok1) *reinterpret_cast<f32x4u*>(ro) = f32x4{val[0], val[1 % PXN], val[2 % PXN], val[3 % PXN]};
                        else if (PXN == 2 && ok1) *reinterpret_cast<f32x2u*>(ro) = f32x2{val[0], val[PXN - 1]};
                        else ro[0] = val[0];
                    }
                    if (p.rgb_u8) {
                        uint8_t* fo = p.rgb_u8 + ((size_t)b0 * plane_out + rgb_off) * 3;
                        if (PXN == 4 && ok1) {  // 12 bytes at a 12-byte multiple: three dword stores
                            uint32_t* fw = reinterpret_cast<uint32_t*>(fo);
                            fw[0] = pix[0] | (pix[1 % PXN] << 24);
                            fw[1] = (pix[1 % PXN] >> 8) | (pix[2 % PXN] << 16);
                            fw[2] = (pix[2 % PXN] >> 16) | (pix[3 % PXN] << 8);
                        } else {
#pragma unroll
                            for (int px = 0; px < PXN; ++px)
                                if (px == 0 || ok1) fo[3 * px] = (uint8_t)pix[px], fo[3 * px + 1] = (uint8_t)(pix[px] >> 8), fo[3 * px + 2] = (uint8_t)(pix[px] >> 16);
                        }
                    }
                }
            }
        }
    }
}

// The tail behind a slab sum, with every rounding spelled out (no fp contraction): reduce_tail_kernel and reduce_tail_rgbpart_kernel must
// store the same bits for the same slabs, whatever the compiler would fuse in either context.
__device__ __forceinline__ float slab_tail(float v, float dv, float nw, float nzv, float bv, bool act) {
#pragma clang fp contract(off)  // (HIP's __fmul_rn / __fadd_rn are plain operators: they do not stop the contraction)
    v = v * dv;
    if (!act) return v;
    const float nz = nw * nzv;
    const float t = v + nz;
    return lrelu_gain(t + bv);
}

// Sum split-K slabs and apply the same tail as the fused epilogue.
__global__ __launch_bounds__(256) void reduce_tail_kernel(const float* __restrict__ ws, int splits, int64_t slab,
                                                          float* __restrict__ y, const float* __restrict__ d,
                                                          const float* __restrict__ noise, int64_t noise_batch_stride,
                                                          const float* __restrict__ noise_w,
                                                          const float* __restrict__ bias, int fuse_act, int cout,
                                                          int64_t plane, int64_t total,
                                                          const maua_frame_source_t* __restrict__ src, int noise_slot) {
    if (src) {
        noise_batch_stride = src->noise_stride[noise_slot];
        noise = src->noise[noise_slot];
        if (noise) noise += (int64_t)src->frame0 * noise_batch_stride;
    }
    const float nw = (fuse_act && noise) ? noise_w[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // every operand of this element is fetched before the first use: the slab values (up to 32, all in flight: with four at a time a
        // 32-way split was eight dependent round trips, 8 us per launch on maps of a few KB), the demodulation factor, the noise value, the bias
        const int64_t bc = i / plane;
        const int64_t pix = i - bc * plane;
        const int b = (int)(bc / cout), o = (int)(bc - (int64_t)b * cout);
        const float dv = d ? d[bc] : 1.f;
        const float nzv = nw != 0.f ? noise[(size_t)b * noise_batch_stride + pix] : 0.f;
        const float bv = (fuse_act && bias) ? bias[o] : 0.f;
        float v = 0.f;
        int s = 0;
        for (; s + 32 <= splits; s += 32) {
            float a[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k] = ws[(size_t)(s + k) * slab + i];
#pragma unroll
            for (int k = 0; k < 32; k += 4) v += (a[k] + a[k + 1]) + (a[k + 2] + a[k + 3]);  // (the association of the 4-wide form)
        }
        if (s + 16 <= splits) {
            float a[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = ws[(size_t)(s + k) * slab + i];
#pragma unroll
            for (int k = 0; k < 16; k += 4) v += (a[k] + a[k + 1]) + (a[k + 2] + a[k + 3]);
            s += 16;
        }
        for (; s + 4 <= splits; s += 4) {
            const float a0 = ws[(size_t)s * slab + i], a1 = ws[(size_t)(s + 1) * slab + i];
            const float a2 = ws[(size_t)(s + 2) * slab + i], a3 = ws[(size_t)(s + 3) * slab + i];
            v += (a0 + a1) + (a2 + a3);
        }
        for (; s < splits; ++s) v += ws[(size_t)s * slab + i];
        y[i] = slab_tail(v, dv, nw, nzv, bv, fuse_act != 0);
    }
}

// ---- low-resolution layers (4^2 .. 32^2 outputs): the split-K reduction fused with what follows it -----------------------------------------
// On these maps every launch is a fixed cost of 5 .. 20 us whatever it computes (tools/rocpd_timeline.py): the transposed layers ran
// convolution -> reduce_tail -> blur + noise + act (three launches, 22 .. 33 us for the last two), the plain ones convolution -> reduce_tail ->
// ToRGB (16 .. 29 us).  The two kernels below take the workspace slabs of the convolution directly.
//
// Up-sampling layer: one workgroup per (image, channel) plane.  The raw (2H+1) x (2W+1) plane — the sum of the split-K slabs times the
// demodulation factor, exactly reduce_tail_kernel's value — is formed in LDS, then the 4 x 4 blur (pad (1, 1)), noise, bias, leaky ReLU and the
// style fold's scale are applied in the arithmetic order of fir_tile_kernel: the result is bit-identical to the three-launch path.
__global__ __launch_bounds__(256) void reduce_blur_tail_kernel(const float* __restrict__ ws, int splits, int64_t slab, float* __restrict__ y,
                                                               const float* __restrict__ d, const float* __restrict__ k4,
                                                               const float* __restrict__ noise, int64_t noise_batch_stride,
                                                               const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                               const float* __restrict__ post_s, int post_stride, int cout, int h, int w,
                                                               const maua_frame_source_t* __restrict__ src, int noise_slot, int edge_slab0) {
    // edge_slab0 (the F(2,2)^2 form, modconv_up2d.hip): raw row 2h and column 2w are written by the edge kernel, into slab 0 only
    extern __shared__ __attribute__((aligned(16))) float raw[];  // [(2h + 3)][(2w + 3)]: the raw plane inside a border of zeros (the blur's padding)
    const int RH = 2 * h + 1, RW = 2 * w + 1, OH = 2 * h, OW = 2 * w, PW = RW + 2;
    const int bc = blockIdx.x, b = bc / cout, c = bc - b * cout;
    const int tid = threadIdx.x;
    if (src) {
        noise_batch_stride = src->noise_stride[noise_slot];
        noise = src->noise[noise_slot];
        if (noise) noise += (int64_t)src->frame0 * noise_batch_stride;
    }
    float kf[4][4];  // flipped taps (uniform loads)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[i][j] = k4[(3 - i) * 4 + (3 - j)];
    const float dv = d ? d[bc] : 1.f;
    const float g = 1.41421356237309515f;
    const float nw = noise ? noise_w[0] * 1.41421356237309515f : 0.f;
    const float bs = bias ? bias[c] * 1.41421356237309515f : 0.f;
    const float post = post_s ? post_s[(size_t)b * post_stride + c] : 1.f;
    const float* wp = ws + (size_t)bc * RH * RW;
    const int n_raw = RH * RW, n_out = OH * OW;
    // the lane's noise values (at most four outputs per thread: 2h x 2w <= 1024, checked by the launcher) travel with the slab loads
    float nzv[4] = {0.f, 0.f, 0.f, 0.f};
    if (noise)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256 * k < n_out) nzv[k] = noise[(size_t)b * noise_batch_stride + tid + 256 * k];
    {
        // a thread owns raw elements tid, tid + 256, ... (at most five: (2h + 1)(2w + 1) <= 33 x 33): the slab loads of ALL of them are issued
        // before the first add, eight slabs at a time (element by element the plane took five dependent round trips)
        constexpr int NE = 5;
        float v[NE];
        int nsl[NE];  // slabs that hold this element
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            v[q] = 0.f;
            const int e = tid + 256 * q, r = e / RW;
            nsl[q] = e >= n_raw ? 0 : (edge_slab0 && (r == RH - 1 || e - r * RW == RW - 1)) ? 1 : splits;
        }
        int sp = 0;
        for (; sp + 8 <= splits; sp += 8) {
            float a[NE][8];
#pragma unroll
            for (int q = 0; q < NE; ++q)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[q][k] = (sp + k < nsl[q]) ? wp[(size_t)(sp + k) * slab + tid + 256 * q] : 0.f;
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                v[q] += (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);  // (reduce_tail_kernel's association)
                v[q] += (a[q][4] + a[q][5]) + (a[q][6] + a[q][7]);
            }
        }
        for (; sp + 4 <= splits; sp += 4) {
            float a[NE][4];
#pragma unroll
            for (int q = 0; q < NE; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) a[q][k] = (sp + k < nsl[q]) ? wp[(size_t)(sp + k) * slab + tid + 256 * q] : 0.f;
#pragma unroll
            for (int q = 0; q < NE; ++q) v[q] += (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);
        }
        for (; sp < splits; ++sp) {
            float a[NE];
#pragma unroll
            for (int q = 0; q < NE; ++q) a[q] = (sp < nsl[q]) ? wp[(size_t)sp * slab + tid + 256 * q] : 0.f;
#pragma unroll
            for (int q = 0; q < NE; ++q) v[q] += a[q];
        }
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = tid + 256 * q;
            if (e < n_raw) {
                const int r = e / RW;
                raw[(r + 1) * PW + (e - r * RW) + 1] = v[q] * dv;
            }
        }
        // the border: rows 0 and RH + 1, columns 0 and RW + 1 of the rows between
        for (int e = tid; e < 2 * PW + 2 * RH; e += 256) {
            const int cell = e < PW ? e : e < 2 * PW ? (RH + 1) * PW + (e - PW) : (1 + ((e - 2 * PW) >> 1)) * PW + ((e & 1) ? RW + 1 : 0);
            raw[cell] = 0.f;
        }
    }
    __syncthreads();
    float* yp = y + (size_t)bc * n_out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = tid + 256 * k;
        if (o >= n_out) break;
        const int Y = o / OW, X = o - Y * OW;
        float acc = 0.f;
        const float* win = raw + Y * PW + X;  // raw[Y - 1 + i][X - 1 + j] of the un-bordered plane
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = (i == 0 && j == 0) ? kf[0][0] * win[0] : fmaf(kf[i][j], win[i * PW + j], acc);
        const float tt = fmaf(acc, g, fmaf(nw, nzv[k], bs));
        yp[o] = fmaxf(tt, 0.2f * tt) * post;
    }
}

// Plain layer followed by a ToRGB: one workgroup per (image, group of 32 output channels, PT pixels; PT = 32, or 16 on 4 x 4 maps).  A thread
// sums the slabs of 32 PT / 256 elements (channels cl, cl + 256 / PT, ... of the group at one pixel: a wave instruction reads whole 128-byte
// rows of PT pixels), applies reduce_tail_kernel's tail (the stored feature map is bit-identical to that kernel's) and multiplies the activated
// values by their channels' three modulated ToRGB weights; the 32 channels of the group meet in LDS and leave as three partial planes
// rgb_part [B][3 * (cout / 32)][H][W] (plane 3 g + colour), which maua_torgb_f32's plane-sum form (5 us) turns into the image — instead of
// a ToRGB pass that re-reads the feature map (10 .. 22 us on these maps).
template <int PT>
__global__ __launch_bounds__(256) void reduce_tail_rgbpart_kernel(const float* __restrict__ ws, int splits, int64_t slab, float* __restrict__ y,
                                                                  const float* __restrict__ d, const float* __restrict__ noise,
                                                                  int64_t noise_batch_stride, const float* __restrict__ noise_w,
                                                                  const float* __restrict__ bias, const float* __restrict__ rgb_w,
                                                                  const float* __restrict__ rgb_s, int s_stride, float rgb_wscale,
                                                                  float* __restrict__ rgb_part, int cout, int plane,
                                                                  const maua_frame_source_t* __restrict__ src, int noise_slot) {
    constexpr int CL = 256 / PT;   // channel lanes
    constexpr int NC = 32 / CL;    // channels per thread
    __shared__ float part[3][CL][PT];
    const int tid = threadIdx.x, px = tid % PT, cl = tid / PT;
    const int ptiles = plane / PT, groups = cout / 32;
    int t = blockIdx.x;
    const int pt = t % ptiles;
    t /= ptiles;
    const int gidx = t % groups, b = t / groups;
    const int pix = pt * PT + px;
    if (src) {
        noise_batch_stride = src->noise_stride[noise_slot];
        noise = src->noise[noise_slot];
        if (noise) noise += (int64_t)src->frame0 * noise_batch_stride;
    }
    const float nw = noise ? noise_w[0] : 0.f;
    const float nzv = nw != 0.f ? noise[(size_t)b * noise_batch_stride + pix] : 0.f;
    float dv[NC], bv[NC], ms[NC], w0[NC], w1[NC], w2[NC], v[NC];
    int64_t idx[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = gidx * 32 + cl + CL * q;
        const int64_t bc = (int64_t)b * cout + c;
        idx[q] = bc * plane + pix;
        dv[q] = d ? d[bc] : 1.f;
        bv[q] = bias ? bias[c] : 0.f;
        ms[q] = rgb_s[(size_t)b * s_stride + c];
        w0[q] = rgb_w[c], w1[q] = rgb_w[cout + c], w2[q] = rgb_w[2 * cout + c];
        v[q] = 0.f;
    }
    int sp = 0;
    for (; sp + 8 <= splits; sp += 8) {  // eight slabs of every element in flight
        float a[NC][8];
#pragma unroll
        for (int q = 0; q < NC; ++q)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[q][k] = ws[(size_t)(sp + k) * slab + idx[q]];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            v[q] += (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);  // (reduce_tail_kernel's association)
            v[q] += (a[q][4] + a[q][5]) + (a[q][6] + a[q][7]);
        }
    }
    for (; sp + 4 <= splits; sp += 4) {
        float a[NC][4];
#pragma unroll
        for (int q = 0; q < NC; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[q][k] = ws[(size_t)(sp + k) * slab + idx[q]];
#pragma unroll
        for (int q = 0; q < NC; ++q) v[q] += (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);
    }
    for (; sp < splits; ++sp)
#pragma unroll
        for (int q = 0; q < NC; ++q) v[q] += ws[(size_t)sp * slab + idx[q]];
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const float t2 = slab_tail(v[q], dv[q], nw, nzv, bv[q], true);
        y[idx[q]] = t2;
        // modulated ToRGB weights as torgb_kernel forms them: (wscale * w) * s
        r0 = fmaf((rgb_wscale * w0[q]) * ms[q], t2, r0);
        r1 = fmaf((rgb_wscale * w1[q]) * ms[q], t2, r1);
        r2 = fmaf((rgb_wscale * w2[q]) * ms[q], t2, r2);
    }
    part[0][cl][px] = r0, part[1][cl][px] = r1, part[2][cl][px] = r2;
    __syncthreads();
    if (tid < 3 * PT) {
        const int col = tid / PT, q = tid % PT;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < CL; ++k) a += part[col][k][q];
        rgb_part[((size_t)b * 3 * groups + 3 * gidx + col) * plane + pt * PT + q] = a;
    }
}

// wp[tap][i][o_pad] = w[o][i][tap], wsq[o][i] = sum_tap w^2.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                          float* __restrict__ wsq, int cout, int cout_pad, int cin,
                                                          int ktaps) {
    const int64_t total = (int64_t)cout_pad * cin;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int o = (int)(idx % cout_pad);
        const int i = (int)(idx / cout_pad);
        float ss = 0.f;
        for (int tp = 0; tp < ktaps; ++tp) {
            const float v = (o < cout) ? w[((size_t)o * cin + i) * ktaps + tp] : 0.f;
            if (wp) wp[((size_t)tp * cin + i) * cout_pad + o] = v;
            ss = fmaf(v, v, ss);
        }
        if (wsq && o < cout) wsq[(size_t)o * cin + i] = ss;
    }
}

// Column layout of the Winograd weight packs (modes 2 and 3).  Layers above 32 output channels run with two 32-channel
// M-tiles per wave; their columns are interleaved [lane][m-tile] inside every group of 64 channels (and padded to a multiple
// of 64) so that a lane fetches both A operands with one 8-byte LDS read.  Destination column -> source channel:
__host__ __device__ inline int wino_cout_pad(int cout) { return cout > 32 ? (cout + 63) / 64 * 64 : (cout + 31) / 32 * 32; }
__device__ inline int wino_src_channel(int od, int cout) {
    return cout > 32 ? (od & ~63) + ((od & 1) << 5) + ((od & 63) >> 1) : od;
}

// Winograd F(2,3) weight transform along kx, tap-major repack: wq[(ky*4 + xi)][i][o_pad],
//   xi 0: g0   1: (g0+g1+g2)/2   2: (g0-g1+g2)/2   3: g2        (g = W[o][i][ky][0..2])
__global__ __launch_bounds__(256) void pack_weight_wino_kernel(const float* __restrict__ w, float* __restrict__ wq, int cout,
                                                               int cout_pad, int cin) {
    const int64_t total = (int64_t)cout_pad * cin;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int od = (int)(idx % cout_pad);  // destination column
        const int o = wino_src_channel(od, cout);
        const int i = (int)(idx / cout_pad);
        for (int ky = 0; ky < 3; ++ky) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (o < cout) {
                const float* gp = w + (((size_t)o * cin + i) * 3 + ky) * 3;
                g0 = gp[0], g1 = gp[1], g2 = gp[2];
            }
            const float u[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
            for (int xi = 0; xi < 4; ++xi) wq[((size_t)(ky * 4 + xi) * cin + i) * cout_pad + od] = u[xi];
        }
    }
}

// Winograd F(4,3) weight transform along kx (G g, interpolation points 0, +-1, +-2, inf): wq[(ky*6 + xi)][i][o_pad],
//   xi 0: g0/4   1: -(g0+g1+g2)/6   2: -(g0-g1+g2)/6   3: (g0+2g1+4g2)/24   4: (g0-2g1+4g2)/24   5: g2
__global__ __launch_bounds__(256) void pack_weight_wino43_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                                 int cout, int cout_pad, int cin) {
    const int64_t total = (int64_t)cout_pad * cin;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int od = (int)(idx % cout_pad);  // destination column
        const int o = wino_src_channel(od, cout);
        const int i = (int)(idx / cout_pad);
        for (int ky = 0; ky < 3; ++ky) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (o < cout) {
                const float* gp = w + (((size_t)o * cin + i) * 3 + ky) * 3;
                g0 = gp[0], g1 = gp[1], g2 = gp[2];
            }
            const float u[6] = {g0 * 0.25f,
                                -(g0 + g1 + g2) * (1.f / 6.f),
                                -(g0 - g1 + g2) * (1.f / 6.f),
                                (g0 + 2.f * g1 + 4.f * g2) * (1.f / 24.f),
                                (g0 - 2.f * g1 + 4.f * g2) * (1.f / 24.f),
                                g2};
            for (int xi = 0; xi < 6; ++xi) wq[((size_t)(ky * 6 + xi) * cin + i) * cout_pad + od] = u[xi];
        }
    }
}

// Transposed conv, F(2,2) on the even x-phase: wq[(ky*4 + j)][i][o_pad],  j 0: g2   1: g0 + g2   2: g0   3: g1
__global__ __launch_bounds__(256) void pack_weight_upwino_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                                 int cout, int cout_pad, int cin) {
    const int64_t total = (int64_t)cout_pad * cin;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int o = (int)(idx % cout_pad);
        const int i = (int)(idx / cout_pad);
        for (int ky = 0; ky < 3; ++ky) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (o < cout) {
                const float* gp = w + (((size_t)o * cin + i) * 3 + ky) * 3;
                g0 = gp[0], g1 = gp[1], g2 = gp[2];
            }
            const float u[4] = {g2, g0 + g2, g0, g1};
            for (int j = 0; j < 4; ++j) wq[((size_t)(ky * 4 + j) * cin + i) * cout_pad + o] = u[j];
        }
    }
}

inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
inline int pow2_ceil(int v) { return 1 << ilog2(v); }

struct Plan {
    int bm, bn, wm;
    bool fallback = false;
    ConvGeom g;
    size_t lds_bytes;
    int64_t blocks;
};

int pad32(int c) { return (c + 31) / 32 * 32; }
#ifdef MAUA_EXPERIMENTS
int g_conv_debug = 0;
int g_conv_cfg = 0;  // tuning key 2: bit0 -> Cout<=64 uses 64x128 (WM 2); bit1 -> Cout<=32 uses 32x512; bit2 -> no flat runs
#endif

// Tile-shape selection (host).  BM follows Cout; the pixel tile is a stack of 32-pixel MFMA groups.
// mode 0 plain, 1 transposed stride 2, 2 plain through Winograd F(2,3) along x (needs an even width)
Plan make_plan(int batch, int cin, int cout, int h, int w, int mode) {
    Plan pl{};
    ConvGeom& g = pl.g;
    const bool up = mode == 1 || mode == 4, uw = mode == 4, wino = mode == 2 || mode == 3, w43 = mode == 3;
    const int wx = w43 ? 4 : 2;  // outputs per Winograd position
    g.B = batch, g.Cin = cin, g.Cout = cout, g.CoutPad = wino ? wino_cout_pad(cout) : pad32(cout), g.H = h, g.W = w;
    if (w43) {
        g.GH = h, g.GW = w / 4, g.OH = h, g.OW = w;  // positions are output quads
        if (cout <= 32) pl.bm = 32, pl.wm = 1, pl.bn = 128;
        else pl.bm = 64, pl.wm = 1, pl.bn = 128;  // (a 128-row tile holds 192 accumulator registers per wave as well, but its
                                                  // 18-row weight tile only leaves room for 2-channel chunks: 4-7 % slower)
    } else if (wino) {
        g.GH = h, g.GW = w / 2, g.OH = h, g.OW = w;  // positions are output pairs
        if (cout <= 32) pl.bm = 32, pl.wm = 1, pl.bn = MAUA_CFG(2) ? 256 : 128;
        else if (cout <= 64) pl.bm = 64, pl.wm = 1, pl.bn = 128;
        else pl.bm = 128, pl.wm = 2, pl.bn = 64;
    } else if (uw) {
        g.GH = h + 1, g.GW = w / 2 + 1, g.OH = 2 * h + 1, g.OW = 2 * w + 1;  // positions are pairs; W + 2 positions per row
        if (cout <= 32) pl.bm = 32, pl.wm = 1, pl.bn = 128;
        else pl.bm = 64, pl.wm = 2, pl.bn = 64;
    } else if (up) {
        g.GH = h + 1, g.GW = w + 1, g.OH = 2 * h + 1, g.OW = 2 * w + 1;
        if (cout <= 32) pl.bm = 32, pl.wm = 1, pl.bn = 128;
        else pl.bm = 64, pl.wm = 2, pl.bn = 64;
    } else {
        g.GH = h, g.GW = w, g.OH = h, g.OW = w;
        if (cout <= 32) pl.bm = 32, pl.wm = 1, pl.bn = MAUA_CFG(2) ? 512 : 256;
        else if (cout <= 64) pl.bm = 64, pl.wm = MAUA_CFG(1) ? 2 : 1, pl.bn = MAUA_CFG(1) ? 128 : 256;
        else pl.bm = 128, pl.wm = 2, pl.bn = 128;
    }
    auto shape = [&](int bn) {
        const int nsub = bn / 32;
        int sw = up ? 8 : (wino ? 16 : 32);
        if (sw > pow2_ceil(g.GW)) sw = pow2_ceil(g.GW);
        const int sh = 32 / sw;
        int nsx = pow2_ceil(ceil_div(g.GW, sw));
        if (nsx > nsub) nsx = nsub;
        if (!up && nsx > 1) nsx = 1;  // plain: keep 32-wide row segments, stack rows instead
        int nsy = pow2_ceil(ceil_div(g.GH, sh));
        if (nsy > nsub / nsx) nsy = nsub / nsx;
        const int ni = nsub / (nsx * nsy);
        g.lsw = ilog2(sw), g.lsh = ilog2(sh), g.lnsx = ilog2(nsx), g.lnsy = ilog2(nsy), g.lni = ilog2(ni);
        const int tw = sw * nsx, th = sh * nsy;
        g.tiles_x = ceil_div(g.GW, tw), g.tiles_y = ceil_div(g.GH, th), g.img_groups = ceil_div(batch, ni);
        g.PH = th + 2, g.PW = (wino ? wx * tw : tw) + 2;
        // LDS row stride: with sub-tiles narrower than 32 pixels the 32 lanes of one MFMA group read SH rows at once;
        // a stride of SW * odd puts the rows on disjoint bank groups (conflict-free ds_read_b32)
        g.PWS = g.PW;
        if (sw < 32 && !wino) {  // (Winograd reads 8-byte pairs: the even PW is kept)
            while (g.PWS % (2 * sw) != sw) ++g.PWS;
            if (ni * g.PH * g.PWS > 512) g.PWS = g.PW;  // tiny maps, many images per tile: take the conflicts
        }
        g.PSTRIDE = ni * g.PH * g.PWS;
    };
    shape(pl.bn);
    if (up && g.GH * g.GW > pl.bn && !MAUA_CFG(4)) {
        // The (H+1)x(W+1) position grid never fits power-of-two 2-D tiles (29 % idle MFMA columns at 65x65); tiles are
        // instead runs of BN consecutive positions of the flattened grid, one image each.  A position reads inputs
        // p, p-1, p-GW, p-GW-1 of the pitch-GW flattened (zero-padded) input: two runs of BN+1 floats per channel.
        // (From two runs per image on — round 6: the 9 x 9 grid of the 8^2 -> 16^2 layer took 16 x 16 = 256 slots as a 2-D tile, 32 % of its
        // MFMA columns useful; as two runs of 64 it is 63 %.)
        g.flat = 1;
        g.lsw = 5, g.lsh = 0, g.lnsx = ilog2(pl.bn / 32), g.lnsy = 0, g.lni = 0;
        g.tiles_x = ceil_div(g.GH * g.GW, pl.bn), g.tiles_y = 1, g.img_groups = batch;
        g.PH = 2, g.PW = (uw ? 2 : 1) * pl.bn + 1, g.PWS = (uw ? 2 : 1) * pl.bn + 2;
        g.PSTRIDE = g.PH * g.PWS;
    }
    if (g.PSTRIDE > ((pl.bn >= 512 || (wino && pl.bn >= 256) || ((w43 || uw) && pl.bn >= 128)) ? 768 : 512)) {  // tiny feature maps under a wide-N config: fall back to the 128-pixel tile
        if (w43) pl.bm = 64, pl.wm = 1, pl.bn = 128;
        else if (wino) pl.bm = 128, pl.wm = 2, pl.bn = 64;
        else if (up) pl.bm = 64, pl.wm = 2, pl.bn = 64;
        else pl.bm = 128, pl.wm = 2, pl.bn = 128;
        shape(pl.bn);
        pl.fallback = true;
    }
    const int CC = chunk_channels(pl.bm, pl.bn, mode);
    g.n_chunks = ceil_div(cin, CC);
    g.m_tiles = ceil_div(g.CoutPad, pl.bm);
    g.n_tiles = g.tiles_x * g.tiles_y * g.img_groups;
    // split-K until the grid covers the chip ~2x (256 CUs), never below 2 chunks per split
    const int64_t base_blocks = (int64_t)g.m_tiles * g.n_tiles;
    int splits = 1;
    while (base_blocks * splits < 512 && g.n_chunks / (splits * 2) >= 2) splits *= 2;
    g.splits = splits;
    g.chunks_per_split = ceil_div(g.n_chunks, splits);
    g.splits = ceil_div(g.n_chunks, g.chunks_per_split);
    g.ws_slab = (int64_t)batch * cout * g.OH * g.OW;
    pl.blocks = base_blocks * g.splits;
    pl.lds_bytes = 2 * ((size_t)(w43 ? 18 : (wino || uw) ? 12 : 9) * CC * pl.bm + (size_t)CC * g.PSTRIDE) * sizeof(float);
    if (g.lni == 0)  // the DMA patch path also stages the styles of one image
        pl.lds_bytes += (size_t)cin * sizeof(float);
    if (pl.lds_bytes < (size_t)2 * pl.bm * sizeof(float)) pl.lds_bytes = (size_t)2 * pl.bm * sizeof(float);
    return pl;
}

char g_last_instance[96] = "";

template <int BM, int BN, int WM, int UP, bool MULTI, bool FAST, int MAXP>
int launch_conv_impl2(const Plan& pl, const ConvPtrs& ptrs, hipStream_t st) {
    auto kern = modconv_mfma_kernel<BM, BN, WM, UP, MULTI, FAST, MAXP>;
    snprintf(g_last_instance, sizeof(g_last_instance), "modconv_mfma_kernel<%d, %d, %d, %d, %s, %s, %d>", BM, BN, WM, UP,
             MULTI ? "true" : "false", FAST ? "true" : "false", MAXP);
    static unsigned long long lds_ok = 0;  // per launcher: devices on which the attribute has been set (common.h)
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(kern), &lds_ok, 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)pl.blocks), dim3(256), pl.lds_bytes, st, pl.g, ptrs);
    MAUA_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WM, int UP, bool MULTI, bool FAST>
int launch_conv_impl(const Plan& pl, const ConvPtrs& ptrs, hipStream_t st) {
    if (pl.g.PSTRIDE <= 256) return launch_conv_impl2<BM, BN, WM, UP, MULTI, FAST, 1>(pl, ptrs, st);
    constexpr bool WIDE = BN >= 512 || (UP == 2 && BN >= 256) || ((UP == 3 || UP == 4) && BN >= 128);  // configs whose patch can exceed 512 floats per channel
    if (pl.g.PSTRIDE <= 512 || !WIDE) return launch_conv_impl2<BM, BN, WM, UP, MULTI, FAST, 2>(pl, ptrs, st);
    return launch_conv_impl2<BM, BN, WM, UP, MULTI, FAST, (WIDE ? 3 : 2)>(pl, ptrs, st);
}

template <int BM, int BN, int WM, int UP>
int launch_conv(const Plan& pl, const ConvPtrs& ptrs, hipStream_t st) {
    constexpr int CC = chunk_channels(BM, BN, UP);
    if (pl.g.PSTRIDE > 768) return MAUA_EINVAL;
    const bool fast = (pl.g.Cin % CC == 0) && (pl.g.CoutPad % BM == 0);
    if (pl.g.lni > 0) return fast ? launch_conv_impl<BM, BN, WM, UP, true, true>(pl, ptrs, st)
                                  : launch_conv_impl<BM, BN, WM, UP, true, false>(pl, ptrs, st);
    return fast ? launch_conv_impl<BM, BN, WM, UP, false, true>(pl, ptrs, st)
                : launch_conv_impl<BM, BN, WM, UP, false, false>(pl, ptrs, st);
}

}  // namespace

#ifdef MAUA_EXPERIMENTS
int maua_conv_debug_set(int v) { g_conv_debug = v; return 0; }
int maua_conv_cfg_set(int v) { g_conv_cfg = v; return 0; }
#endif

extern "C" int maua_pack_weight_f32(const float* w, float* wp, float* wsq, int cout, int cin, int ktaps, void* stream) {
    if (!w || cout <= 0 || cin <= 0 || ktaps <= 0) return MAUA_EINVAL;
    const int cout_pad = pad32(cout);
    const int64_t total = (int64_t)cout_pad * cin;
    const int64_t blocks = ceil_div64(total, 256);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, w, wp, wsq, cout, cout_pad, cin, ktaps);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_pack_weight_wino_f32(const float* w, float* wq, int cout, int cin, void* stream) {
    if (!w || !wq || cout <= 0 || cin <= 0) return MAUA_EINVAL;
    const int cout_pad = wino_cout_pad(cout);
    const int64_t blocks = ceil_div64((int64_t)cout_pad * cin, 256);
    hipLaunchKernelGGL(pack_weight_wino_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, w, wq, cout, cout_pad, cin);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_pack_weight_upwino_f32(const float* w, float* wq, int cout, int cin, void* stream) {
    if (!w || !wq || cout <= 0 || cin <= 0) return MAUA_EINVAL;
    const int cout_pad = pad32(cout);
    const int64_t blocks = ceil_div64((int64_t)cout_pad * cin, 256);
    hipLaunchKernelGGL(pack_weight_upwino_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, w, wq, cout, cout_pad, cin);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_pack_weight_wino43_f32(const float* w, float* wq, int cout, int cin, void* stream) {
    if (!w || !wq || cout <= 0 || cin <= 0) return MAUA_EINVAL;
    const int cout_pad = wino_cout_pad(cout);
    const int64_t blocks = ceil_div64((int64_t)cout_pad * cin, 256);
    hipLaunchKernelGGL(pack_weight_wino43_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, w, wq, cout, cout_pad, cin);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t maua_modconv_ws_floats(int batch, int cin, int cout, int h, int w, int up) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || up == 5 || up == 7) return 0;
    if (up == 6 || up == 8) return maua_up2d_ws_floats(batch, cin, h);  // the exported last input column
    Plan pl = make_plan(batch, cin, cout, h, w, up);
    return pl.g.splits > 1 ? pl.g.ws_slab * pl.g.splits : 0;
}

// Name of the kernel template instance the LAST maua_modconv3x3_f32 / maua_styledconv_torgb_f32 call of this process
// launched, as rocprofv3 prints it ("modconv_mfma_kernel<BM, BN, WM, MODE, MULTI, FAST, MAXP>"): lets bench.py join its live
// timings with the per-instance PMC tables under profiles/ without re-implementing the plan.
extern "C" int maua_modconv_last_instance(char* buf, int buf_len) {
    if (!buf || buf_len <= 0) return MAUA_EINVAL;
    snprintf(buf, (size_t)buf_len, "%s", g_last_instance);
    return 0;
}

namespace {
struct RgbArgs {
    const float* w; const float* s; const float* bias; const float* skip; const float* k4; float* out;
    float wscale; int store_features; uint8_t* u8;
};

int modconv_impl(const float* x, const float* wp, const float* s, int s_stride, const float* d, float* y, int batch,
                 int cin, int cout, int h, int w, int up, float wscale, int fuse_act, const float* noise,
                 int64_t noise_batch_stride, const float* noise_w, const float* bias, float* ws, const RgbArgs* rgb,
                 const maua_frame_source_t* src, int noise_slot, const float* post_s, void* stream, int* partial_splits = nullptr) {
    // partial_splits != NULL (modes 0 .. 3, the low-resolution entries): the convolution leaves its split-K slabs (one slab when K is not
    // split) in ws and the caller reduces them; *partial_splits receives the slab count
    if (!x || !wp || !y || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return MAUA_EINVAL;
    if (partial_splits && (up > 3 || rgb || !ws)) return MAUA_EINVAL;
    // the style fold (include/maua_hip.h): s == NULL = x arrives multiplied by this layer's styles — the 2-D Winograd and the F(2,2)^2
    // transposed kernels have instances without the multiply; post_s = the consumer's styles, applied to the stored map by the
    // 2-D Winograd kernels' epilogue
    if ((!s && up != 5 && up != 6) || (post_s && up != 5)) return MAUA_ENOSYS;
    if ((noise || (src && fuse_act)) && !noise_w) return MAUA_EINVAL;
    if (src && (noise_slot < 0 || noise_slot >= MAUA_MAX_NOISE_SLOTS)) return MAUA_EINVAL;
    if (up == 5) {  // 2-D Winograd F(2x4, 3x3), modconv_w2d.hip
        const int rc = maua_w2d_launch(x, wp, s, s_stride, d, y, batch, cin, cout, h, w, wscale, fuse_act, noise, noise_batch_stride,
                                       noise_w, bias, rgb ? rgb->w : nullptr, rgb ? rgb->s : nullptr, rgb ? rgb->wscale : 0.f,
                                       rgb ? rgb->bias : nullptr, rgb ? rgb->skip : nullptr, rgb ? rgb->k4 : nullptr,
                                       rgb ? rgb->out : nullptr, rgb ? rgb->u8 : nullptr, rgb ? (rgb->store_features ? 1 : 2) : 0,
                                       src, noise_slot, post_s, stream);
        if (rc == 0) snprintf(g_last_instance, sizeof(g_last_instance), "%s", maua_w2d_last_instance());
        return rc;
    }
    if (up == 7 || up == 8) {  // plain (7) / transposed (8) conv with split-bf16 products (side measurement), modconv_sbf16.hip
        if (rgb) return MAUA_ENOSYS;
        const int rc = maua_sbf16_launch(x, wp, s, s_stride, d, y, ws, batch, cin, cout, h, w, up == 8, wscale, fuse_act, noise,
                                         noise_batch_stride, noise_w, bias, src, noise_slot, stream);
        if (rc == 0) snprintf(g_last_instance, sizeof(g_last_instance), "%s", maua_sbf16_last_instance());
        return rc;
    }
    if (up == 6) {  // transposed conv, F(2,2) on both axes, modconv_up2d.hip: raw output only (the blur kernel applies the tail)
        if (fuse_act || rgb) return MAUA_EINVAL;
        const int rc = maua_up2d_launch(x, wp, s, s_stride, d, y, ws, batch, cin, cout, h, w, wscale, stream);
        if (rc == 0) snprintf(g_last_instance, sizeof(g_last_instance), "%s", maua_up2d_last_instance());
        return rc;
    }
    if ((int64_t)batch * cin * h * w > 0x7fffffffLL) return MAUA_EINVAL;  // 32-bit patch offsets
    if (up < 0 || up > 4 || ((up == 2 || up == 4) && (w & 1)) || (up == 3 && (w & 3))) return MAUA_EINVAL;
    if (up == 4 && (fuse_act || rgb)) return MAUA_EINVAL;  // raw output only: the blur kernel applies the tail
    Plan pl = make_plan(batch, cin, cout, h, w, up);
    if (pl.g.splits > 1 && !ws) return MAUA_EINVAL;
    pl.g.force_ws = partial_splits ? 1 : 0;
    pl.g.s_stride = s_stride;
    pl.g.wscale = wscale;
    pl.g.fuse_act = fuse_act;
    pl.g.noise_batch_stride = noise_batch_stride;
#ifdef MAUA_EXPERIMENTS
    pl.g.debug = g_conv_debug;
#endif
    pl.g.rgb = 0;
    pl.g.rgb_wscale = 0.f;
    ConvPtrs ptrs{x, wp, s, d, noise, noise_w, bias, y, ws, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, src, noise_slot};
    if (rgb) {
        // fusable only when one workgroup holds every channel of its pixels in a single wave row (BM >= Cout, WM == 1),
        // no split-K, one image per tile, the tail fused
        const bool ok = up != 1 && up != 4 && fuse_act && pl.wm == 1 && pl.g.m_tiles == 1 && pl.g.splits == 1 && pl.g.lni == 0 &&
                        rgb->w && rgb->s && rgb->bias && (rgb->out || rgb->u8) && (!rgb->skip || (rgb->k4 && !(h & 1) && !(w & 1)));
        if (!ok) return MAUA_ENOSYS;
        pl.g.rgb = rgb->store_features ? 1 : 2;
        pl.g.rgb_wscale = rgb->wscale;
        ptrs.rgb_w = rgb->w, ptrs.rgb_s = rgb->s, ptrs.rgb_bias = rgb->bias, ptrs.rgb_skip = rgb->skip;
        ptrs.rgb_k4 = rgb->k4, ptrs.rgb_out = rgb->out, ptrs.rgb_u8 = rgb->u8;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (up == 4) {
        if (!pl.g.flat || pl.fallback) return MAUA_EINVAL;  // grids too small for flat pair runs use mode 1
        if (pl.bm == 32) rc = launch_conv<32, 128, 1, 4>(pl, ptrs, st);
        else rc = launch_conv<64, 64, 2, 4>(pl, ptrs, st);
    } else if (up == 3) {
        if (pl.bm == 32) rc = launch_conv<32, 128, 1, 3>(pl, ptrs, st);
        else rc = launch_conv<64, 128, 1, 3>(pl, ptrs, st);
    } else if (up == 2) {
        if (pl.bm == 32 && pl.bn == 256) rc = launch_conv<32, 256, 1, 2>(pl, ptrs, st);
        else if (pl.bm == 32) rc = launch_conv<32, 128, 1, 2>(pl, ptrs, st);
        else if (pl.bm == 64) rc = launch_conv<64, 128, 1, 2>(pl, ptrs, st);
        else rc = launch_conv<128, 64, 2, 2>(pl, ptrs, st);
    } else if (up == 1) {
        if (pl.fallback) rc = launch_conv<64, 64, 2, 1>(pl, ptrs, st);
        else if (pl.bm == 32) rc = launch_conv<32, 128, 1, 1>(pl, ptrs, st);
        else rc = launch_conv<64, 64, 2, 1>(pl, ptrs, st);
    } else {
        if (pl.bm == 32 && pl.bn == 512) rc = launch_conv<32, 512, 1, 0>(pl, ptrs, st);
        else if (pl.bm == 32) rc = launch_conv<32, 256, 1, 0>(pl, ptrs, st);
        else if (pl.bm == 64 && pl.bn == 128) rc = launch_conv<64, 128, 2, 0>(pl, ptrs, st);
        else if (pl.bm == 64) rc = launch_conv<64, 256, 1, 0>(pl, ptrs, st);
        else rc = launch_conv<128, 128, 2, 0>(pl, ptrs, st);
    }
    if (rc) return rc;
    if (partial_splits) {
        *partial_splits = pl.g.splits;
        return 0;
    }
    if (pl.g.splits > 1) {
        const int64_t total = pl.g.ws_slab;
        const int64_t blocks = ceil_div64(total, 256);
        hipLaunchKernelGGL(reduce_tail_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, ws,
                           pl.g.splits, pl.g.ws_slab, y, d, noise, noise_batch_stride, noise_w, bias, fuse_act, cout,
                           (int64_t)pl.g.OH * pl.g.OW, total, src, noise_slot);
        MAUA_LAUNCH_CHECK();
    }
    return 0;
}
}  // namespace

extern "C" int maua_modconv3x3_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d,
                                   float* y, int batch, int cin, int cout, int h, int w, int up, float wscale,
                                   int fuse_act, const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                   const float* bias, float* ws, const maua_frame_source_t* src, int noise_slot, void* stream) {
    return modconv_impl(x, wp, s, s_stride, d, y, batch, cin, cout, h, w, up, wscale, fuse_act, noise, noise_batch_stride,
                        noise_w, bias, ws, nullptr, src, noise_slot, nullptr, stream);
}

// ---- low-resolution entries: convolution (modes 0 / 1) + the fused reducers above
extern "C" int maua_lowres_ok(int cin, int cout, int h, int w, int up) {
    if (cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (up != 0 && up != 1 && up != 2 && up != 3 && up != 6)) return 0;
    if (up == 6) return 4 * h * w <= 1024 && maua_up2d16_ok(cin, cout, h, w);  // the F(2,2)^2 kernel's 16-column tiles (modconv_up2d.hip)
    if (up == 1) return 4 * h * w <= 1024;                              // reduce_blur_tail_kernel: at most four outputs per thread
    if ((up == 2 && (w & 1)) || (up == 3 && (w & 3))) return 0;         // plain layer through Winograd F(2,3) / F(4,3) along x
    return cout % 32 == 0 && (h * w) % 16 == 0 && h * w <= 1024;        // reduce_tail_rgbpart_kernel: 32-channel groups, 16 / 32-pixel tiles
}

extern "C" int64_t maua_lowres_ws_floats(int batch, int cin, int cout, int h, int w, int up) {
    if (batch <= 0 || !maua_lowres_ok(cin, cout, h, w, up)) return 0;
    if (up == 6)  // slabs + the exported last input column
        return (int64_t)maua_up2d16_splits(batch, cin, cout, h, w) * batch * cout * (2 * h + 1) * (2 * w + 1) + maua_up2d_ws_floats(batch, cin, h);
    Plan pl = make_plan(batch, cin, cout, h, w, up);
    return pl.g.ws_slab * pl.g.splits;
}

extern "C" int maua_upconv_blur_lowres_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d, float* y, float* ws,
                                           const float* k4, const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                           const float* bias, const maua_frame_source_t* src, int noise_slot, int batch, int cin, int cout,
                                           int h, int w, int up, float wscale, const float* post_s, void* stream) {
    if (!x || !wp || !s || !y || !ws || !k4 || batch <= 0 || (up != 1 && up != 6)) return MAUA_EINVAL;
    if (!maua_lowres_ok(cin, cout, h, w, up)) return MAUA_ENOSYS;
    if ((noise || src) && !noise_w) return MAUA_EINVAL;
    if (src && (noise_slot < 0 || noise_slot >= MAUA_MAX_NOISE_SLOTS)) return MAUA_EINVAL;
    int splits = 0;
    const int64_t slab = (int64_t)batch * cout * (2 * h + 1) * (2 * w + 1);
    if (up == 6) {  // F(2,2) on both axes (wp = maua_pack_weight_up2d_f32), K split over workgroups; the exported column behind the slabs
        float* xcol = ws + (int64_t)maua_up2d16_splits(batch, cin, cout, h, w) * slab;
        if (int rc = maua_up2d16_launch(x, wp, s, s_stride, ws, xcol, batch, cin, cout, h, w, wscale, &splits, stream)) return rc;
        snprintf(g_last_instance, sizeof(g_last_instance), "%s", maua_up2d_last_instance());
    } else if (int rc = modconv_impl(x, wp, s, s_stride, nullptr, y, batch, cin, cout, h, w, 1, wscale, 0, nullptr, 0, nullptr, nullptr, ws,
                                     nullptr, nullptr, 0, nullptr, stream, &splits))  // (the convolution writes slabs only: y is a placeholder)
        return rc;
    const size_t lds = (size_t)(2 * h + 3) * (2 * w + 3) * sizeof(float);
    hipLaunchKernelGGL(reduce_blur_tail_kernel, dim3((unsigned)(batch * cout)), dim3(256), lds, (hipStream_t)stream, ws, splits, slab, y, d, k4,
                       noise, noise_batch_stride, noise_w, bias, post_s, s_stride, cout, h, w, src, noise_slot, up == 6 ? 1 : 0);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_styledconv_rgbpart_lowres_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d, float* y,
                                                  float* ws, const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                                  const float* bias, const float* rgb_w, const float* rgb_s, float rgb_wscale,
                                                  float* rgb_partial, const maua_frame_source_t* src, int noise_slot, int batch, int cin,
                                                  int cout, int h, int w, int mode, float wscale, void* stream) {
    if (!x || !wp || !s || !y || !ws || !rgb_w || !rgb_s || !rgb_partial || batch <= 0 || (mode != 0 && mode != 2 && mode != 3)) return MAUA_EINVAL;
    if (!maua_lowres_ok(cin, cout, h, w, mode)) return MAUA_ENOSYS;
    if ((noise || src) && !noise_w) return MAUA_EINVAL;
    if (src && (noise_slot < 0 || noise_slot >= MAUA_MAX_NOISE_SLOTS)) return MAUA_EINVAL;
    int splits = 0;
    if (int rc = modconv_impl(x, wp, s, s_stride, nullptr, y, batch, cin, cout, h, w, mode, wscale, 0, nullptr, 0, nullptr, nullptr, ws, nullptr,
                              nullptr, 0, nullptr, stream, &splits))
        return rc;
    const int64_t slab = (int64_t)batch * cout * h * w;
    const int pt = (h * w) % 32 == 0 ? 32 : 16;
    const int64_t blocks = (int64_t)batch * (cout / 32) * (h * w / pt);
    if (pt == 32)
        hipLaunchKernelGGL(reduce_tail_rgbpart_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, splits, slab, y, d, noise,
                           noise_batch_stride, noise_w, bias, rgb_w, rgb_s, s_stride, rgb_wscale, rgb_partial, cout, h * w, src, noise_slot);
    else
        hipLaunchKernelGGL(reduce_tail_rgbpart_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, splits, slab, y, d, noise,
                           noise_batch_stride, noise_w, bias, rgb_w, rgb_s, s_stride, rgb_wscale, rgb_partial, cout, h * w, src, noise_slot);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_styledconv_torgb_partial_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d,
                                                 float* y, int batch, int cin, int cout, int h, int w, int mode, float wscale,
                                                 const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                                 const float* bias, const float* rgb_w, const float* rgb_s, float rgb_wscale,
                                                 float* rgb_partial, const maua_frame_source_t* src, int noise_slot,
                                                 const float* post_s, void* stream) {
    if (mode != 5) return MAUA_ENOSYS;  // only the 2-D Winograd kernel leaves partial ToRGB sums
    if (!x || !wp || !y || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || ((noise || src) && !noise_w)) return MAUA_EINVAL;
    if (src && (noise_slot < 0 || noise_slot >= MAUA_MAX_NOISE_SLOTS)) return MAUA_EINVAL;
    const int rc = maua_w2d_launch(x, wp, s, s_stride, d, y, batch, cin, cout, h, w, wscale, 1, noise, noise_batch_stride, noise_w, bias,
                                   rgb_w, rgb_s, rgb_wscale, nullptr, nullptr, nullptr, rgb_partial, nullptr, 3, src, noise_slot, post_s, stream);
    if (rc == 0) snprintf(g_last_instance, sizeof(g_last_instance), "%s", maua_w2d_last_instance());
    return rc;
}

extern "C" int maua_styledconv_torgb_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d,
                                         float* y, int batch, int cin, int cout, int h, int w, int mode, float wscale,
                                         const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                         const float* bias, const float* rgb_w, const float* rgb_s, float rgb_wscale,
                                         const float* rgb_bias, const float* rgb_skip, const float* rgb_k4, float* rgb_out,
                                         int store_features, uint8_t* frames_u8, const maua_frame_source_t* src, int noise_slot,
                                         const float* post_s, void* stream) {
    RgbArgs rgb{rgb_w, rgb_s, rgb_bias, rgb_skip, rgb_k4, rgb_out, rgb_wscale, store_features, frames_u8};
    if (mode == 1) return MAUA_ENOSYS;
    return modconv_impl(x, wp, s, s_stride, d, y, batch, cin, cout, h, w, mode, wscale, 1, noise, noise_batch_stride, noise_w,
                        bias, nullptr, &rgb, src, noise_slot, post_s, stream);
}

// Stride-2 transposed 3x3 modulated convolution through F(2,2) on BOTH axes of its polyphase form, on the fp32 matrix cores
// (mode 6 of maua_modconv3x3_f32).
//
// Behavioural contract: /root/reference/models/stylegan2.py:229-237 (conv_transpose2d, stride 2, pad 0; the Blur that follows is
// maua_blur_noise_act_f32) in the input-scale -> shared-weight contraction -> output-demod formulation of modconv.hip.
//
// Polyphase form: position (p, q) of the (H+1) x (W+1) grid owns the outputs (2p + a', 2q + b'); along one axis the even output
// is a 2-tap correlation (taps 0 and 2 on inputs p, p-1), the odd output a 1-tap one (tap 1 on input p).  For a PAIR of
// positions the two even outputs come from three products instead of four,
//     m0 = g2 (d0 - d1),  m1 = (g0 + g2) d1,  m2 = g0 (d2 - d1);   e_p = m0 + m1,  e_p+1 = m1 + m2        (d = inputs p-1, p, p+1)
// (modconv.hip's mode 4 does this along x only: 30 products per 2x2 positions).  Applied along y as well, a 2x2 BLOCK of
// positions needs 9 (even, even) + 6 (even, odd) + 6 (odd, even) + 4 (odd, odd) = 25 products per (cin, cout) pair instead of 36
// (direct polyphase, mode 1): 25/36 of the matrix-core cycles of mode 1, 5/6 of mode 4's.  Transform constants are +-1: the fp32
// error stays at the level of the direct form (tests/test_winograd_algebra.py holds the identity and this kernel's index scheme).
//
//   window (3x3 inputs around the block, scaled by the style)  ->  row forms R = (r0 - r1, r1, r2 - r1, r2), column forms likewise
//   -> 16 operand values B[a][b];   transformed kernel: 16 entries U (9 ee + 3 eo + 3 oe + g11: the eo / oe / oo phases reuse one
//   entry for 2 / 2 / 4 products)   ->  25 accumulators per output channel, summed in pairs / quads in the epilogue.
//
// Work decomposition: a workgroup owns 32 output channels x (4 block rows x 16 block columns = 8 x 32 positions = 16 x 64
// output pixels); wave w takes block row w: 25 x 2 accumulator tiles of v_mfma_f32_16x16x4_f32 (200 registers), 50 MFMAs per
// 4-channel K step against 23 VALU (9 style multiplies + 14 subtractions) and 22 LDS reads (6 window + 16 weight rows).  No
// cross-wave exchange: every wave holds all 25 products of its blocks.  Operands reach LDS by MUBUF `buffer_load ... lds` DMA,
// double buffered, one barrier per K step; the packed weight (maua_pack_weight_up2d_f32) is stored in HBM as the LDS tile image.
//
// Measured and NOT kept (round 3, alternating runs inside bench.py): 4 instead of 8 channels per K step (+2.5 % on this family),
// s_setprio(1) around the MFMA phase (no change), persistent workgroups that walk several tiles and issue the next tile's first K
// step under the current tile's last one (-0.5..+1 % : two co-resident workgroups per CU already overlap one's prologue / epilogue
// with the other's K loop, and the tile loop costs 12 spilled registers per tile).
//
// The main kernel covers the (H/2) x (W/2) blocks of positions p < H, q < W (H, W powers of two on this path: tiles never hang
// over).  The remaining output row 2H and column 2W belong to positions whose own input is the zero padding: two 1-D polyphase
// transposed convolutions of the last input row / column with the kernel's last row / column, 1.5 MAC per output
// (up2d_edge_kernel: direct MFMA, ~1/H of the layer's work).
#include "common.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <queue>
#include <vector>
#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__)
#define MAUA_DEVICE_PASS 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4u __attribute__((aligned(4)));
typedef f32x2 f32x2u __attribute__((aligned(4)));

constexpr int U2_NU = 16;                                 // transformed-kernel entries per (cout, cin)
constexpr int U2_BM = 32;                                 // output channels per workgroup (two 16-row m-tiles, interleaved)
constexpr int U2_PROWS = 9;                               // staged input rows: 8 position rows + the row above
constexpr int U2_PSEGS = 9;                               // 16-byte segments per staged row: image columns tx0-4 .. tx0+31
constexpr int U2_PWS = 4 * U2_PSEGS;                      // LDS row stride (floats)
constexpr int U2_PLANE = 352;                             // floats per staged channel: 9 x 36 = 324, padded to 32 mod 64 so that the
                                                          // two K lane groups of a half-wave read disjoint bank halves (ds_read_b64)
// Input channels per K step (CC): 8 = two MFMA K groups per barrier wherever the channel count allows it (every layer of a real
// generator), 4 otherwise.  The packed weight layout depends on it: both sides derive it from cin with this one rule.
#ifndef MAUA_UP2D_MAX_CC
#define MAUA_UP2D_MAX_CC 8
#endif
__host__ __device__ constexpr int u2_cc(int cin) { return (MAUA_UP2D_MAX_CC >= 8 && cin % 8 == 0) ? 8 : 4; }
__host__ __device__ constexpr int u2_a_floats(int cc) { return U2_NU * cc * U2_BM; }       // weight tile: 2 cc DMA instructions of 1 KiB
__host__ __device__ constexpr int u2_p_instr(int cc) { return (cc * (U2_PLANE / 4) + 63) / 64; }  // patch: whole DMA instructions
__host__ __device__ constexpr int u2_pbuf(int cc) { return u2_p_instr(cc) * 256; }         // >= cc x 352 floats

// single `ds_read_b64` / `ds_read_b32` through inline assembly with explicit lgkmcnt waits: see modconv_w2d.hip (left alone the
// compiler pairs 8-byte reads into ds_read2_b64, which is serviced at half the bytes per clock on a 32-bank modulus)
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
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct Up2dArgs {
    const float* x;
    const float* wq;    // packed transformed kernel (LDS tile images), then the five edge tap matrices
    const float* s;
    const float* d;
    float* y;
    float* xcol;        // workspace [B][Cin][H]: the last input column, exported by the main kernel's right-edge tiles for the edge kernel
    int B, Cin, Cout, H, W;
    int s_stride;
    float wscale;
    int tiles_x, tiles_y, m_tiles, n_chunks;
    // ---- FUSE != 0: the Blur + noise + bias + leaky ReLU of the up-sampling StyledConv (reference models/stylegan2.py:229-238, :310-343) run
    // in this kernel's epilogue; yb [B, Cout, 2H, 2W] receives the activated map and the raw (2H+1) x (2W+1) map is never written.
    float* yb;
    const float* k4;       // [4][4] blur taps, SEPARABLE (row sums x column sums / total; checked by the launcher's caller)
    const float* noise;    // [B or 1, 1, 2H, 2W] or null
    const float* noise_w;  // [1]
    const float* bias;     // [Cout] or null
    int64_t noise_batch_stride;
    const maua_frame_source_t* src;  // frame source: noise from src->noise[noise_slot] at frame src->frame0
    int noise_slot;
    // FUSE == 2 (exact): tiles_y counts vertical SEGMENTS of seg_tiles tiles, walked by one workgroup (tiles_total_y = H / 8 + 1 tiles: the
    // last one holds raw row 2H only); an x tile keeps 60 of its 64 raw columns (tile column u: [60 u, 60 u + 60)); hbuf [B][Cout][n_seg - 1][6][2W]
    // receives the h-rows either side of a segment boundary for up2d_seam_kernel
    float* hbuf;
    int seg_tiles, tiles_total_y;
    // the style fold (round 6; include/maua_hip.h).  Producer side (FUSE == 2): post_s [B, s_stride] = the styles of the layer that consumes yb,
    // multiplied into the stored map.  Consumer side: s == NULL (instances with PRE = true) = x arrives multiplied by this layer's styles.
    const float* post_s;
    // FUSE == 0, the low-resolution entry (maua_upconv_blur_lowres_f32 with up = 6): K is split over `splits` workgroups per tile, split k
    // takes chunks [k * chunks_per_split, ...) and writes its partial raw map (gain = wscale, no demodulation) to y + k * slab
    int splits, chunks_per_split;
    int64_t slab;
#ifdef MAUA_EXPERIMENTS
    int real_blocks;       // FUSE == 1 (tools/fuse_probe.py): blocks beyond this number repeat earlier tiles (the price of an overlapped tiling)
#endif
};

// Compile-time ablation mask of the fused kernel, experiments builds only (results wrong by construction): 1 no second barrier / saved rows,
// 2 no seam exports, 4 no output stores, 16 aligned 64-column tiles that keep every lane (the halo-free tiling, with everything else of FUSE == 2)
#if !defined(MAUA_EXPERIMENTS)
#undef MAUA_FUSE_ABL
#endif
#ifndef MAUA_FUSE_ABL
#define MAUA_FUSE_ABL 0
#endif
#define FUSE_ABL(bits) ((MAUA_FUSE_ABL & (bits)) != 0)

// Separable factors of the 4 x 4 blur: K[a][b] (flipped taps: out[Y][X] = sum K[a][b] raw[Y - 1 + a][X - 1 + b]) = ky[a] * kx[b]
__device__ __forceinline__ void u2_blur_taps(const float* k4, float (&kx)[4], float (&ky)[4]) {
    float kk[4][4], total = 0.f, rs[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) kk[a][b] = k4[(3 - a) * 4 + (3 - b)], rs[a] += kk[a][b], cs[b] += kk[a][b], total += kk[a][b];
    const float inv = 1.f / total;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ky[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rs[a] * inv)));
        kx[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cs[a])));
    }
}

// FUSE: 0 = the raw (2H+1) x (2W+1) map (mode 6 of maua_modconv3x3_f32); 2 = the whole up-sampling StyledConv, exact (maua_upconv_blur_f32);
// 1 = experiments builds only: the fused epilogue with the tile halos taken as zero (the measurement that preceded the exact form)
// TW: position columns of a tile.  32 = 8 x 32 positions, wave w takes block row w (every layer from 32-wide inputs on); 16 = 16 x 16 positions
// for 16-wide inputs (round 6: the 16^2 -> 32^2 layer of a generator, which ran the polyphase kernel of modconv.hip at 0.46 of the matrix
// peak): wave w takes block rows 2 w, 2 w + 1 (lanes 0 .. 7 / 8 .. 15 of a K lane group), the staged patch is 17 rows of five 16-byte
// segments (340 of the 352 floats per channel plane), everything else — K loop, operand layout of the packed weight — is unchanged.
template <int CC, int FUSE = 0, bool PRE = false, int TW = 32>
__global__ __launch_bounds__(256, 2) void modconv_up2d_kernel(Up2dArgs p) {
    static_assert(TW == 32 || (TW == 16 && FUSE == 0), "16-column tiles exist for the raw-output form only");
    constexpr int T_ROWS = 256 / TW;              // position rows of a tile
    constexpr int U2_PROWS = T_ROWS + 1;          // staged input rows: the tile's rows + the row above   (shadow the 32-column constants)
    constexpr int U2_PSEGS = TW / 4 + 1;          // 16-byte segments per staged row: image columns tx0 - 4 .. tx0 + TW - 1
    constexpr int U2_PWS = 4 * U2_PSEGS;          // LDS row stride (floats)
    static_assert(U2_PROWS * U2_PWS <= U2_PLANE, "the staged plane must fit the channel pitch");
    constexpr int U2_A_FLOATS = u2_a_floats(CC), U2_PBUF = u2_pbuf(CC), U2_P_INSTR = u2_p_instr(CC);
    constexpr int A_PER_WAVE = 2 * CC / 4;                 // weight DMA instructions per wave and K step
    constexpr int P_PER_WAVE = (U2_P_INSTR + 3) / 4;       // patch DMA instructions per wave and K step (the last ones may be idle)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: As[2][A_FLOATS] | Ps[2][PBUF] | Ss[Cin] | Eg[32] | (FUSE == 2) SV[4][3][64][8]: the h-rows the next tile of the segment needs
    float* Ps = lds + 2 * U2_A_FLOATS;
    float* Ss = Ps + 2 * U2_PBUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's block row inside the tile
    const int j = lane & 15, kq = lane >> 4;

    int t = xcd_remap(blockIdx.x, gridDim.x);
#ifdef MAUA_EXPERIMENTS
    if (FUSE == 1 && t >= p.real_blocks) t -= p.real_blocks;  // (redundant tiles: see real_blocks)
#endif
    const int mt_id = t % p.m_tiles;
    t /= p.m_tiles;
    const int tile_x = t % p.tiles_x;
    t /= p.tiles_x;
    const int tile_y = t % p.tiles_y;                  // FUSE == 2: the vertical segment
    t /= p.tiles_y;
    const int b0 = FUSE == 0 ? t % p.B : t;
    const int split = FUSE == 0 ? t / p.B : 0;         // (FUSE == 0 with splits > 1: this workgroup's share of K)
    // first position column of the tile.  FUSE == 2: a tile keeps 60 of its 64 raw columns, [60 u, 60 u + 60): tile column 0 starts at position 0 and keeps its
    // lanes 0 .. 14 (the image's left padding is a true zero); the others start at the ODD position 30 u - 1 (raw column 60 u - 2) and shift
    // every lane's outputs right by two columns (its own columns 2, 3 + columns 0, 1 of lane j + 1), which keeps the stores 16-byte aligned;
    // their operand DMA segments are then only 4-byte aligned in HBM (the LDS-DMA accepts that: tools/dma_align_probe.hip)
    const bool shifted = FUSE == 2 && !FUSE_ABL(16) && tile_x > 0;
    const int tx0 = (FUSE == 2 && !FUSE_ABL(16)) ? (tile_x ? 30 * tile_x - 1 : 0) : tile_x * TW;
    const int first_tile = FUSE == 2 ? tile_y * p.seg_tiles : tile_y;
    const int n_tiles = (FUSE == 2 && !FUSE_ABL(32)) ? min(p.seg_tiles, p.tiles_total_y - first_tile) : 1;  // (ablation 32: one tile per workgroup, known at compile time; run with MAUA_FUSE_SEG=1)
    const int m0 = mt_id * U2_BM;
    const size_t plane = (size_t)p.H * p.W;

    if constexpr (!PRE)
        for (int e = tid; e < p.Cin; e += 256) Ss[e] = p.s[(size_t)b0 * p.s_stride + e];
    (void)Ss;

    const char* ximg = reinterpret_cast<const char*>(p.x + (size_t)b0 * p.Cin * plane);
    const size_t plane_bytes = plane * sizeof(float);
    unsigned rel_bytes[P_PER_WAVE];
    (void)ximg, (void)plane_bytes, (void)rel_bytes;
#ifdef MAUA_DEVICE_PASS
    // (FUSE == 2: the exact size of the image — a segment that straddles its last element must not touch memory behind it; the range check is per dword
    // and includes the scalar offset: tools/dma_range_probe.hip)
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ximg), 0, FUSE == 2 ? (int)((size_t)p.Cin * plane_bytes) : 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wq), 0, 0x7fffffff, 0x00020000);
#endif
    auto issue = [&](int chunk, int buf) {
#ifdef MAUA_DEVICE_PASS
        const int wbase = (int)(((size_t)mt_id * p.n_chunks + chunk) * U2_A_FLOATS * sizeof(float));
#pragma unroll
        for (int k = 0; k < A_PER_WAVE; ++k) {  // weight tile: linear copy, 1 KiB per wave instruction, instructions w, w + 4, ...
            const int i = wv + 4 * k;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(lds + buf * U2_A_FLOATS + i * 256),
                                                     16, (i * 256 + lane * 4) * 4, wbase, 0, 0);
        }
        const int xbase = (int)((size_t)chunk * CC * plane_bytes);
#pragma unroll
        for (int k = 0; k < P_PER_WAVE; ++k) {
            const int i = wv + 4 * k;  // (scalar)
            if (i < U2_P_INSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(Ps + buf * U2_PBUF + i * 256),
                                                         16, (int)rel_bytes[k], xbase, 0, 0);
        }
#else
        (void)chunk, (void)buf;
#endif
    };

    // LDS byte addresses (buffer 0) of this lane's operands; the second buffer is a constant distance away.
    // window of block j, channel kq: staged rows 2 w .. 2 w + 2, floats 2 j + 3 .. 2 j + 5 of a row (float 3 = image column
    // tx0 - 1), read as two aligned 8-byte pairs (2 j + 2, 2 j + 3) and (2 j + 4, 2 j + 5)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    // weight row of entry u, channel c of the K step: (u * CC + c) * 32 + 2 * j (m-tile pair interleaved: one 8-byte read feeds
    // both m-tiles)
    constexpr unsigned A_BUF_BYTES = U2_A_FLOATS * 4u, P_BUF_BYTES = U2_PBUF * 4u;
    constexpr int U_BYTES = CC * U2_BM * 4;     // distance between transformed-kernel entries in the weight tile
    constexpr int KG_A_BYTES = 4 * U2_BM * 4;   // ... between the MFMA K groups (4 channels) of one entry
    constexpr int KG_P_BYTES = 4 * U2_PLANE * 4;  // ... between the K groups' channel planes in the patch
    constexpr int ROW_BYTES = U2_PWS * 4;

    // per-channel gain (wscale * demod) of the epilogue: fetched under the first DMA wait into LDS behind the styles (loaded after
    // the main loop, its round trip was exposed in every workgroup)
    float* Eg = lds + 2 * U2_A_FLOATS + 2 * U2_PBUF + ((p.Cin + 3) & ~3);
    for (int i = tid; i < U2_BM; i += 256) {
        float gain = p.wscale;
        if (p.d) gain *= p.d[(size_t)b0 * p.Cout + m0 + i];
        Eg[i] = gain;
        if constexpr (FUSE == 2) {
            Eg[U2_BM + i] = p.post_s ? p.post_s[(size_t)b0 * p.s_stride + m0 + i] : 1.f;  // scale of the stored map
            Eg[2 * U2_BM + i] = p.bias ? p.bias[m0 + i] * 1.41421356237309515f : 0.f;       // bias * sqrt2, once per workgroup (round 6: it was
        }                                                                                   // two global loads per channel pair and TILE)
    }
    float* SV = Eg + 3 * U2_BM;
    if constexpr (FUSE == 2) {
        // h-rows above the segment's first tile: zero — exact at the top of the image (raw rows -3 .. -1 are padding); below a segment
        // boundary the three output rows that would need them are left to up2d_seam_kernel
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(SV + (i * 256 + tid) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#if defined(MAUA_EXPERIMENTS) && defined(MAUA_FUSE_STAGGER)
    // (experiment: the two workgroups of a CU start in lock-step and, with equal work, stay there — K loops together, epilogues together;
    // delaying every other workgroup by part of a tile time puts one's latency-bound epilogue under the other's matrix-bound K loop)
    if (FUSE == 2 && (blockIdx.x & 1))
        for (int i = 0; i < MAUA_FUSE_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    for (int tile = 0; tile < n_tiles; ++tile) {
    const int ty0 = (first_tile + tile) * T_ROWS;   // first position row of the tile
    // (FUSE == 2: everything derived from the lane id is re-derived per tile behind an opaque copy — hoisted out of the tile loop these
    // values would have to survive a K loop that uses 238 registers, i.e. live in scratch)
    int lane_t = lane;
    if constexpr (FUSE == 2) asm volatile("" : "+v"(lane_t));
    const int j_t = lane_t & 15, kq_t = lane_t >> 4;
    // this lane's block: (row, column) = (w, j) on 32-column tiles, (2 w + j / 8, j % 8) on 16-column ones
    const int brow_t = TW == 32 ? wv : 2 * wv + (j_t >> 3), bcol_t = TW == 32 ? j_t : (j_t & 7);
    const unsigned b_addr = lds0 + (unsigned)(2 * U2_A_FLOATS + kq_t * U2_PLANE + (2 * brow_t) * U2_PWS + 2 * bcol_t + 2) * 4u;
    const unsigned a_addr = lds0 + (unsigned)(kq_t * U2_BM + 2 * j_t) * 4u;
    // ---- patch DMA of this lane.  Slot s = 64 i + lane of instruction i is 16-byte slot s of the buffer:
    // channel s / 88, then row (s % 88) / 9 and segment (s % 88) % 9 (slots 81..87 of a channel are padding).  Rows above / below
    // the image, segments left / right of it and the padding get an offset beyond the buffer descriptor's range, for which a
    // raw buffer load returns 0: the DMA itself writes the zero padding.  Wave w issues instructions w, w + 4, ...
#pragma unroll
    for (int g = 0; g < P_PER_WAVE; ++g) {
        const int s = 64 * (wv + 4 * g) + lane_t;
        const int c = s / (U2_PLANE / 4), rem = s % (U2_PLANE / 4);
        const int pr = rem / U2_PSEGS, sg = rem % U2_PSEGS;
        const int yy = ty0 - 1 + pr, xx = tx0 - 4 + 4 * sg;
        const bool ok = c < CC && rem < U2_PROWS * U2_PSEGS && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        rel_bytes[g] = ok ? ((unsigned)c * (unsigned)plane + (unsigned)yy * (unsigned)p.W + (unsigned)xx) * 4u : 0x80000000u;  // (< 2^31: checked by the launcher)
    }

    // ---- accumulators (one 16 x 16 tile = 4 registers each; [.][m-tile]): 25 products x 2 m-tiles = 200 registers
    // (zeroed here: a peeled first K group with C = 0, as in modconv_w2d.hip, spills 20 registers in the CC = 8 instance)
    f32x4 acc_ee[3][3][2], acc_eo[3][2][2], acc_oe[2][3][2], acc_oo[2][2][2];
    {
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b) acc_ee[a][b][m] = z4;
#pragma unroll
                for (int b = 0; b < 2; ++b) acc_eo[a][b][m] = z4, acc_oe[b][a][m] = z4;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc_oo[a][b][m] = z4;
        }
    }
    const int chunk_begin = FUSE == 0 ? split * p.chunks_per_split : 0;
    const int chunk_end = FUSE == 0 ? min(p.n_chunks, chunk_begin + p.chunks_per_split) : p.n_chunks;
    unsigned s_addr = lds0 + (unsigned)(2 * U2_A_FLOATS + 2 * U2_PBUF + chunk_begin * CC + kq_t) * 4u;

    // FUSE == 2, tiles with an odd origin that reach the image's right edge: the 16-byte segment that holds column W - 1 also holds the first
    // floats of the NEXT image row where column W belongs; x[., W] feeds raw columns 2W and 2W + 1 and must be zero (columns beyond feed nothing
    // that is kept).  One float per staged row and channel, overwritten in LDS behind the DMA's arrival.
    const int fix_col = p.W - (tx0 - 4);   // row float of image column W
    const bool fix_edge = FUSE == 2 && shifted && fix_col >= 1 && fix_col <= 35 && (fix_col & 3) != 0;
    auto zero_edge = [&](int buf) {  // (behind the barrier that follows the DMA wait: every wave waits for its OWN instructions only)
        if (fix_edge) {              // (workgroup-uniform)
            if (tid < CC * U2_PROWS) Ps[buf * U2_PBUF + (tid / U2_PROWS) * U2_PLANE + (tid % U2_PROWS) * U2_PWS + fix_col] = 0.f;
            __syncthreads();
        }
    };
    issue(chunk_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    zero_edge(0);
    // The right-edge tiles of the first m-tile export the last input column from their staged patch (row float 35 = image column
    // tx0 + 31 = W - 1) into xcol[b][c][row]: the edge kernel then reads that column with unit stride (gathering it from x costs
    // one 128-byte line per element: 32 of the edge launch's 45 us)
    // (CC * T_ROWS values per chunk: the lanes of the last EXW waves)
    constexpr int EXW = (CC * T_ROWS + 63) / 64;
    const int ex_l = (wv - (4 - EXW)) * 64 + lane;
    const bool export_col = FUSE == 0 && mt_id == 0 && tx0 + TW == p.W && wv >= 4 - EXW && ex_l < CC * T_ROWS;
    const int ex_c = ex_l / T_ROWS, ex_r = ex_l % T_ROWS;
    int cur = 0;
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        if (chunk + 1 < chunk_end) issue(chunk + 1, cur ^ 1);
        if (export_col)
            p.xcol[((size_t)b0 * p.Cin + (size_t)chunk * CC + ex_c) * p.H + ty0 + ex_r] =
                Ps[cur * U2_PBUF + ex_c * U2_PLANE + (ex_r + 1) * U2_PWS + TW + 3];
        static_for<0, CC / 4>([&](auto ks_c) {  // the MFMA K groups of this step: 4 channels each
        constexpr int ks = decltype(ks_c)::value;
        const unsigned ap = a_addr + (cur ? A_BUF_BYTES : 0u) + ks * KG_A_BYTES, pb = b_addr + (cur ? P_BUF_BYTES : 0u) + ks * KG_P_BYTES;
        // ---- operand reads: style, the 3 x 3 window (six 8-byte reads), the first weight row behind them (LDS returns in order)
        float sc = 1.f;
        if constexpr (!PRE) {
            sc = lds_read32(s_addr);
            s_addr += 4 * 4u;
        }
        f32x2 w0l = lds_read64<0>(pb), w0h = lds_read64<8>(pb);
        f32x2 w1l = lds_read64<ROW_BYTES>(pb), w1h = lds_read64<ROW_BYTES + 8>(pb);
        f32x2 w2l = lds_read64<2 * ROW_BYTES>(pb), w2h = lds_read64<2 * ROW_BYTES + 8>(pb);
        f32x2 a2[2];
        a2[0] = lds_read64<0>(ap);
        if constexpr (PRE) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w0l), "+v"(w0h), "+v"(w1l), "+v"(w1h), "+v"(w2l), "+v"(w2h));
        else
        asm volatile("s_waitcnt lgkmcnt(1)"
                     : "+v"(sc), "+v"(w0l), "+v"(w0h), "+v"(w1l), "+v"(w1h), "+v"(w2l), "+v"(w2h));
        // ---- window forms: rows (r0 - r1, r1, r2 - r1, r2), then columns likewise: B[a][b], 9 multiplies (none when the map arrives
        // pre-scaled: PRE) + 14 subtractions
        float bv[4][4];
        {
            float d00 = w0l.y, d01 = w0h.x, d02 = w0h.y, d10 = w1l.y, d11 = w1h.x, d12 = w1h.y, d20 = w2l.y, d21 = w2h.x, d22 = w2h.y;
            if constexpr (!PRE) d00 *= sc, d01 *= sc, d02 *= sc, d10 *= sc, d11 *= sc, d12 *= sc, d20 *= sc, d21 *= sc, d22 *= sc;
            const float r[4][3] = {{d00 - d10, d01 - d11, d02 - d12}, {d10, d11, d12}, {d20 - d10, d21 - d11, d22 - d12}, {d20, d21, d22}};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bv[a][0] = r[a][0] - r[a][1];
                bv[a][1] = r[a][1];
                bv[a][2] = r[a][2] - r[a][1];
                bv[a][3] = r[a][2];
            }
        }
        // ---- MFMA phase: the weight row of the next entry is read one step ahead
        static_for<0, U2_NU>([&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            if constexpr (u + 1 < U2_NU) a2[(u + 1) & 1] = lds_read64<(u + 1) * U_BYTES>(ap);
            lds_wait<(u + 1 < U2_NU) ? 1 : 0>(a2[u & 1]);
            const float a_lo = a2[u & 1].x, a_hi = a2[u & 1].y;
            auto mac = [&](f32x4(&acc)[2], float b) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_lo, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_hi, b, acc[1], 0, 0, 0);
            };
            if constexpr (u < 9) {                       // (even, even): entry [a][b] x window form [a][b]
                mac(acc_ee[u / 3][u % 3], bv[u / 3][u % 3]);
            } else if constexpr (u < 12) {               // (even row, odd column): row form a, raw columns c1 / c2
                mac(acc_eo[u - 9][0], bv[u - 9][1]);
                mac(acc_eo[u - 9][1], bv[u - 9][3]);
            } else if constexpr (u < 15) {               // (odd row, even column): raw rows r1 / r2, column form b
                mac(acc_oe[0][u - 12], bv[1][u - 12]);
                mac(acc_oe[1][u - 12], bv[3][u - 12]);
            } else {                                     // (odd, odd): g11 on the raw 2 x 2 inputs
                mac(acc_oo[0][0], bv[1][1]);
                mac(acc_oo[0][1], bv[1][3]);
                mac(acc_oo[1][0], bv[3][1]);
                mac(acc_oo[1][1], bv[3][3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (FUSE == 2) zero_edge(cur ^ 1);  // (the chunk that has just landed)
        cur ^= 1;
    }

    if constexpr (FUSE != 0) {
        // ---- fused epilogue: raw 4 x 4 patches -> separable 4-tap blur -> noise / bias / leaky ReLU -> [B, Cout, 2H, 2W].
        // out[Y][X] = sum_ab K[a][b] raw[Y - 1 + a][X - 1 + b] (upfirdn2d with pad (1, 1), reference models/stylegan2.py:229-238).
        // This lane: raw rows R0 .. R0 + 3 (R0 = 2 ty0 + 4 wv), raw columns C0 .. C0 + 3 (C0 = 2 tx0 + 4 j), channels 16 m + 4 kq + v.
        //   horizontal pass: the lane's 4 columns need raw columns C0 - 1 .. C0 + 5: column 3 of lane j - 1 and columns 0, 1 of lane j + 1
        //     (DPP row shifts inside the 16-lane row of one K lane group; the row ends read 0: lane 0 is exact only at the image's left
        //     edge, lane 15 never).  FUSE == 2 keeps 60 columns per tile: tile column 0 its lanes 0 .. 14 as they are; the others start two raw
        //     columns early and every lane emits its columns 2, 3 and columns 0, 1 of lane j + 1 (four fetches from the right neighbour, none
        //     from the left), lanes 0 .. 14 again — lane 0's left neighbour and lane 15's right one are never needed;
        //   vertical pass: wave w emits output rows R0 - 2 .. R0 + 1 from the h-rows R0 - 3 .. R0 + 3: its own four and rows 1 .. 3 of the
        //     wave above, through LDS (in the operand buffers the K loop has released); wave 0 takes them from the PREVIOUS tile of its
        //     segment (SV, written by wave 3), so a workgroup walking down a segment never recomputes or re-reads a halo row.
        // All arithmetic on channel PAIRS (registers 2 vp, 2 vp + 1 of an accumulator tile): v_pk_* instructions.
        // The kernel arguments the epilogue needs are RE-READ from the kernarg segment per tile (scalar loads through an opaque pointer):
        // taken from `p` they are loop invariants that the compiler keeps in ~50 SGPRs across the K loop, i.e. spills to vector lanes.
        typedef const __attribute__((address_space(4))) Up2dArgs* KernargPtr;
        KernargPtr pk = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        if constexpr (FUSE == 2) asm volatile("" : "+s"(pk));
        // (per-tile copies of the lane's coordinates behind an opaque barrier: without it the compiler hoists the epilogue's address arithmetic
        // out of the tile loop and carries ~90 registers across the K loop — as spills)
        int lane_e = lane_t;
        asm volatile("" : "+v"(lane_e));
        const int j_e = lane_e & 15, kq_e = lane_e >> 4;
        float kx[4], ky[4];
        u2_blur_taps(pk->k4, kx, ky);
        const float act_gain = 1.41421356237309515f;
        const int OHb = 2 * pk->H, OWb = 2 * pk->W;
        const int Y0 = 2 * ty0 + 4 * wv - 2, X0 = 2 * tx0 + 4 * j_e + (shifted ? 2 : 0);  // (FUSE == 2: 60 tile_x + 4 j either way)
        const float* noise_base = pk->noise;
        int64_t noise_bstride = pk->noise_batch_stride;
        if (pk->src) {  // (uniform scalar loads)
            noise_bstride = pk->src->noise_stride[pk->noise_slot];
            noise_base = pk->src->noise[pk->noise_slot];
            if (noise_base) noise_base += (int64_t)pk->src->frame0 * noise_bstride;
        }
        const float nw = noise_base ? pk->noise_w[0] * act_gain : 0.f;
        // which of this lane's outputs are kept: FUSE == 2: lanes 1 .. 14 (+ lane 0 of the first tile column), inside the map
        const bool x_keep = ((FUSE == 2 && !FUSE_ABL(16)) ? j_e <= 14 : true) && X0 < OWb;  // (lane 15 has no right neighbour)
        const bool seam_above = FUSE == 2 && tile == 0 && tile_y > 0;  // the three rows above this tile's first kept row go to the seam kernel
        const bool seam_below = FUSE == 2 && tile == n_tiles - 1 && first_tile + tile + 1 < pk->tiles_total_y;
        const int Yx = Y0 < 0 ? 0 : (Y0 + 3 < OHb ? Y0 : (OHb >= 4 ? OHb - 4 : 0));  // (clamped row base of the noise loads: always inside the map)
        const float* nzp = noise_base ? noise_base + (size_t)b0 * noise_bstride + (size_t)Yx * OWb + (X0 < OWb ? X0 : 0) : pk->k4;
        const int nzs = noise_base ? OWb : 0;
        float* xch = lds;  // exchange region [parity 2][wave 4][row 3][half 2][lane 64][4 floats] = 48 KB over the weight / patch buffers (a lane's 8
                           // floats as two 16-byte pieces 1 KB apart: consecutive lanes hit consecutive banks — [lane][8] measured 22.6 % bank conflicts)
        float* ybimg = pk->yb + ((size_t)b0 * pk->Cout + m0) * ((size_t)OHb * OWb);
        // neighbour lanes of the 16-lane row (one K lane group): DPP row shifts, 0 beyond the row ends.  (Scalar helpers on purpose:
        // __builtin_bit_cast applied to the .y ELEMENT of an ext-vector gave poison in this compiler and the second lane's move vanished.)
        auto dpp_from_left = [](float a) {
            return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x111, 0xf, 0xf, true));  // row_shr:1: lane j - 1
        };
        auto dpp_from_right = [](float a) {
            return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x101, 0xf, 0xf, true));  // row_shl:1: lane j + 1
        };
        auto shr1 = [&](f32x2 v) {
            const float a = v.x, b = v.y;
            return f32x2{dpp_from_left(a), dpp_from_left(b)};
        };
        auto shl1 = [&](f32x2 v) {
            const float a = v.x, b = v.y;
            return f32x2{dpp_from_right(a), dpp_from_right(b)};
        };
        // phase sums first, for all eight channels: 200 accumulator registers become 128 raw values before the blur's temporaries exist
        f32x2 Rall[4][4][4];  // [channel pair][row][col]
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = it >> 1, vp = it & 1;
            auto pr = [&](const f32x4& a) { return vp ? f32x2{a[2], a[3]} : f32x2{a[0], a[1]}; };
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x2 ee00 = pr(acc_ee[i][0][m]), ee01 = pr(acc_ee[i][1][m]), ee02 = pr(acc_ee[i][2][m]);
                const f32x2 ee10 = pr(acc_ee[i + 1][0][m]), ee11 = pr(acc_ee[i + 1][1][m]), ee12 = pr(acc_ee[i + 1][2][m]);
                const f32x2 mid = ee01 + ee11;
                Rall[it][2 * i][0] = (ee00 + ee10) + mid;
                Rall[it][2 * i][2] = mid + (ee02 + ee12);
                Rall[it][2 * i][1] = pr(acc_eo[i][0][m]) + pr(acc_eo[i + 1][0][m]);
                Rall[it][2 * i][3] = pr(acc_eo[i][1][m]) + pr(acc_eo[i + 1][1][m]);
                const f32x2 oe1 = pr(acc_oe[i][1][m]);
                Rall[it][2 * i + 1][0] = pr(acc_oe[i][0][m]) + oe1;
                Rall[it][2 * i + 1][2] = oe1 + pr(acc_oe[i][2][m]);
                Rall[it][2 * i + 1][1] = pr(acc_oo[i][0][m]);
                Rall[it][2 * i + 1][3] = pr(acc_oo[i][1][m]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < (FUSE_ABL(64) ? 1 : 4); ++it) {  // (ablation 64: one channel pair of four)
            const int m = it >> 1, vp = it & 1;
            const f32x2(&R)[4][4] = Rall[it];
            // horizontal pass
            f32x2 Hh[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 right0 = shl1(R[r][0]), right1 = shl1(R[r][1]);
                if (!shifted) {  // (uniform) outputs = the lane's own four columns: raw columns C0 - 1 .. C0 + 5
                    const f32x2 left = shr1(R[r][3]);
                    const f32x2 e[7] = {left, R[r][0], R[r][1], R[r][2], R[r][3], right0, right1};
#pragma unroll
                    for (int c = 0; c < 4; ++c) Hh[r][c] = ((e[c] * kx[0] + e[c + 1] * kx[1]) + e[c + 2] * kx[2]) + e[c + 3] * kx[3];
                } else {         // outputs = columns C0 + 2 .. C0 + 5: raw columns C0 + 1 .. C0 + 7
                    const f32x2 right2 = shl1(R[r][2]), right3 = shl1(R[r][3]);
                    const f32x2 e[7] = {R[r][1], R[r][2], R[r][3], right0, right1, right2, right3};
#pragma unroll
                    for (int c = 0; c < 4; ++c) Hh[r][c] = ((e[c] * kx[0] + e[c + 1] * kx[1]) + e[c + 2] * kx[2]) + e[c + 3] * kx[3];
                }
            }
            const int ol = 16 * m + 4 * kq_e + 2 * vp;   // first channel of the pair inside the m-tile pair
            // rows 1 .. 3 to the wave below
            float* mine = xch + (size_t)(((it & 1) * 4 + wv) * 3) * 64 * 8 + lane_e * 4;
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                *reinterpret_cast<f32x4*>(mine + (r - 1) * 64 * 8) = f32x4{Hh[r][0].x, Hh[r][0].y, Hh[r][1].x, Hh[r][1].y};
                *reinterpret_cast<f32x4*>(mine + (r - 1) * 64 * 8 + 256) = f32x4{Hh[r][2].x, Hh[r][2].y, Hh[r][3].x, Hh[r][3].y};
            }
            if constexpr (FUSE == 2) {
                // the h-rows either side of a segment boundary go to hbuf [B][Cout][n_seg - 1][6][2W] (rows 0 .. 2: above, 3 .. 5: below)
                if (!FUSE_ABL(2) && ((seam_above && wv == 0) || (seam_below && wv == 3))) {
                    if (x_keep) {
                        const int bnd = wv == 0 ? tile_y - 1 : tile_y;
                        float* hb = pk->hbuf + ((((size_t)b0 * pk->Cout + m0 + ol) * (pk->tiles_y - 1) + bnd) * 6 + (wv == 0 ? 3 : 0)) * (size_t)OWb + X0;
                        const size_t chs = (size_t)(pk->tiles_y - 1) * 6 * OWb;  // floats between channels
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const f32x2(&row)[4] = Hh[wv == 0 ? r : r + 1];
                            *reinterpret_cast<f32x4*>(hb + (size_t)r * OWb) = f32x4{row[0].x, row[1].x, row[2].x, row[3].x};
                            *reinterpret_cast<f32x4*>(hb + (size_t)r * OWb + chs) = f32x4{row[0].y, row[1].y, row[2].y, row[3].y};
                        }
                    }
                }
            }
            __syncthreads();
            f32x2 S[7][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) S[r][c] = f32x2{0.f, 0.f};  // (FUSE == 1: the tile above, zero in the experiment)
            if (wv != 0 || FUSE == 2) {  // (uniform)
                const float* above = wv != 0 ? xch + (size_t)(((it & 1) * 4 + (wv - 1)) * 3) * 64 * 8 + lane_e * 4
                                             : SV + (size_t)(it * 3) * 64 * 8 + lane_e * 4;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(above + r * 64 * 8), hi = *reinterpret_cast<const f32x4*>(above + r * 64 * 8 + 256);
                    S[r][0] = f32x2{lo[0], lo[1]}, S[r][1] = f32x2{lo[2], lo[3]}, S[r][2] = f32x2{hi[0], hi[1]}, S[r][3] = f32x2{hi[2], hi[3]};
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) S[3 + r][c] = Hh[r][c];
            // vertical pass + tail + stores
            const f32x2 gain2 = f32x2{Eg[ol] * act_gain, Eg[ol + 1] * act_gain};
            const f32x2 bias2 = FUSE == 2 ? f32x2{Eg[2 * U2_BM + ol], Eg[2 * U2_BM + ol + 1]}
                                          : (pk->bias ? f32x2{pk->bias[m0 + ol] * act_gain, pk->bias[m0 + ol + 1] * act_gain} : f32x2{0.f, 0.f});
            const f32x2 post2 = FUSE == 2 ? f32x2{Eg[U2_BM + ol], Eg[U2_BM + ol + 1]} : f32x2{1.f, 1.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x2 val[4];
                const f32x4 nzq = *reinterpret_cast<const f32x4*>(nzp + (Y0 + q - Yx < 0 ? 0 : (Y0 + q - Yx > 3 ? 3 : Y0 + q - Yx)) * nzs) * nw;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x2 bl = ((S[q][c] * ky[0] + S[q + 1][c] * ky[1]) + S[q + 2][c] * ky[2]) + S[q + 3][c] * ky[3];
                    const f32x2 tt = bl * gain2 + (f32x2{nzq[c], nzq[c]} + bias2);
                    val[c] = __builtin_elementwise_max(tt, tt * 0.2f) * post2;
                }
                const int Y = Y0 + q;
                const bool row_keep = Y >= 0 && Y < OHb && !(seam_above && wv == 0 && q < 3);  // (uniform)
                if (row_keep && x_keep && !FUSE_ABL(4)) {
                    float* dst = ybimg + ((size_t)ol * OHb + Y) * OWb + X0;
                    *reinterpret_cast<f32x4*>(dst) = f32x4{val[0].x, val[1].x, val[2].x, val[3].x};
                    *reinterpret_cast<f32x4*>(dst + (size_t)OHb * OWb) = f32x4{val[0].y, val[1].y, val[2].y, val[3].y};
                }
            }
            if constexpr (FUSE == 2 && !FUSE_ABL(1)) {
                // wave 3's rows 1 .. 3 are the next tile's rows above: written once every wave has read this tile's SV[it]
                __syncthreads();
                if (wv == 3) {
                    float* sv = SV + (size_t)(it * 3) * 64 * 8 + lane_e * 4;
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        *reinterpret_cast<f32x4*>(sv + (r - 1) * 64 * 8) = f32x4{Hh[r][0].x, Hh[r][0].y, Hh[r][1].x, Hh[r][1].y};
                        *reinterpret_cast<f32x4*>(sv + (r - 1) * 64 * 8 + 256) = f32x4{Hh[r][2].x, Hh[r][2].y, Hh[r][3].x, Hh[r][3].y};
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FUSE == 2) __syncthreads();  // (the next tile's operand DMA overwrites the exchange region)
    } else {
        // ---- epilogue: phase sums, per-channel gain, 16-byte stores of the 4 x 4 output patch
        const int OW = 2 * p.W + 1;
        const size_t plane_out = (size_t)(2 * p.H + 1) * OW;
        float* yimg = p.y + (size_t)split * p.slab + ((size_t)b0 * p.Cout + m0) * plane_out;
        const int brow = TW == 32 ? wv : 2 * wv + (j >> 3), bcol = TW == 32 ? j : (j & 7);
        const unsigned pix_off = (unsigned)(2 * (ty0 + 2 * brow)) * (unsigned)OW + (unsigned)(2 * (tx0 + 2 * bcol));
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ol = m * 16 + 4 * kq + v;  // row of the 16 x 16 result tile held in register v
                const float gain = Eg[ol];
                float* dst = yimg + (size_t)ol * plane_out + pix_off;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // even output row of position row i: (ee, eo, ee, eo);  odd one: (oe, oo, oe, oo)
                    const float e0 = (acc_ee[i][0][m][v] + acc_ee[i][1][m][v]) + (acc_ee[i + 1][0][m][v] + acc_ee[i + 1][1][m][v]);
                    const float e1 = (acc_ee[i][1][m][v] + acc_ee[i][2][m][v]) + (acc_ee[i + 1][1][m][v] + acc_ee[i + 1][2][m][v]);
                    const float o0 = acc_eo[i][0][m][v] + acc_eo[i + 1][0][m][v];
                    const float o1 = acc_eo[i][1][m][v] + acc_eo[i + 1][1][m][v];
                    *reinterpret_cast<f32x4u*>(dst + (size_t)(2 * i) * OW) = f32x4{e0 * gain, o0 * gain, e1 * gain, o1 * gain};
                    const float f0 = acc_oe[i][0][m][v] + acc_oe[i][1][m][v];
                    const float f1 = acc_oe[i][1][m][v] + acc_oe[i][2][m][v];
                    *reinterpret_cast<f32x4u*>(dst + (size_t)(2 * i + 1) * OW) =
                        f32x4{f0 * gain, acc_oo[i][0][m][v] * gain, f1 * gain, acc_oo[i][1][m][v] * gain};
                }
            }
        }

    }
    }  // tiles of the segment
}

// The three output rows either side of every segment boundary of the exact fused kernel: vertical pass + tail over the six h-rows
// modconv_up2d_kernel<., 2> left in hbuf [B][Cout][n_bnd][6][OW].  Boundary g lies above raw row R = 16 seg_tiles (g + 1): outputs R - 2, R - 1, R.
__global__ __launch_bounds__(256) void up2d_seam_kernel(Up2dArgs p, int n_bnd) {
    const int OHb = 2 * p.H, OWb = 2 * p.W;
    const int xq = OWb / 4;  // 16-byte column groups per row
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.Cout * n_bnd * xq;
    if (t >= total) return;
    const int x4 = (int)(t % xq);
    t /= xq;
    const int g = (int)(t % n_bnd);
    t /= n_bnd;
    const int c = (int)(t % p.Cout), b = (int)(t / p.Cout);
    float kx[4], ky[4];
    u2_blur_taps(p.k4, kx, ky);
    (void)kx;
    const float act_gain = 1.41421356237309515f;
    const float* noise_base = p.noise;
    int64_t noise_bstride = p.noise_batch_stride;
    if (p.src) {
        noise_bstride = p.src->noise_stride[p.noise_slot];
        noise_base = p.src->noise[p.noise_slot];
        if (noise_base) noise_base += (int64_t)p.src->frame0 * noise_bstride;
    }
    const float nw = noise_base ? p.noise_w[0] * act_gain : 0.f;
    float gain = p.wscale * act_gain;
    if (p.d) gain *= p.d[(size_t)b * p.Cout + c];
    const float bias = p.bias ? p.bias[c] * act_gain : 0.f;
    const float post = p.post_s ? p.post_s[(size_t)b * p.s_stride + c] : 1.f;
    const float* hb = p.hbuf + (((size_t)b * p.Cout + c) * n_bnd + g) * 6 * (size_t)OWb + 4 * x4;
    f32x4 h[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) h[r] = *reinterpret_cast<const f32x4*>(hb + (size_t)r * OWb);
    const int R = 16 * p.seg_tiles * (g + 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int Y = R - 2 + k;
        if (Y < 0 || Y >= OHb) continue;
        f32x4 nz = f32x4{0.f, 0.f, 0.f, 0.f};
        if (noise_base) nz = *reinterpret_cast<const f32x4*>(noise_base + (size_t)b * noise_bstride + (size_t)Y * OWb + 4 * x4) * nw;
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bl = ((h[k][e] * ky[0] + h[k + 1][e] * ky[1]) + h[k + 2][e] * ky[2]) + h[k + 3][e] * ky[3];
            const float tt = bl * gain + (nz[e] + bias);
            out[e] = fmaxf(tt, tt * 0.2f) * post;
        }
        *reinterpret_cast<f32x4*>(p.yb + (((size_t)b * p.Cout + c) * OHb + Y) * (size_t)OWb + 4 * x4) = out;
    }
}

// Output row 2H (line 0) and column 2W (line 1) of the transposed convolution: positions whose own input is the zero padding.
// Along the line, with v[n] = the last input row / column and (t0, t1, t2) = the kernel's last row / column:
//     y[2n] = t0 v[n] + t2 v[n-1],   y[2n+1] = t1 v[n]      (v[-1] = v[N] = 0; the corner y[2N] belongs to line 0)
// One workgroup per (image, line, 16 output channels, 16 line positions): direct v_mfma_f32_16x16x4_f32 over the input channels,
// operands straight from global memory (the edge tap matrices [5][Cin][Cout] behind the packed weight; the line of x with its
// unit stride; the last column comes from the main kernel's export).  The work is ~1/H of the layer's multiply-adds but a chain of dependent global loads: the four waves of the
// workgroup split the input channels (a single wave per tile ran 32 serial round trips at one wave per SIMD: 49 us per launch,
// 4 % of the frame) and keep eight K steps of loads in flight; the partial tiles meet in LDS.
__global__ __launch_bounds__(256) void up2d_edge_kernel(Up2dArgs p, const float* __restrict__ taps, int n_tiles0, int n_tiles1) {
    __shared__ float part[4][2][64][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    int t = blockIdx.x;
    const int m_tiles = p.Cout / 16;
    const int per_image = m_tiles * (n_tiles0 + n_tiles1);
    const int b0 = t / per_image;
    t -= b0 * per_image;
    const int line = t >= m_tiles * n_tiles0 ? 1 : 0;
    if (line) t -= m_tiles * n_tiles0;
    const int nt = line ? n_tiles1 : n_tiles0;
    const int mt = t / nt, n0 = (t - mt * nt) * 16;
    const int o0 = mt * 16;
    const int N = line ? p.H : p.W;               // inputs along the line
    const int n = n0 + i16;                       // this lane's line position (B operand column / result column)
    const size_t plane = (size_t)p.H * p.W;
    // v[n]: line 0 = x[c][H-1][n] (a row of x), line 1 = x[c][n][W-1] = xcol[c][n] (exported by the main kernel): both unit stride
    const float* xb = line ? p.xcol + (size_t)b0 * p.Cin * p.H : p.x + (size_t)b0 * p.Cin * plane + (size_t)(p.H - 1) * p.W;
    const size_t cstride = line ? (size_t)p.H : plane;  // floats between channels
    const bool ok_n = n < N, ok_m = n >= 1 && n - 1 < N;
    const float* pv = xb + (ok_n ? n : 0);
    const float* pm = xb + (ok_m ? n - 1 : 0);
    // taps: line 0 -> (g20, g21, g22) = matrices 0, 1, 2;  line 1 -> (g02, g12, g22) = matrices 3, 4, 2
    const size_t tsz = (size_t)p.Cin * p.Cout;
    const float* t0 = taps + (line ? 3 : 0) * tsz + o0 + i16;
    const float* t1 = taps + (line ? 4 : 1) * tsz + o0 + i16;
    const float* t2 = taps + 2 * tsz + o0 + i16;
    const float* sp = p.s ? p.s + (size_t)b0 * p.s_stride : nullptr;  // (NULL: the map arrives pre-scaled)
    f32x4 even = f32x4{0.f, 0.f, 0.f, 0.f}, odd = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int UNR = 8;
    const int c_per_wave = (p.Cin / 4 + 3) / 4 * 4;  // channels per wave, a multiple of the MFMA K
    const int c_begin = wv * c_per_wave, c_end = min(p.Cin, c_begin + c_per_wave);
    for (int c0 = c_begin; c0 < c_end; c0 += 4 * UNR) {
        float a0[UNR], a1[UNR], a2v[UNR], bn[UNR], bm[UNR], sc[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {  // all loads of eight K steps first
            const int c = c0 + 4 * q + kq;
            const bool okc = c < c_end;
            const int cc = okc ? c : c_begin;
            sc[q] = okc ? (sp ? sp[cc] : 1.f) : 0.f;
            a0[q] = t0[(size_t)cc * p.Cout];
            a1[q] = t1[(size_t)cc * p.Cout];
            a2v[q] = t2[(size_t)cc * p.Cout];
            bn[q] = ok_n ? pv[(size_t)cc * cstride] : 0.f;
            bm[q] = ok_m ? pm[(size_t)cc * cstride] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const float vn = bn[q] * sc[q], vm = bm[q] * sc[q];  // (sc = 0 masks the channels beyond this wave's range)
            even = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], vn, even, 0, 0, 0);
            even = __builtin_amdgcn_mfma_f32_16x16x4f32(a2v[q], vm, even, 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], vn, odd, 0, 0, 0);
        }
    }
    // the four waves' partial tiles meet in LDS; wave 0 adds them up and stores
#pragma unroll
    for (int v = 0; v < 4; ++v) part[wv][0][lane][v] = even[v], part[wv][1][lane][v] = odd[v];
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int w = 1; w < 4; ++w)
#pragma unroll
        for (int v = 0; v < 4; ++v) even[v] += part[w][0][lane][v], odd[v] += part[w][1][lane][v];
    // result tile: column = line position n (lane & 15), row = output channel o0 + 4 kq + v
    const int OW = 2 * p.W + 1, OH = 2 * p.H + 1;
    const size_t plane_out = (size_t)OH * OW;
    if (n > N || (line && n >= N)) return;  // line 1 leaves the corner (n = N) to line 0
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int o = o0 + 4 * kq + v;
        float gain = p.wscale;
        if (p.d) gain *= p.d[(size_t)b0 * p.Cout + o];
        float* yp = p.y + ((size_t)b0 * p.Cout + o) * plane_out;
        if (line == 0) {
            float* dst = yp + (size_t)(2 * p.H) * OW + 2 * n;
            dst[0] = even[v] * gain;
            if (n < N) dst[1] = odd[v] * gain;
        } else {
            yp[(size_t)(2 * n) * OW + 2 * p.W] = even[v] * gain;
            yp[(size_t)(2 * n + 1) * OW + 2 * p.W] = odd[v] * gain;
        }
    }
}

// wq: [m_tile][chunk][u 16][channel of the K step: CC][32 columns: 2 * (o % 16) + (o % 32) / 16], then the edge tap matrices [5][cin][cout]:
// g20, g21, g22, g02, g12 (kernel's last row, then the other two entries of its last column)
__global__ __launch_bounds__(256) void pack_weight_up2d_kernel(const float* __restrict__ w, float* __restrict__ wq, int cout, int cin) {
    const int U2_CC = u2_cc(cin), U2_A_FLOATS = u2_a_floats(U2_CC);
    const int n_chunks = cin / U2_CC;
    const int64_t total = (int64_t)cout * cin;
    float* edge = wq + (size_t)U2_NU * cin * cout;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int o = (int)(idx % cout), i = (int)(idx / cout);
        const float* g = w + ((size_t)o * cin + i) * 9;
        float k[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) k[a][b] = g[a * 3 + b];
        // one axis transforms as (tap 2, tap 0 + tap 2, tap 0)
        float v[3][3];  // vertical transform of every kernel column
        for (int b = 0; b < 3; ++b) v[0][b] = k[2][b], v[1][b] = k[0][b] + k[2][b], v[2][b] = k[0][b];
        float u[U2_NU];
        for (int a = 0; a < 3; ++a) {
            u[3 * a + 0] = v[a][2], u[3 * a + 1] = v[a][0] + v[a][2], u[3 * a + 2] = v[a][0];  // (even, even)
            u[9 + a] = v[a][1];                                                                 // (even row, odd column): kernel column 1
        }
        u[12] = k[1][2], u[13] = k[1][0] + k[1][2], u[14] = k[1][0];                            // (odd row, even column): kernel row 1
        u[15] = k[1][1];
        const int mtile = o / U2_BM, ol = o % U2_BM;
        const int chunk = i / U2_CC, kq = i % U2_CC;
        float* dst = wq + ((size_t)mtile * n_chunks + chunk) * U2_A_FLOATS + kq * U2_BM + 2 * (ol % 16) + ol / 16;
        for (int e = 0; e < U2_NU; ++e) dst[e * (U2_CC * U2_BM)] = u[e];
        const size_t tsz = (size_t)cin * cout;
        float* ed = edge + (size_t)i * cout + o;
        ed[0] = k[2][0], ed[tsz] = k[2][1], ed[2 * tsz] = k[2][2], ed[3 * tsz] = k[0][2], ed[4 * tsz] = k[1][2];
    }
}

char g_up2d_instance[64] = "";

}  // namespace

// Layer shapes the kernel accepts: whole tiles of 8 x 32 positions, 32-channel m-tiles, 4-channel K steps.
extern "C" int maua_modconv_up2d_ok(int cin, int cout, int h, int w) {
    return cin > 0 && cout > 0 && cin % 4 == 0 && cout % U2_BM == 0 && h >= 8 && h % 8 == 0 && w >= 32 && w % 32 == 0;
}

extern "C" int64_t maua_pack_weight_up2d_floats(int cout, int cin) { return (int64_t)(U2_NU + 5) * cin * cout; }

extern "C" int maua_pack_weight_up2d_f32(const float* w, float* wq, int cout, int cin, void* stream) {
    if (!w || !wq || cout <= 0 || cin <= 0 || cin % 4 || cout % U2_BM) return MAUA_EINVAL;
    const int64_t blocks = ceil_div64((int64_t)cout * cin, 256);
    hipLaunchKernelGGL(pack_weight_up2d_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, w, wq,
                       cout, cin);
    MAUA_LAUNCH_CHECK();
    return 0;
}

const char* maua_up2d_last_instance() { return g_up2d_instance; }

// The edge lines alone (output row 2H, column 2W of the transposed convolution) from the five fp32 edge tap matrices [5][cin][cout] and the
// exported last input column xcol [B, cin, H]: used by the split-bf16 side path (modconv_sbf16.hip), whose phase kernels cover p < H, q < W.
int maua_up2d_edge_launch(const float* x, const float* edge_taps, const float* s, int s_stride, const float* d, float* y, const float* xcol,
                          int batch, int cin, int cout, int h, int w, float wscale, void* stream) {
    if (!x || !edge_taps || !y || !xcol || cout % 16) return MAUA_EINVAL;
    Up2dArgs a{};
    a.x = x, a.wq = nullptr, a.s = s, a.d = d, a.y = y, a.xcol = const_cast<float*>(xcol);
    a.B = batch, a.Cin = cin, a.Cout = cout, a.H = h, a.W = w, a.s_stride = s_stride, a.wscale = wscale;
    const int nt0 = ceil_div(w + 1, 16), nt1 = ceil_div(h, 16);
    const int64_t eblocks = (int64_t)batch * (cout / 16) * (nt0 + nt1);
    hipLaunchKernelGGL(up2d_edge_kernel, dim3((unsigned)eblocks), dim3(256), 0, (hipStream_t)stream, a, edge_taps, nt0, nt1);
    MAUA_LAUNCH_CHECK();
    return 0;
}

// ---- the whole up-sampling StyledConv in one pass over the accumulators (FUSE == 2) -------------------------------------------------------
namespace {
struct FusePlan {
    int tiles_x, tiles_total_y, seg_tiles, n_seg;
};
// x: a tile keeps 60 of its 64 raw columns, ceil(2W / 60) tile columns (see the kernel);
// y: H / 8 tiles + one for raw row 2H, walked in segments of seg_tiles by one workgroup each.  The segment length decides how the launch ENDS:
// every XCD takes a contiguous run of the workgroup list (xcd_remap), its CUs hold two workgroups each, a workgroup alone on its CU runs ~1.6 x
// faster, and a freed slot takes the next workgroup of the run.  fuse_end_time plays that out (a workgroup costs its tiles + 0.2 of a tile for its
// prologue); fuse_plan keeps the length that ends first.  Fitted on 27 measured (layer, length) points of the 1024^2 generator's three largest
// up-sampling layers: 2.7 % rms, the measured optimum or its runner-up (within 1 %) chosen in each (profiles/r05_fused_upconv_blur.md).
double fuse_end_time(const std::vector<float>& cost, int cus_per_xcd) {
    constexpr int NX = 8;
    constexpr float ALONE = 1.6f;
    const int n = (int)cost.size(), q = n / NX, r = n % NX;
    double worst = 0.0;
    std::vector<float> left(2 * cus_per_xcd);
    for (int x = 0, base = 0; x < NX; ++x) {
        const int count = q + (x < r);
        std::fill(left.begin(), left.end(), 0.f);
        int next = base;
        double now = 0.0;
        for (;;) {
            // fill the free slots in order, empty CUs first
            for (int pass = 0; pass < 2 && next < base + count; ++pass)
                for (int c = 0; c < cus_per_xcd && next < base + count; ++c) {
                    const bool a = left[2 * c] > 0.f, b = left[2 * c + 1] > 0.f;
                    if (pass == 0 ? (!a && !b) : (a != b)) left[2 * c + (a ? 1 : 0)] = cost[next++];
                }
            float dt = -1.f;
            for (int c = 0; c < cus_per_xcd; ++c) {
                const float a = left[2 * c], b = left[2 * c + 1];
                const float d = (a > 0.f && b > 0.f) ? std::min(a, b) : std::max(a, b) / ALONE;
                if (d > 0.f && (dt < 0.f || d < dt)) dt = d;
            }
            if (dt < 0.f) break;
            now += dt;
            for (int c = 0; c < cus_per_xcd; ++c) {
                float& a = left[2 * c];
                float& b = left[2 * c + 1];
                const float step = (a > 0.f && b > 0.f) ? dt : dt * ALONE;
                a = a - step > 1e-4f ? a - step : 0.f;
                b = b - step > 1e-4f ? b - step : 0.f;
            }
        }
        worst = std::max(worst, now);
        base += count;
    }
    return worst;
}
FusePlan fuse_plan(int batch, int cout, int h, int w) {
    static std::mutex lock;
    static std::map<std::array<int, 5>, FusePlan> cache;
    static int cus_of[64] = {};  // per device: the plan models THIS device's CUs (a second, different GPU must not inherit the first one's plan)
    std::lock_guard<std::mutex> guard(lock);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cus_of[dev] && (hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus_of[dev] < 8)) cus_of[dev] = 256;
    cus = cus_of[dev];
    const std::array<int, 5> key{batch, cout, h, w, cus};
    FusePlan f{};
    f.tiles_x = (2 * w + 59) / 60;  // a tile keeps 60 of its 64 raw columns
    if (FUSE_ABL(16)) f.tiles_x = w / 32;
    f.tiles_total_y = h / 8 + 1;
#ifdef MAUA_EXPERIMENTS
    if (const char* force = getenv("MAUA_FUSE_SEG")) {  // (A/B: a fixed segment length)
        f.seg_tiles = atoi(force) > 0 ? atoi(force) : 1;
        f.n_seg = (f.tiles_total_y + f.seg_tiles - 1) / f.seg_tiles;
        return f;
    }
#endif
    if (auto it = cache.find(key); it != cache.end()) return it->second;
    const int per_seg = f.tiles_x * (cout / U2_BM);   // workgroups of one (image, segment): consecutive in the list
    double best = 1e30;
    std::vector<float> cost;
    for (int sg = 2; sg <= 16 && sg <= f.tiles_total_y; ++sg) {
        const int n_seg = (f.tiles_total_y + sg - 1) / sg;
        cost.clear();
        for (int b = 0; b < batch; ++b)
            for (int g = 0; g < n_seg; ++g) cost.insert(cost.end(), per_seg, std::min(sg, f.tiles_total_y - g * sg) + 0.2f);
        const double end = fuse_end_time(cost, cus / 8) + 1e-3 * n_seg;  // (ties: fewer seams)
        if (end < best) best = end, f.seg_tiles = sg, f.n_seg = n_seg;
    }
    if (!f.seg_tiles) f.seg_tiles = f.tiles_total_y, f.n_seg = 1;
    cache[key] = f;
    return f;
}
}  // namespace

extern "C" int maua_upconv_blur_ok(int cin, int cout, int h, int w) {
    return maua_modconv_up2d_ok(cin, cout, h, w) && u2_cc(cin) == 8 && cin <= 256;  // (LDS: operand buffers + styles + the 24 KB of saved rows)
}

extern "C" int64_t maua_upconv_blur_ws_floats(int batch, int cin, int cout, int h, int w) {
    if (!maua_upconv_blur_ok(cin, cout, h, w)) return 0;
    const FusePlan f = fuse_plan(batch, cout, h, w);
    return (int64_t)batch * cout * (f.n_seg - 1) * 6 * 2 * w + 4;
}

extern "C" int maua_upconv_blur_f32(const float* x, const float* wq, const float* s, int s_stride, const float* d, float* y, float* ws,
                                    const float* k4, const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                    const float* bias, const maua_frame_source_t* src, int noise_slot, int batch, int cin, int cout, int h,
                                    int w, float wscale, const float* post_s, void* stream) {
    if (!x || !wq || !y || !k4 || batch <= 0) return MAUA_EINVAL;
    if (!maua_upconv_blur_ok(cin, cout, h, w)) return MAUA_ENOSYS;
    if ((noise || src) && !noise_w) return MAUA_EINVAL;
    if (src && (noise_slot < 0 || noise_slot >= MAUA_MAX_NOISE_SLOTS)) return MAUA_EINVAL;
    if ((int64_t)cin * h * w * 4 > 0x7fffffffLL || (int64_t)U2_NU * cin * cout * 4 > 0x7fffffffLL) return MAUA_EINVAL;  // descriptor ranges
    const FusePlan f = fuse_plan(batch, cout, h, w);
    if (f.n_seg > 1 && !ws) return MAUA_EINVAL;
    constexpr int cc = 8;
    Up2dArgs a{};
    a.x = x, a.wq = wq, a.s = s, a.d = d, a.y = nullptr, a.xcol = nullptr;
    a.B = batch, a.Cin = cin, a.Cout = cout, a.H = h, a.W = w, a.s_stride = s_stride, a.wscale = wscale;
    a.tiles_x = f.tiles_x, a.tiles_y = f.n_seg, a.m_tiles = cout / U2_BM, a.n_chunks = cin / cc;
    a.seg_tiles = f.seg_tiles, a.tiles_total_y = f.tiles_total_y;
    a.yb = y, a.hbuf = ws, a.k4 = k4, a.noise = noise, a.noise_w = noise_w, a.bias = bias, a.noise_batch_stride = noise_batch_stride;
    a.src = src, a.noise_slot = noise_slot, a.post_s = post_s;
    const size_t lds_bytes = sizeof(float) * ((size_t)2 * u2_a_floats(cc) + (size_t)2 * u2_pbuf(cc) + (size_t)((cin + 3) & ~3) + 3 * U2_BM + 4 * 3 * 64 * 8);
    static_assert((size_t)2 * u2_a_floats(8) + (size_t)2 * u2_pbuf(8) >= 2 * 4 * 3 * 64 * 8, "the exchange region lives in the operand buffers");
    if (lds_bytes > 80 * 1024) return MAUA_ENOSYS;  // two workgroups per CU
    const int64_t blocks = (int64_t)batch * f.n_seg * f.tiles_x * a.m_tiles;
    static unsigned long long lds_ok = 0, lds_ok_pre = 0;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<8, 2, false>), &lds_ok, 160 * 1024)) return rc;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<8, 2, true>), &lds_ok_pre, 160 * 1024)) return rc;
    snprintf(g_up2d_instance, sizeof(g_up2d_instance), s ? "modconv_up2d_kernel<8, 2, false, 32>" : "modconv_up2d_kernel<8, 2, true, 32>");
    hipStream_t st = (hipStream_t)stream;
#ifdef MAUA_EXPERIMENTS
    if (getenv("MAUA_FUSE_DEBUG")) {
        int occ = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(modconv_up2d_kernel<8, 2>), 256, lds_bytes);
        fprintf(stderr, "[maua_upconv_blur] %d->%d @%dx%d B=%d: x tiles %d, y tiles %d in %d segments of %d, %lld workgroups, LDS %zu B, occupancy %d WG/CU\n",
                cin, cout, h, w, batch, f.tiles_x, f.tiles_total_y, f.n_seg, f.seg_tiles, (long long)blocks, lds_bytes, occ);
    }
#endif
    if (s) hipLaunchKernelGGL((modconv_up2d_kernel<8, 2, false>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL((modconv_up2d_kernel<8, 2, true>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    MAUA_LAUNCH_CHECK();
    if (f.n_seg > 1) {
        const int64_t threads = (int64_t)batch * cout * (f.n_seg - 1) * (2 * w / 4);
        hipLaunchKernelGGL(up2d_seam_kernel, dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0, st, a, f.n_seg - 1);
        MAUA_LAUNCH_CHECK();
    }
    return 0;
}

#ifdef MAUA_EXPERIMENTS
// Experiment entry (tools/fuse_probe.py): the transposed convolution with the Blur + noise + bias + activation in its epilogue, tile halos
// taken as zero; `extra_pct` launches that many per cent of redundant tiles on top (the price of an overlapped tiling).
extern "C" int maua_exp_upconv_blur_fused_f32(const float* x, const float* wq, const float* s, int s_stride, const float* d, float* yb,
                                              const float* k4, const float* noise, int64_t noise_batch_stride, const float* noise_w,
                                              const float* bias, int batch, int cin, int cout, int h, int w, float wscale, int extra_pct,
                                              void* stream) {
    if (!maua_modconv_up2d_ok(cin, cout, h, w) || !yb || !k4) return MAUA_EINVAL;
    const int cc = u2_cc(cin);
    if (cc != 8) return MAUA_ENOSYS;
    Up2dArgs a{};
    a.x = x, a.wq = wq, a.s = s, a.d = d, a.y = nullptr, a.xcol = nullptr;
    a.B = batch, a.Cin = cin, a.Cout = cout, a.H = h, a.W = w, a.s_stride = s_stride, a.wscale = wscale;
    a.tiles_x = w / 32, a.tiles_y = h / 8, a.m_tiles = cout / U2_BM, a.n_chunks = cin / cc;
    a.yb = yb, a.k4 = k4, a.noise = noise, a.noise_w = noise_w, a.bias = bias, a.noise_batch_stride = noise_batch_stride;
    const size_t k_loop = sizeof(float) * ((size_t)2 * u2_a_floats(cc) + (size_t)2 * u2_pbuf(cc) + (size_t)((cin + 3) & ~3) + 3 * U2_BM);
    const size_t lds_bytes = k_loop > 49152 + 4096 ? k_loop : 49152 + 4096;
    const int64_t blocks = (int64_t)batch * a.tiles_y * a.tiles_x * a.m_tiles;
    a.real_blocks = (int)blocks;
    const int64_t launched = blocks + blocks * extra_pct / 100;
    static unsigned long long lds_ok = 0;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<8, 1>), &lds_ok, 160 * 1024)) return rc;
    hipLaunchKernelGGL((modconv_up2d_kernel<8, 1>), dim3((unsigned)launched), dim3(256), lds_bytes, (hipStream_t)stream, a);
    MAUA_LAUNCH_CHECK();
    return 0;
}
#endif

int64_t maua_up2d_ws_floats(int batch, int cin, int h) { return (int64_t)batch * cin * h; }

// ---- 16-wide inputs, K split over several workgroups per tile, partial raw maps in slabs (the low-resolution entry, modconv.hip)
int maua_up2d16_ok(int cin, int cout, int h, int w) { return cin > 0 && cout > 0 && cin % 8 == 0 && cout % U2_BM == 0 && w == 16 && h >= 16 && h % 16 == 0; }

// splits: until the launch covers the chip twice (two workgroups per CU), never below four K steps per workgroup
int maua_up2d16_splits(int batch, int cin, int cout, int h, int w) {
    const int n_chunks = cin / 8;
    const int64_t base = (int64_t)batch * (h / 16) * (w / 16) * (cout / U2_BM);
    int splits = 1;
    while (base * splits < 512 && n_chunks / (splits * 2) >= 4) splits *= 2;
    return splits;
}

// y: `splits` slabs of [B][Cout][2H+1][2W+1] (split k's partial map, gain = wscale; the edge lines — row 2H, column 2W — in slab 0 only);
// xcol: [B][Cin][H] (the exported last input column for the edge kernel)
int maua_up2d16_launch(const float* x, const float* wq, const float* s, int s_stride, float* y, float* xcol, int batch, int cin, int cout, int h,
                       int w, float wscale, int* splits_out, void* stream) {
    if (!maua_up2d16_ok(cin, cout, h, w) || !x || !wq || !s || !y || !xcol || !splits_out) return MAUA_EINVAL;
    if ((int64_t)cin * h * w * 4 > 0x7fffffffLL || (int64_t)U2_NU * cin * cout * 4 > 0x7fffffffLL) return MAUA_EINVAL;  // descriptor ranges
    if ((int64_t)U2_BM * (2 * h + 1) * (2 * w + 1) * 4 > 0xffffffffLL) return MAUA_EINVAL;                              // 32-bit store offsets
    constexpr int cc = 8;
    Up2dArgs a{};
    a.x = x, a.wq = wq, a.s = s, a.d = nullptr, a.y = y, a.xcol = xcol;
    a.B = batch, a.Cin = cin, a.Cout = cout, a.H = h, a.W = w, a.s_stride = s_stride, a.wscale = wscale;
    a.tiles_x = w / 16, a.tiles_y = h / 16, a.m_tiles = cout / U2_BM, a.n_chunks = cin / cc;
    a.splits = maua_up2d16_splits(batch, cin, cout, h, w);
    a.chunks_per_split = (a.n_chunks + a.splits - 1) / a.splits;
    a.splits = (a.n_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    a.slab = (int64_t)batch * cout * (2 * h + 1) * (2 * w + 1);
    *splits_out = a.splits;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = sizeof(float) * ((size_t)2 * u2_a_floats(cc) + (size_t)2 * u2_pbuf(cc) + (size_t)((cin + 3) & ~3) + 3 * U2_BM);
    const int64_t blocks = (int64_t)batch * a.tiles_y * a.tiles_x * a.m_tiles * a.splits;
    static unsigned long long lds_ok = 0;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<8, 0, false, 16>), &lds_ok, 160 * 1024)) return rc;
    snprintf(g_up2d_instance, sizeof(g_up2d_instance), "modconv_up2d_kernel<8, 0, false, 16>");
    hipLaunchKernelGGL((modconv_up2d_kernel<8, 0, false, 16>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    MAUA_LAUNCH_CHECK();
    const int nt0 = ceil_div(w + 1, 16), nt1 = ceil_div(h, 16);
    const int64_t eblocks = (int64_t)batch * (cout / 16) * (nt0 + nt1);
    hipLaunchKernelGGL(up2d_edge_kernel, dim3((unsigned)eblocks), dim3(256), 0, st, a, wq + (size_t)U2_NU * cin * cout, nt0, nt1);
    MAUA_LAUNCH_CHECK();
    return 0;
}

int maua_up2d_launch(const float* x, const float* wq, const float* s, int s_stride, const float* d, float* y, float* ws, int batch, int cin,
                     int cout, int h, int w, float wscale, void* stream) {
    if (!maua_modconv_up2d_ok(cin, cout, h, w) || !ws) return MAUA_EINVAL;
    if ((int64_t)cin * h * w * 4 > 0x7fffffffLL || (int64_t)U2_NU * cin * cout * 4 > 0x7fffffffLL) return MAUA_EINVAL;  // descriptor ranges
    if ((int64_t)U2_BM * (2 * h + 1) * (2 * w + 1) * 4 > 0xffffffffLL) return MAUA_EINVAL;                              // 32-bit store offsets
    Up2dArgs a{};
    a.x = x, a.wq = wq, a.s = s, a.d = d, a.y = y, a.xcol = ws;
    a.B = batch, a.Cin = cin, a.Cout = cout, a.H = h, a.W = w, a.s_stride = s_stride, a.wscale = wscale;
    const int cc = u2_cc(cin);
    a.tiles_x = w / 32, a.tiles_y = h / 8, a.m_tiles = cout / U2_BM, a.n_chunks = cin / cc;
    a.splits = 1, a.chunks_per_split = a.n_chunks, a.slab = 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = sizeof(float) * ((size_t)2 * u2_a_floats(cc) + (size_t)2 * u2_pbuf(cc) + (size_t)((cin + 3) & ~3) + 3 * U2_BM);
    const int64_t blocks = (int64_t)batch * a.tiles_y * a.tiles_x * a.m_tiles;
    static unsigned long long lds_ok[4] = {0, 0, 0, 0};  // per instance: devices on which the attribute has been set (common.h)
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<4, 0, false>), &lds_ok[0], 160 * 1024)) return rc;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<8, 0, false>), &lds_ok[1], 160 * 1024)) return rc;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<4, 0, true>), &lds_ok[2], 160 * 1024)) return rc;
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_up2d_kernel<8, 0, true>), &lds_ok[3], 160 * 1024)) return rc;
    snprintf(g_up2d_instance, sizeof(g_up2d_instance), s ? "modconv_up2d_kernel<%d, 0, false, 32>" : "modconv_up2d_kernel<%d, 0, true, 32>", cc);
    if (cc == 8 && s) hipLaunchKernelGGL((modconv_up2d_kernel<8, 0, false>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    else if (cc == 8) hipLaunchKernelGGL((modconv_up2d_kernel<8, 0, true>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    else if (s) hipLaunchKernelGGL((modconv_up2d_kernel<4, 0, false>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    else hipLaunchKernelGGL((modconv_up2d_kernel<4, 0, true>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    MAUA_LAUNCH_CHECK();
    // edge lines: W + 1 positions along the bottom row (incl. the corner), H along the right column
    const int nt0 = ceil_div(w + 1, 16), nt1 = ceil_div(h, 16);
    const int64_t eblocks = (int64_t)batch * (cout / 16) * (nt0 + nt1);
    hipLaunchKernelGGL(up2d_edge_kernel, dim3((unsigned)eblocks), dim3(256), 0, st, a, wq + (size_t)U2_NU * cin * cout, nt0, nt1);
    MAUA_LAUNCH_CHECK();
    return 0;
}

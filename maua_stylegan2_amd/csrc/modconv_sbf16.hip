// SIDE MEASUREMENT, never the headline path: plain 3x3 modulated convolution with SPLIT-bf16 products on the
// bf16 matrix cores (mode 7 of maua_modconv3x3_f32), fp32 accumulation.
//
// Behavioural contract: /root/reference/models/stylegan2.py:217-254 (plain branch :248-252) + the StyledConv tail :338-343, in the
// input-scale -> shared-weight contraction -> output-demod formulation of modconv.hip.  What changes is the arithmetic of the
// contraction: every fp32 operand is split into two bf16 terms, a = a_h + a_l (a_h = bf16(a), a_l = bf16(a - a_h)), and
//
//     a b  ~=  a_h b_h + a_h b_l + a_l b_h          (the a_l b_l term, <= 2^-16 |a b|, is dropped; products and sums in fp32)
//
// i.e. three v_mfma_f32_32x32x16_bf16 per K slice instead of one fp32 MFMA.  The bf16 matrix pipe is 16 x the fp32 one
// (MI355X_MICROARCH.md: 2.5 PFLOP/s vs 157.3 TFLOP/s), so the DIRECT 9-tap form costs 9 x 3 / 16 = 1.7 fp32-MFMA units per MAC
// against 3 for the 2-D Winograd kernel — without its input transforms, weight-tile DMA and exchange epilogue.  The relative error
// of a product is <= 2^-16 + 2^-17 (dropped term + the rounding of the two low halves), of the same order as the fp32 Winograd
// forms (measured in tests/test_layers_gpu.py::test_modconv_split_bf16_vs_oracle and reported by bench.py next to the time).
// The headline dtype stays f32: this mode is OFF unless ModulatedConv2d.split_bf16_min_cout is lowered (bench.py side_configs).
//
// Work decomposition: a workgroup owns 128 output channels x (8 rows x 32 columns) pixels; wave w takes channels 32 w .. 32 w + 31
// (one 32-row m-tile) x all 8 rows (eight 32-column n-tiles): 8 accumulator tiles of 32 x 32 = 128 registers — every wave streams
// only ITS quarter of the packed weight (the first version split 2 x 2: two waves fetched the same weight records, 6.4 TB/s of L2
// reads at 0.38 ms per 256-channel layer).  K runs over chunks of 16 input channels x 9 taps.
//   B operand: the 10 x 34 halo patch of a chunk is fetched with buffer loads (lane = pixel, scalar channel offset), scaled by the
//     style, split, and written to LDS as 16-byte records [hi | lo][k half][row][col][8 channels]; a tap is a shifted view of the
//     patch (no im2col), one conflict-free ds_read_b128 per (n-tile, hi | lo).  Double buffered: the next chunk's loads are in
//     flight under the nine taps of the current one.
//   A operand: the packed weight (maua_pack_weight_sbf16_f32) is stored in HBM in MFMA lane order, [m-tile][chunk][tap][hi | lo]
//     [lane][8], and goes straight to registers (one 16-byte load per lane and half, two taps ahead): no LDS, no DMA.
// The transposed layers (mode 8) run the same arithmetic in their polyphase form on the input grid — modconv_sbf16_up_kernel below.
#include "common.h"

#include <cstdio>
#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__)
#define MAUA_DEVICE_PASS 1
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_BM = 128;   // output channels per workgroup
constexpr int SB_KC = 16;    // input channels per K chunk = K of v_mfma_f32_32x32x16_bf16
constexpr int SB_TH = 8, SB_TW = 32;
constexpr int SB_PH = SB_TH + 2, SB_PW = SB_TW + 2;
constexpr int SB_ITEMS = 2 * SB_PH * SB_PW;             // (k half, row, col) records per chunk and hi | lo plane
constexpr int SB_PER_THREAD = (SB_ITEMS + 255) / 256;   // 3
constexpr int SB_PLANE_BYTES = SB_ITEMS * 16;           // one hi (or lo) plane of a chunk
constexpr int SB_BUF_BYTES = 2 * SB_PLANE_BYTES;
constexpr unsigned SB_OOB = 0x80000000u;

struct SbArgs {
    const float* x;
    const bf16x8* wq;
    const float* s;
    const float* d;
    const float* noise;
    const float* noise_w;
    const float* bias;
    float* y;
    int B, Cin, Cout, H, W;
    int s_stride;
    float wscale;
    int fuse_act;
    int64_t noise_batch_stride;
    int tiles_x, tiles_y, m_tiles, n_chunks;
    const maua_frame_source_t* src;
    int noise_slot;
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const bf16x2 l = __builtin_convertvector(f32x2{a - hf.x, b - hf.y}, bf16x2);
    hi = __builtin_bit_cast(unsigned, h), lo = __builtin_bit_cast(unsigned, l);
}

// Tap table of the plain convolution: tap (ky, kx) reads patch record (n + ky, l + kx) (patch origin = (ty0 - 1, tx0 - 1)).  (The kernel is
// written against a table so that other tap sets can be tried; the transposed layers have their own kernel below.)
struct SbTap { int wt, ro, co; };  // packed weight tap (ky * 3 + kx), patch row / column offset of the record
template <int PHASE> struct SbTaps;
template <> struct SbTaps<-1> { static constexpr int NT = 9; static constexpr SbTap t[9] = {{0, 0, 0}, {1, 0, 1}, {2, 0, 2}, {3, 1, 0}, {4, 1, 1}, {5, 1, 2}, {6, 2, 0}, {7, 2, 1}, {8, 2, 2}}; };

template <int PHASE>
__global__ __launch_bounds__(256, 2) void modconv_sbf16_kernel(SbArgs p) {
    using Taps = SbTaps<PHASE>;
    constexpr int NT = Taps::NT;
    constexpr int STEPS = 2 * NT;        // half-taps of a chunk: four n-tiles (12 matrix instructions) each
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // LDS: patch[2 buffers][hi | lo][k half][PH][PW] 16-byte records | Ss[Cin] | Eg[128] | Eb[128]
    float* Ss = reinterpret_cast<float*>(lds_raw + 2 * SB_BUF_BYTES);
    float* Eg = Ss + p.Cin;
    float* Eb = Eg + SB_BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi32 = lane >> 5;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int mt_id = t % p.m_tiles;
    t /= p.m_tiles;
    const int tile_x = t % p.tiles_x;
    t /= p.tiles_x;
    const int tile_y = t % p.tiles_y;
    const int b0 = t / p.tiles_y;
    const int ty0 = tile_y * SB_TH, tx0 = tile_x * SB_TW;
    const int m0 = mt_id * SB_BM;
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_bytes = (unsigned)plane * 4u;

    const bool act = p.fuse_act != 0;
    const float act_gain = act ? 1.41421356237309515f : 1.f;
    for (int e = tid; e < p.Cin; e += 256) Ss[e] = p.s[(size_t)b0 * p.s_stride + e];
    for (int i = tid; i < SB_BM; i += 256) {
        float gain = p.wscale * act_gain;
        if (p.d) gain *= p.d[(size_t)b0 * p.Cout + m0 + i];
        Eg[i] = gain;
        Eb[i] = (act && p.bias) ? p.bias[m0 + i] * act_gain : 0.f;
    }

    // ---- staging map of this thread: items it, it + 256, it + 512 of (k half, row, col); lane = consecutive columns
    unsigned item_voff[SB_PER_THREAD];   // byte offset of the item's pixel in channel (8 * k half) of the chunk, or out of range
    unsigned item_lds[SB_PER_THREAD];    // byte offset of its record inside a plane
    int item_kb[SB_PER_THREAD];
#pragma unroll
    for (int q = 0; q < SB_PER_THREAD; ++q) {
        const int idx = tid + 256 * q;
        const int kb = idx / (SB_PH * SB_PW), rem = idx % (SB_PH * SB_PW);
        const int row = rem / SB_PW, col = rem % SB_PW;
        const int yy = ty0 - 1 + row, xx = tx0 - 1 + col;
        const bool ok = idx < SB_ITEMS && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        item_voff[q] = ok ? (unsigned)kb * 8u * plane_bytes + ((unsigned)yy * (unsigned)p.W + (unsigned)xx) * 4u : SB_OOB;
        item_lds[q] = (unsigned)idx * 16u;
        item_kb[q] = idx < SB_ITEMS ? kb : -1;
    }
#ifdef MAUA_DEVICE_PASS
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (size_t)b0 * p.Cin * plane, 0, (int)((unsigned)p.Cin * plane_bytes), 0x00020000);
#endif
    // one item (8 channels of one patch pixel) is in flight at a time: item q of the next chunk is requested at tap 3 q and written to LDS
    // three taps later — 8 staging registers instead of 24
    float stage[8];
    auto fetch = [&](int chunk, int q) {
#ifdef MAUA_DEVICE_PASS
#pragma unroll
        for (int e = 0; e < 8; ++e)
            stage[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, item_voff[q],
                                                                                         (unsigned)(chunk * SB_KC + e) * plane_bytes, 0));
#else
        (void)chunk, (void)q;
#endif
    };
    auto commit = [&](int chunk, int buf, int q) {  // style, split, 16-byte records into the hi and lo planes of buffer `buf`
        if (item_kb[q] < 0) return;
        const float* sp = Ss + chunk * SB_KC + item_kb[q] * 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
        unsigned h[4], l[4];
        const float(&v)[8] = stage;
        split2(v[0] * s0[0], v[1] * s0[1], h[0], l[0]);
        split2(v[2] * s0[2], v[3] * s0[3], h[1], l[1]);
        split2(v[4] * s1[0], v[5] * s1[1], h[2], l[2]);
        split2(v[6] * s1[2], v[7] * s1[3], h[3], l[3]);
        unsigned char* dst = lds_raw + buf * SB_BUF_BYTES + item_lds[q];
        *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(dst + SB_PLANE_BYTES) = u32x4{l[0], l[1], l[2], l[3]};
    };

    // ---- accumulators [n-tile = tile row]
    f32x16 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[n][j] = 0.f;

    // B records of this lane: k half hi32, patch row n + ky, column l31 + kx
    const unsigned b_base = (unsigned)(hi32 * SB_PH * SB_PW + l31) * 16u;
    // A records: [m-tile][chunk][tap][hi | lo][lane] of 16 bytes; this wave's m-tile
    const bf16x8* wq = p.wq + ((size_t)(mt_id * 4 + wave) * p.n_chunks * 9 * 2) * 64 + lane;
    auto load_a = [&](bf16x8(&a)[2], int chunk, int wt) {
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) a[hl] = wq[((size_t)(chunk * 9 + wt) * 2 + hl) * 64];
    };

    __syncthreads();  // styles and gains are in LDS
#pragma unroll
    for (int q = 0; q < SB_PER_THREAD; ++q) {
        fetch(0, q);
        commit(0, 0, q);
    }
    __syncthreads();
    // Explicitly software-pipelined, sched_barrier-pinned (left alone the compiler sinks every load to just in front of its first use —
    // the weight loads of a tap right before its matrix instructions, a global round trip exposed per tap: 33 % of the wave cycles
    // parked at s_waitcnt in the first version):
    //   * weight records are requested TWO taps ahead (3 x 2 x 4 registers), across chunk boundaries;
    //   * a tap runs as two half-taps of four n-tiles (12 matrix instructions); the eight B records of half-tap h + 1 are read while
    //     the matrix instructions of half-tap h run (2 x 32 registers); an accumulator is touched every fourth instruction; the small
    //     terms go first.
    auto load_a_ahead = [&](bf16x8(&a)[2], int chunk, auto tap_c) {  // tap number `tap` (may run past NT: into the following chunks)
        constexpr int tap = decltype(tap_c)::value, dc = tap / NT, tt = tap % NT;
        if (chunk + dc < p.n_chunks) load_a(a, chunk + dc, Taps::t[tt].wt);
    };
    bf16x8 a0[2], a1[2], a2[2];
    load_a_ahead(a0, 0, std::integral_constant<int, 0>{});
    load_a_ahead(a1, 0, std::integral_constant<int, 1>{});
    int cur = 0;
    static_assert(SB_PER_THREAD == 3 && NT == 9, "the tap loop stages one item per three taps");
    bf16x8 bh[2][4], bl[2][4];
    auto read_b = [&](const unsigned char* pb, auto h_c, bf16x8(&dh)[4], bf16x8(&dl)[4]) {
        constexpr int h = decltype(h_c)::value, tap = h / 2, nq = h % 2, ro = Taps::t[tap].ro, co = Taps::t[tap].co;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* rec = pb + ((4 * nq + i + ro) * SB_PW + co) * 16;
            dh[i] = *reinterpret_cast<const bf16x8*>(rec);
            dl[i] = *reinterpret_cast<const bf16x8*>(rec + SB_PLANE_BYTES);
        }
    };
    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
        const bool more = chunk + 1 < p.n_chunks;
        const unsigned char* pb = lds_raw + cur * SB_BUF_BYTES + b_base;
        read_b(pb, std::integral_constant<int, 0>{}, bh[0], bl[0]);
        static_for<0, STEPS>([&](auto h_c) {
            constexpr int h = decltype(h_c)::value, tap = h / 2, nq = h % 2, slot = h % 2;
            if constexpr (nq == 0) {
                if (more) {
                    if constexpr (tap % 3 == 0) {
                        if constexpr (tap > 0) commit(chunk + 1, cur ^ 1, tap / 3 - 1);
                        fetch(chunk + 1, tap / 3);
                    }
                }
                load_a_ahead(a2, chunk, std::integral_constant<int, tap + 2>{});
            }
            if constexpr (h + 1 < STEPS) read_b(pb, std::integral_constant<int, h + 1>{}, bh[slot ^ 1], bl[slot ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * nq + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], bh[slot][i], acc[4 * nq + i], 0, 0, 0);  // a_l b_h
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * nq + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bl[slot][i], acc[4 * nq + i], 0, 0, 0);  // a_h b_l
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * nq + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], bh[slot][i], acc[4 * nq + i], 0, 0, 0);  // a_h b_h
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (nq == 1) {
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) a0[hl] = a1[hl], a1[hl] = a2[hl];
            }
        });
        if (more) commit(chunk + 1, cur ^ 1, SB_PER_THREAD - 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: gain (wscale * demod), noise, bias, leaky ReLU * sqrt2 as max(t, 0.2 t) on pre-scaled operands; a lane holds
    // 16 output channels of one pixel column per tile: rows (j & 3) + 8 (j >> 2) + 4 hi32 of the 32 x 32 result tile
    const float* noise_base = p.noise;
    int64_t noise_bstride = p.noise_batch_stride;
    if (p.src) {
        noise_bstride = p.src->noise_stride[p.noise_slot];
        noise_base = p.src->noise[p.noise_slot];
        if (noise_base) noise_base += (int64_t)p.src->frame0 * noise_bstride;
    }
    const float nw = (act && noise_base) ? p.noise_w[0] * act_gain : 0.f;
    const float slope = act ? 0.2f : 1.f;
    float* yimg = p.y + ((size_t)b0 * p.Cout + m0 + wave * 32) * plane;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int oy = ty0 + n, ox = tx0 + l31;
        const size_t pix = (size_t)oy * p.W + ox;
        const float nz = nw != 0.f ? noise_base[(size_t)b0 * noise_bstride + pix] * nw : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ol = (j & 3) + 8 * (j >> 2) + 4 * hi32;
            const float tt = fmaf(acc[n][j], Eg[wave * 32 + ol], nz + Eb[wave * 32 + ol]);
            yimg[(size_t)ol * plane + pix] = fmaxf(tt, tt * slope);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The transposed layers in ONE launch (mode 8): a workgroup owns one 32-channel m-tile x (8 rows x 32 columns of POSITIONS) and ALL
// FOUR output phases; wave w takes position rows 2 w, 2 w + 1: 4 phases x 2 n-tiles = 8 accumulator tiles.  A transposed layer has four
// output phases per input position, i.e. a quarter of a plain layer's accumulator reuse per weight record, so here the records go through
// LDS once per workgroup (LDS-DMA of the 18 KB (9 taps x hi | lo x 64 lanes x 16 bytes) of a chunk, double buffered; the packed layout IS
// that tile) and every wave reads them from there; the patch (9 x 33 positions) is staged as in the plain kernel.
constexpr int SU_PH = SB_TH + 1, SU_PW = SB_TW + 1;           // the taps reach one row up and one column left
constexpr int SU_ITEMS = 2 * SU_PH * SU_PW;                    // 594 records per chunk and hi | lo plane
constexpr int SU_PER_THREAD = (SU_ITEMS + 255) / 256;          // 3
constexpr int SU_PLANE_BYTES = SU_ITEMS * 16;
constexpr int SU_BBUF_BYTES = 2 * SU_PLANE_BYTES;
constexpr int SU_ABUF_BYTES = 18 * 1024;
// (phase = 2 * (odd row) + (odd column), weight tap ky * 3 + kx, patch row / column offset of the record: patch origin = (ty0 - 1, tx0 - 1))
struct SuTap { int ph, wt, ro, co; };
__device__ constexpr SuTap SU_TAPS[9] = {{0, 0, 1, 1}, {1, 1, 1, 1}, {2, 3, 1, 1}, {3, 4, 1, 1}, {0, 2, 1, 0},
                                         {1, 7, 0, 1}, {2, 5, 1, 0}, {0, 6, 0, 1}, {0, 8, 0, 0}};

__global__ __launch_bounds__(256, 2) void modconv_sbf16_up_kernel(SbArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // LDS: A[2][18 KB] | patch[2 buffers][hi | lo][k half][9][33] 16-byte records | Ss[Cin] | Eg[32]
    unsigned char* Bs = lds_raw + 2 * SU_ABUF_BYTES;
    float* Ss = reinterpret_cast<float*>(Bs + 2 * SU_BBUF_BYTES);
    float* Eg = Ss + p.Cin;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi32 = lane >> 5;

    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int mt_id = t % p.m_tiles;  // (32-channel m-tiles here)
    t /= p.m_tiles;
    const int tile_x = t % p.tiles_x;
    t /= p.tiles_x;
    const int tile_y = t % p.tiles_y;
    const int b0 = t / p.tiles_y;
    const int ty0 = tile_y * SB_TH, tx0 = tile_x * SB_TW;
    const int m0 = mt_id * 32;
    const size_t plane = (size_t)p.H * p.W;
    const unsigned plane_bytes = (unsigned)plane * 4u;

    for (int e = tid; e < p.Cin; e += 256) Ss[e] = p.s[(size_t)b0 * p.s_stride + e];
    if (tid < 32) Eg[tid] = p.wscale * (p.d ? p.d[(size_t)b0 * p.Cout + m0 + tid] : 1.f);

    unsigned item_voff[SU_PER_THREAD], item_lds[SU_PER_THREAD];
    int item_kb[SU_PER_THREAD];
#pragma unroll
    for (int q = 0; q < SU_PER_THREAD; ++q) {
        const int idx = tid + 256 * q;
        const int kb = idx / (SU_PH * SU_PW), rem = idx % (SU_PH * SU_PW);
        const int row = rem / SU_PW, col = rem % SU_PW;
        const int yy = ty0 - 1 + row, xx = tx0 - 1 + col;
        const bool ok = idx < SU_ITEMS && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        item_voff[q] = ok ? (unsigned)kb * 8u * plane_bytes + ((unsigned)yy * (unsigned)p.W + (unsigned)xx) * 4u : SB_OOB;
        item_lds[q] = (unsigned)idx * 16u;
        item_kb[q] = idx < SU_ITEMS ? kb : -1;
    }
#ifdef MAUA_DEVICE_PASS
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (size_t)b0 * p.Cin * plane, 0, (int)((unsigned)p.Cin * plane_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16x8*>(p.wq) + (size_t)mt_id * p.n_chunks * 18 * 64, 0, (int)((unsigned)p.n_chunks * SU_ABUF_BYTES), 0x00020000);
#endif
    float stage[SU_PER_THREAD][8];
    auto fetch = [&](int chunk) {
#ifdef MAUA_DEVICE_PASS
#pragma unroll
        for (int q = 0; q < SU_PER_THREAD; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                stage[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, item_voff[q],
                                                                                             (unsigned)(chunk * SB_KC + e) * plane_bytes, 0));
#else
        (void)chunk;
#endif
    };
    auto commit = [&](int chunk, int buf) {
#pragma unroll
        for (int q = 0; q < SU_PER_THREAD; ++q) {
            if (item_kb[q] < 0) continue;
            const float* sp = Ss + chunk * SB_KC + item_kb[q] * 8;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
            unsigned h[4], l[4];
            split2(stage[q][0] * s0[0], stage[q][1] * s0[1], h[0], l[0]);
            split2(stage[q][2] * s0[2], stage[q][3] * s0[3], h[1], l[1]);
            split2(stage[q][4] * s1[0], stage[q][5] * s1[1], h[2], l[2]);
            split2(stage[q][6] * s1[2], stage[q][7] * s1[3], h[3], l[3]);
            unsigned char* dst = Bs + buf * SU_BBUF_BYTES + item_lds[q];
            *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(dst + SU_PLANE_BYTES) = u32x4{l[0], l[1], l[2], l[3]};
        }
    };
    auto issue_a = [&](int chunk, int buf) {  // 18 pieces of 1 KiB, dealt to the four waves
#ifdef MAUA_DEVICE_PASS
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = wave + 4 * k;  // (scalar)
            if (i < 18)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(lds_raw + buf * SU_ABUF_BYTES + i * 1024),
                                                         16, i * 1024 + lane * 16, chunk * SU_ABUF_BYTES, 0, 0);
        }
#else
        (void)chunk, (void)buf;
#endif
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[ph][n][j] = 0.f;

    // B records of this lane: k half hi32, patch row 2 wave + n + ro, column l31 + co;  A records: (tap * 2 + hi | lo) * 1 KiB + lane * 16
    const unsigned b_base = (unsigned)((hi32 * SU_PH + 2 * wave) * SU_PW + l31) * 16u;
    const unsigned a_base = (unsigned)lane * 16u;

    __syncthreads();  // styles and gains are in LDS
    issue_a(0, 0);
    fetch(0);
    commit(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    bf16x8 av[2][2], bh[2][2], bl[2][2];  // [pipeline slot][hi | lo] / [slot][n-tile]
    auto read_tap = [&](const unsigned char* pa, const unsigned char* pb, auto t_c, int slot) {
        constexpr int tp = decltype(t_c)::value;
        constexpr SuTap T = SU_TAPS[tp];
        av[slot][0] = *reinterpret_cast<const bf16x8*>(pa + (T.wt * 2) * 1024);
        av[slot][1] = *reinterpret_cast<const bf16x8*>(pa + (T.wt * 2 + 1) * 1024);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const unsigned char* rec = pb + ((n + T.ro) * SU_PW + T.co) * 16;
            bh[slot][n] = *reinterpret_cast<const bf16x8*>(rec);
            bl[slot][n] = *reinterpret_cast<const bf16x8*>(rec + SU_PLANE_BYTES);
        }
    };
    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
        const bool more = chunk + 1 < p.n_chunks;
        if (more) {
            issue_a(chunk + 1, cur ^ 1);
            fetch(chunk + 1);
        }
        const unsigned char* pa = lds_raw + cur * SU_ABUF_BYTES + a_base;
        const unsigned char* pb = Bs + cur * SU_BBUF_BYTES + b_base;
        read_tap(pa, pb, std::integral_constant<int, 0>{}, 0);
        static_for<0, 9>([&](auto t_c) {
            constexpr int tp = decltype(t_c)::value, slot = tp % 2, ph = SU_TAPS[tp].ph;
            if constexpr (tp + 1 < 9) read_tap(pa, pb, std::integral_constant<int, tp + 1>{}, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[ph][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][1], bh[slot][n], acc[ph][n], 0, 0, 0);  // a_l b_h
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[ph][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][0], bl[slot][n], acc[ph][n], 0, 0, 0);  // a_h b_l
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[ph][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][0], bh[slot][n], acc[ph][n], 0, 0, 0);  // a_h b_h
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) commit(chunk + 1, cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: the 2 x 2 output block of every position — raw map [B, Cout, 2 H + 1, 2 W + 1], gain = wscale * demod
    const int OW = 2 * p.W + 1;
    const size_t plane_out = (size_t)(2 * p.H + 1) * OW;
    float* yup = p.y + ((size_t)b0 * p.Cout + m0) * plane_out;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const size_t pix = (size_t)(2 * (ty0 + 2 * wave + n)) * OW + 2 * (tx0 + l31);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ol = (j & 3) + 8 * (j >> 2) + 4 * hi32;
            const float g = Eg[ol];
            float* dst = yup + (size_t)ol * plane_out + pix;
            dst[0] = acc[0][n][j] * g, dst[1] = acc[1][n][j] * g;
            dst[OW] = acc[2][n][j] * g, dst[OW + 1] = acc[3][n][j] * g;
        }
    }
}

// the last input column x[b][c][:, W - 1] -> xcol[b][c][:] for the edge lines of the transposed form (modconv_up2d.hip's edge kernel reads
// it with unit stride)
__global__ __launch_bounds__(256) void export_last_column_kernel(const float* __restrict__ x, float* __restrict__ xcol, int64_t rows, int w) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) xcol[i] = x[i * w + (w - 1)];
}

// wq: [cout / 32][cin / 16][9 taps][hi | lo][64 lanes][8] bf16 — lane l of an m-tile holds row l % 32, input channels 8 (l / 32) .. + 7
__global__ __launch_bounds__(256) void pack_weight_sbf16_kernel(const float* __restrict__ w, unsigned* __restrict__ wq, int cout, int cin) {
    const int n_chunks = cin / SB_KC;
    const int64_t total = (int64_t)(cout / 32) * n_chunks * 9 * 64 * 4;  // one thread per (m-tile, chunk, tap, lane, channel pair)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx;
        const int pr = (int)(r % 4);
        r /= 4;
        const int l = (int)(r % 64);
        r /= 64;
        const int tap = (int)(r % 9);
        r /= 9;
        const int chunk = (int)(r % n_chunks);
        const int mt = (int)(r / n_chunks);
        const int o = mt * 32 + (l & 31), c = chunk * SB_KC + 8 * (l >> 5) + 2 * pr;
        unsigned hi, lo;
        split2(w[((size_t)o * cin + c) * 9 + tap], w[((size_t)o * cin + c + 1) * 9 + tap], hi, lo);
        const size_t base = ((((size_t)mt * n_chunks + chunk) * 9 + tap) * 2) * 64;
        wq[(base + l) * 4 + pr] = hi;
        wq[(base + 64 + l) * 4 + pr] = lo;
    }
}

// edge tap matrices [5][cin][cout] of the transposed form: g20, g21, g22 (the kernel's last row), g02, g12 (the rest of its last column) —
// the layout modconv_up2d.hip's edge kernel reads
__global__ __launch_bounds__(256) void pack_edge_taps_kernel(const float* __restrict__ w, float* __restrict__ edge, int cout, int cin) {
    const int64_t total = (int64_t)cout * cin, tsz = total;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int o = (int)(idx % cout), i = (int)(idx / cout);
        const float* g = w + ((size_t)o * cin + i) * 9;
        float* ed = edge + (size_t)i * cout + o;
        ed[0] = g[6], ed[tsz] = g[7], ed[2 * tsz] = g[8], ed[3 * tsz] = g[2], ed[4 * tsz] = g[5];
    }
}

char g_sbf16_instance[48] = "";

}  // namespace

int maua_sbf16_ok(int cin, int cout, int h, int w) {
    return cin > 0 && cout > 0 && cin % SB_KC == 0 && cout % SB_BM == 0 && h % SB_TH == 0 && w % SB_TW == 0;
}
// (the transposed kernel works on 32-channel m-tiles)
int maua_sbf16_up_ok(int cin, int cout, int h, int w) {
    return cin > 0 && cout > 0 && cin % SB_KC == 0 && cout % 32 == 0 && h % SB_TH == 0 && w % SB_TW == 0;
}

const char* maua_sbf16_last_instance() { return g_sbf16_instance; }

namespace {
template <int PHASE>
int sbf16_launch_phase(const SbArgs& a, size_t lds_bytes, hipStream_t st) {
    static unsigned long long lds_ok = 0;  // per launcher: devices on which the attribute has been set (common.h)
    if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_sbf16_kernel<PHASE>), &lds_ok, 160 * 1024)) return rc;
    const int64_t blocks = (int64_t)a.B * a.tiles_y * a.tiles_x * a.m_tiles;
    hipLaunchKernelGGL(modconv_sbf16_kernel<PHASE>, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    MAUA_LAUNCH_CHECK();
    return 0;
}
}  // namespace

// up = 0: the plain convolution (+ tail when fuse_act).  up = 1: the stride-2 transposed convolution, raw map [B, Cout, 2H+1, 2W+1]:
// four phase launches over the positions p < H, q < W, then the edge lines (output row 2H, column 2W) by modconv_up2d.hip's fp32 edge
// kernel on the edge tap matrices stored behind the bf16 records; ws = [B, cin, H] floats (the exported last input column).
int maua_sbf16_launch(const float* x, const void* wq, const float* s, int s_stride, const float* d, float* y, float* ws, int batch, int cin,
                      int cout, int h, int w, int up, float wscale, int fuse_act, const float* noise, int64_t noise_batch_stride,
                      const float* noise_w, const float* bias, const maua_frame_source_t* src, int noise_slot, void* stream) {
    if (!(up ? maua_sbf16_up_ok(cin, cout, h, w) : maua_sbf16_ok(cin, cout, h, w))) return MAUA_EINVAL;
    if ((int64_t)cin * h * w * 4 > 0x7fffffffLL) return MAUA_EINVAL;  // descriptor range / 32-bit offsets
    if (up && (fuse_act || !ws)) return MAUA_EINVAL;
    SbArgs a{};
    a.x = x, a.wq = static_cast<const bf16x8*>(wq), a.s = s, a.d = d, a.noise = noise, a.noise_w = noise_w, a.bias = bias, a.y = y;
    a.B = batch, a.Cin = cin, a.Cout = cout, a.H = h, a.W = w, a.s_stride = s_stride, a.wscale = wscale, a.fuse_act = fuse_act;
    a.noise_batch_stride = noise_batch_stride, a.src = src, a.noise_slot = noise_slot;
    a.tiles_x = w / SB_TW, a.tiles_y = h / SB_TH, a.m_tiles = cout / SB_BM, a.n_chunks = cin / SB_KC;
    const size_t lds_bytes = (size_t)2 * SB_BUF_BYTES + sizeof(float) * ((size_t)cin + 2 * SB_BM);
    hipStream_t st = (hipStream_t)stream;
    if (!up) {
        snprintf(g_sbf16_instance, sizeof(g_sbf16_instance), "modconv_sbf16_kernel<-1>");
        return sbf16_launch_phase<-1>(a, lds_bytes, st);
    }
    snprintf(g_sbf16_instance, sizeof(g_sbf16_instance), "modconv_sbf16_up_kernel");
    {
        static unsigned long long lds_ok = 0;  // per launcher: devices on which the attribute has been set (common.h)
        if (int rc = maua_allow_full_lds(reinterpret_cast<const void*>(modconv_sbf16_up_kernel), &lds_ok, 160 * 1024)) return rc;
        a.m_tiles = cout / 32;
        const size_t up_lds = (size_t)2 * SU_ABUF_BYTES + (size_t)2 * SU_BBUF_BYTES + sizeof(float) * ((size_t)cin + 32);
        const int64_t blocks = (int64_t)batch * a.tiles_y * a.tiles_x * a.m_tiles;
        hipLaunchKernelGGL(modconv_sbf16_up_kernel, dim3((unsigned)blocks), dim3(256), up_lds, st, a);
        MAUA_LAUNCH_CHECK();
    }
    const int64_t rows = (int64_t)batch * cin * h;
    const int64_t blocks = ceil_div64(rows, 256);
    hipLaunchKernelGGL(export_last_column_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x, ws, rows, w);
    MAUA_LAUNCH_CHECK();
    const float* edge = reinterpret_cast<const float*>(static_cast<const unsigned char*>(wq) + (size_t)cout * cin * 9 * 2 * 2);
    return maua_up2d_edge_launch(x, edge, s, s_stride, d, y, ws, batch, cin, cout, h, w, wscale, stream);
}

// bf16 records (cout * cin * 9 * 2 * 2 bytes), then the five fp32 edge tap matrices [5][cin][cout] of the transposed form
extern "C" int64_t maua_pack_weight_sbf16_bytes(int cout, int cin) {
    return maua_sbf16_up_ok(cin, cout, SB_TH, SB_TW) ? (int64_t)cout * cin * 9 * 2 * 2 + (int64_t)5 * cin * cout * 4 : 0;
}

extern "C" int maua_pack_weight_sbf16_f32(const float* w, void* wq, int cout, int cin, void* stream) {
    if (!w || !wq || !maua_sbf16_up_ok(cin, cout, SB_TH, SB_TW)) return MAUA_EINVAL;
    const int64_t total = (int64_t)(cout / 32) * (cin / SB_KC) * 9 * 64 * 4;
    const int64_t blocks = ceil_div64(total, 256);
    hipLaunchKernelGGL(pack_weight_sbf16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<unsigned*>(wq), cout, cin);
    MAUA_LAUNCH_CHECK();
    float* edge = reinterpret_cast<float*>(static_cast<unsigned char*>(wq) + (size_t)cout * cin * 9 * 2 * 2);
    const int64_t eb = ceil_div64((int64_t)cout * cin, 256);
    hipLaunchKernelGGL(pack_edge_taps_kernel, dim3((unsigned)(eb < 4096 ? eb : 4096)), dim3(256), 0, (hipStream_t)stream, w, edge, cout, cin);
    MAUA_LAUNCH_CHECK();
    return 0;
}

extern "C" int maua_modconv_sbf16_ok(int cin, int cout, int h, int w) { return maua_sbf16_ok(cin, cout, h, w); }
extern "C" int maua_modconv_sbf16_up_ok(int cin, int cout, int h, int w) { return maua_sbf16_up_ok(cin, cout, h, w); }

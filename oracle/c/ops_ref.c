/*
 * C restatement (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py) of the two native ops of the reference, written
 * as plain scalar loops so that it shares no code path with either the PyTorch oracle (oracle/ops_oracle.py, conv2d
 * based) or the HIP kernels:
 *
 *   ref_upfirdn2d      <- /root/reference/op/upfirdn2d.py:159-200 (upfirdn2d_native) and the un-tiled gather kernel
 *                         /root/reference/op/upfirdn2d_kernel.cu:49-105: zero-stuff by up, pad (negative = crop),
 *                         TRUE convolution with the taps, keep every down-th sample; layout [major, h, w, minor].
 *   ref_fused_bias_act <- /root/reference/op/fused_bias_act_kernel.cu:18-49: bias index (i / step_b) % size_b,
 *                         act*10+grad switch, out = y * scale.
 *
 * Pinned by tests/test_oracle_golden.py::test_c_restatement_matches_golden against tests/golden/ops_*.npz (outputs
 * of the imported reference).  Accumulation is in double so the comparison tolerance is set by the fp32 reference.
 */
#include <stddef.h>
#include <stdint.h>

int ref_upfirdn2d(const float* x, const float* k, float* y, int major, int in_h, int in_w, int minor, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1) {
    const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    if (out_h <= 0 || out_w <= 0) return -1;
    for (int m = 0; m < major; ++m)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox)
                for (int mi = 0; mi < minor; ++mi) {
                    double acc = 0.0;
                    for (int i = 0; i < kh; ++i) {
                        /* row of the padded, zero-stuffed canvas hit by flipped tap row i */
                        const int cy = oy * down_y + i - pad_y0;
                        if (cy < 0 || cy % up_y != 0 || cy / up_y >= in_h) continue;
                        for (int j = 0; j < kw; ++j) {
                            const int cx = ox * down_x + j - pad_x0;
                            if (cx < 0 || cx % up_x != 0 || cx / up_x >= in_w) continue;
                            const float tap = k[(kh - 1 - i) * kw + (kw - 1 - j)];
                            acc += (double)tap * (double)x[(((size_t)m * in_h + cy / up_y) * in_w + cx / up_x) * minor + mi];
                        }
                    }
                    y[(((size_t)m * out_h + oy) * out_w + ox) * minor + mi] = (float)acc;
                }
    return 0;
}

int ref_fused_bias_act(const float* x, const float* b, const float* ref, float* y, int64_t n, int size_b, int step_b,
                       int act, int grad, float alpha, float scale) {
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i];
        if (b && size_b > 0) v += b[(i / step_b) % size_b];
        const float r = ref ? ref[i] : 0.0f;
        float out;
        switch (act * 10 + grad) {
            case 12:
            case 32: out = 0.0f; break;
            case 30: out = v > 0.0f ? v : v * alpha; break;
            case 31: out = r > 0.0f ? v : v * alpha; break;
            default: out = v; break; /* 10, 11 and the reference's `default:` label */
        }
        y[i] = out * scale;
    }
    return 0;
}

/* render.py:40-43: clamp(-1,1), (x+1)*127.5, truncating cast; [B,3,H,W] -> [B,H,W,3] */
int ref_frames_to_u8(const float* img, uint8_t* out, int batch, int h, int w) {
    const size_t plane = (size_t)h * w;
    for (int b = 0; b < batch; ++b)
        for (size_t p = 0; p < plane; ++p)
            for (int c = 0; c < 3; ++c) {
                float v = img[((size_t)b * 3 + c) * plane + p];
                v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
                out[((size_t)b * plane + p) * 3 + c] = (uint8_t)((v + 1.0f) * 127.5f);
            }
    return 0;
}

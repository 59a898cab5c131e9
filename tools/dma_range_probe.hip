// Probe: range checking of `buffer_load_dwordx4 ... lds` (raw buffer, stride 0) when the 16 bytes of a lane straddle the end of the
// descriptor's range, or start at a "negative" (wrapped) offset: is the check per dword?  (Round 5: tiles of the fused up-sampling kernel
// with 4-byte-aligned operand segments would straddle the image edges.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, float* out, int num_records_bytes, unsigned voffset, unsigned soffset) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, num_records_bytes, 0x00020000);
    if (threadIdx.x == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voffset, (int)soffset, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x < 8) out[threadIdx.x] = lds[threadIdx.x];
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)(i + 100);
    float *d, *o;
    (void)hipMalloc(&d, 4096 * 4);
    (void)hipMalloc(&o, 8 * 4);
    (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    struct Case { const char* what; int bytes; unsigned voff, soff; const float* base; };
    const Case cases[] = {
        {"in range: voffset 64", 1024, 64, 0, d},
        {"straddles the END: num_records 1024, voffset 1020 (dword 0 valid, 1..3 beyond)", 1024, 1020, 0, d},
        {"straddles the END through soffset: voffset 20, soffset 1000", 1024, 20, 1000, d},
        {"wrapped voffset 0xFFFFFFFC on base = src + 16 floats (dword 0 'before' the buffer, 1..3 = elements 0..2)", 1024, 0xFFFFFFFCu, 0, d + 16},
        {"voffset 0x80000000 (the kernels' OOB marker)", 1024, 0x80000000u, 0, d},
        {"wrapped sum: voffset 0xFFFFFFFC + soffset 64 (= 60: in range if the sum wraps at 32 bits)", 1024, 0xFFFFFFFCu, 64, d},
    };
    for (const Case& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, c.base, o, c.bytes, c.voff, c.soff);
        float r[8];
        hipError_t e = hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("%-110s rc %d -> %g %g %g %g\n", c.what, (int)e, r[0], r[1], r[2], r[3]);
    }
    return 0;
}

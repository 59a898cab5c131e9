// Probe: does `buffer_load_dwordx4 ... lds` (LDS DMA, 16 bytes per lane) accept an LDS destination that is only 8-byte aligned, and a
// global source that is only 8-byte aligned?  (Would allow odd channel planes of the 2-D Winograd patch to sit two floats off, which makes
// the 8-byte window reads of the two K lane groups of a half-wave hit disjoint banks.)  Prints what arrives for each combination.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, float* out, int lds_shift_floats, int src_shift_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + lds_shift_floats), 16,
                                             (int)(threadIdx.x * 16 + src_shift_floats * 4), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int ls : {0, 2, 1}) for (int ss : {0, 2, 1}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d, o, ls, ss);
        std::vector<float> r(512);
        hipError_t e = hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        int good = 0, total = 256;
        for (int i = 0; i < 256; ++i) good += r[ls + i] == (float)(ss + i);
        printf("lds_shift %d floats, src_shift %d floats: rc %d, %d / %d in place; first values:", ls, ss, (int)e, good, total);
        for (int i = 0; i < 12; ++i) printf(" %g", r[i]);
        printf("\n");
    }
    return 0;
}

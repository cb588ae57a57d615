// Probe: does an out-of-range `buffer_load_dwordx4 ... lds` write zeros into LDS (or leave it untouched)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* src, float* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = -7.0f;   // sentinel
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..63 out of range (offset beyond num_records)
    int voff = lane < 32 ? lane * 16 : 0x40000000 + lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    // second piece with a non-zero soffset (uniform) into the second KiB
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 256), 16, voff, 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 4096 * 4);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    printf("piece0 in-range  lane0: %g %g %g %g | lane31: %g\n", r[0], r[1], r[2], r[3], r[31 * 4]);
    printf("piece0 OOB       lane32: %g %g | lane63: %g\n", r[32 * 4], r[32 * 4 + 1], r[63 * 4]);
    printf("piece1 (soffset 1024 B = +256 floats) lane0: %g lane1: %g | OOB lane40: %g\n", r[256], r[256 + 4], r[256 + 40 * 4]);
    return 0;
}

// Prints what v_permlane16_swap_b32 / v_permlane32_swap_b32 do to two registers holding (lane) and (100 + lane): the lane
// maps attention.hip's two-strand kernel relies on (round 4).  Build: hipcc --offload-arch=gfx950 -o permlane_probe.bin permlane_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
    const unsigned l = threadIdx.x;
    unsigned a = l, b = 100 + l;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    a = l; b = 100 + l;
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
    unsigned* d; unsigned h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char* names[4] = {"permlane16_swap vdst'", "permlane16_swap src' ", "permlane32_swap vdst'", "permlane32_swap src' "};
    for (int k = 0; k < 4; ++k) {
        printf("%s:", names[k]);
        for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 * k + i]);
        printf("\n");
    }
    return 0;
}

// prints what v_permlane16_swap does (run on the GPU box): hipcc --offload-arch=gfx950 tools/probe_permlane16.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 1000u + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row) printf("row%d lane%2d: a'=%u b'=%u\n", row, row * 16, h[row * 16], h[64 + row * 16]);
    return 0;
}

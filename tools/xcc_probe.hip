// xcc_probe.hip -- which XCD does workgroup b run on?  (HW_REG_XCC_ID vs blockIdx % 8)
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11));
}

int main() {
    const int n = 2048;
    int *d;
    hipMalloc(&d, n * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
    std::vector<int> h(n);
    hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    int hist[16] = {0}, match = 0;
    for (int b = 0; b < n; b++) {
        hist[h[b] & 15]++;
        match += (h[b] & 7) == (b & 7);
    }
    printf("first 24 ids:");
    for (int b = 0; b < 24; b++) printf(" %d", h[b]);
    printf("\nhistogram:");
    for (int x = 0; x < 16; x++) printf(" %d", hist[x]);
    printf("\nblocks with xcc == blockIdx %% 8: %d of %d\n", match, n);
    return 0;
}

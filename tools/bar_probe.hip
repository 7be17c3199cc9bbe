// tools/bar_probe.hip -- can the CPU write straight into device memory (large BAR) so that a single-query scan needs no
// staging kernel for its 6 KB query?  hipExtMallocWithFlags(hipDeviceMallocFinegrained) and plain hipMalloc + a CPU store.
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <string.h>
__global__ void sum_kernel(const float *p, int n, float *out) { float s = 0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i]; atomicAdd(out, s); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int n = 1536;
    float host[n]; for (int i = 0; i < n; i++) host[i] = 1.0f + i;
    float *out; hipHostMalloc((void **)&out, 64, hipHostMallocDefault);
    int large_bar = -1; (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar: %d\n", large_bar);
    for (int mode = 0; mode < 2; mode++) {
        float *dev = nullptr;
        hipError_t e = mode == 0 ? hipExtMallocWithFlags((void **)&dev, n * 4, hipDeviceMallocFinegrained) : hipMalloc((void **)&dev, n * 4);
        printf("mode %d (%s): alloc %s ptr %p\n", mode, mode == 0 ? "hipDeviceMallocFinegrained" : "hipMalloc", hipGetErrorString(e), (void *)dev);
        if (e != hipSuccess) continue;
        hipPointerAttribute_t at; hipPointerGetAttributes(&at, dev);
        printf("   attributes: type %d, hostPointer %p, devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
        if (mode == 1) { hipFree(dev); continue; }   // a CPU store into plain hipMalloc memory would fault: not tried
        hipMemset(dev, 0, n * 4); hipDeviceSynchronize();
        double t0 = now();
        memcpy(dev, host, n * 4);          // CPU stores into device memory
        __builtin_ia32_sfence();
        double t1 = now();
        *out = 0; hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, dev, n, out); hipDeviceSynchronize();
        printf("   CPU memcpy of 6 KB into it: %.2f us; kernel sum %.1f (expected %.1f)\n", (t1 - t0) * 1e6, *out, n * 1.0 + n * (n - 1) / 2.0);
        // repeat timing
        t0 = now(); for (int r = 0; r < 1000; r++) { host[0] = (float)r; memcpy(dev, host, n * 4); __builtin_ia32_sfence(); } t1 = now();
        printf("   1000 x memcpy + sfence: %.2f us each\n", (t1 - t0) * 1e3);
        // coherence across launches: each round the CPU rewrites the buffer and a kernel (which also read the previous
        // contents, so its L2 lines are warm) must see the new values
        int stale = 0;
        for (int r = 0; r < 5000; r++) {
            for (int i = 0; i < n; i++) host[i] = (float)((r * 7 + i) & 1023);
            double want = 0; for (int i = 0; i < n; i++) want += host[i];
            memcpy(dev, host, n * 4); __builtin_ia32_sfence();
            *out = 0; hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, dev, n, out); hipDeviceSynchronize();
            if (*out != (float)want) stale++;
        }
        printf("   5000 rewrite+launch rounds: %d stale reads\n", stale);
        hipFree(dev);
    }
    return 0;
}

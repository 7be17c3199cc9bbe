// tools/ipc_probe3.hip -- several SMALL allocations of one process exported while the others live / after they went
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char **argv) {
    size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : 20000;
    void *p[64]; hipIpcMemHandle_t h; int fails = 0;
    for (int i = 0; i < 64; i++) {
        hipMalloc(&p[i], bytes + 512 * (size_t)i);
        hipError_t e = hipIpcGetMemHandle(&h, p[i]);
        if (e != hipSuccess) { (void)hipGetLastError(); fails++; printf("live #%d %p: %s\n", i, p[i], hipGetErrorString(e)); }
    }
    printf("64 live small allocations exported one after the other: %d failed\n", fails);
    fails = 0;
    for (int i = 0; i < 64; i += 2) hipFree(p[i]);
    for (int r = 0; r < 200; r++) {
        void *q; hipMalloc(&q, bytes + 64 * (size_t)(r % 7));
        hipError_t e = hipIpcGetMemHandle(&h, q);
        if (e != hipSuccess) { (void)hipGetLastError(); if (!fails) printf("refill #%d %p: %s\n", r, q, hipGetErrorString(e)); fails++; }
        if (r % 3) hipFree(q);
    }
    printf("200 more into the holes: %d failed\n", fails);
    return 0;
}

// tools/ipc_probe.hip -- which hipMalloc sizes hipIpcGetMemHandle exports on this box (round 6: the 20 KB mirror of
// tests/c/ext_driver.c's "DROP INDEX" phase got "invalid argument" on every try, 3 MB mirrors never did).
// usage: ipc_probe [cycles]   -- every size, alone and with other small allocations alive, from a fresh process
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
static int probe(size_t bytes, int cycles, int noise) {
    void *live[64]; int nl = 0;
    for (int i = 0; i < noise; i++) if (hipMalloc(&live[nl], 1000 + 3000 * (size_t)i) == hipSuccess) nl++;
    int fails = 0; void *first = nullptr;
    for (int i = 0; i < cycles; i++) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc(%zu) failed\n", bytes); return -1; }
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipGetLastError(); fails++; if (!first) first = p; }
        (void)hipFree(p);
    }
    for (int i = 0; i < nl; i++) (void)hipFree(live[i]);
    printf("size %9zu  noise %2d: %d of %d exports failed%s\n", bytes, noise, fails, cycles, fails ? "  <-- " : "");
    return fails;
}
int main(int argc, char **argv) {
    int cycles = argc > 1 ? atoi(argv[1]) : 20;
    const size_t sizes[] = {256, 4096, 16384, 20000, 32768, 65536, 100000, 262144, 1 << 20, (2 << 20) - 4096, 2 << 20, 3 << 20};
    for (int noise = 0; noise <= 40; noise += 20)
        for (size_t s : sizes) probe(s, cycles, noise);
    return 0;
}

// stream_patterns.hip -- what does the row-streaming side of mfma_scan_kernel cost on MI355X, by access pattern?
//
// A stand-alone micro-benchmark (not part of the library): persistent workgroups pull "tasks" of 128 rows x
// row_bytes out of one big buffer and bring them into LDS stage by stage, exactly like the scan kernel's slice
// loop (wait, barrier, issue the next stage), but score nothing.  Variants differ in how a stage is cut out of
// the task and in how it reaches LDS:
//   mode 0  LDS-DMA, a stage = 128 rows x SLICE bytes (SLICE = 128 / 256 / 512): the K-sliced layout an MFMA tile needs
//   mode 1  LDS-DMA, a stage = the next STAGE bytes of the task, linear (what tile_scan_kernel does)
//   mode 2  global_load_dwordx4 -> registers -> ds_write_b128, K-sliced like mode 0
//   mode 3  global_load_dwordx4 -> registers -> ds_write_b128, linear like mode 1
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_patterns tools/stream_patterns.hip && /tmp/stream_patterns
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 128;  // rows per task

// THREADS threads; stage = kRows x SLICE bytes (mode 0/2) or kRows * SLICE linear bytes (mode 1/3); NBUF buffers
template <int MODE, int SLICE, int THREADS, int NBUF>
__global__ __launch_bounds__(THREADS) void stream_kernel(const char *__restrict__ rows, int ntasks, int row_bytes,
                                                         int *__restrict__ counter, unsigned *__restrict__ sink,
                                                         int lds_pad) {
    constexpr int STAGE = kRows * SLICE;
    constexpr int NW = THREADS / 64;
    constexpr int NINSTR = STAGE / 1024;        // 1 KB per wavefront instruction
    constexpr int PER_WAVE = NINSTR / NW;
    static_assert(NINSTR % NW == 0, "stage must split evenly");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF * STAGE (+ pad to steer occupancy)
    __shared__ int lds_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nstages = row_bytes / SLICE;
    unsigned acc = 0;
    if (threadIdx.x == 0) lds_task = atomicAdd(counter, 1);
    __syncthreads();
    for (;;) {
        const int t = lds_task;
        if (t >= ntasks) break;
        int next = 0;
        if (threadIdx.x == 0) next = atomicAdd(counter, 1);
        const char *base = rows + (size_t)t * kRows * row_bytes;
        constexpr int LPR = SLICE / 16;       // lanes per row piece (mode 0/2)
        constexpr int RPI = 64 / LPR;         // rows per instruction
        auto src_of = [&](int s, int j) -> const char * {
            const int i = wave + NW * j;  // instruction of the stage
            if (MODE == 0 || MODE == 2) {
                const int row = i * RPI + lane / LPR;
                return base + (size_t)row * row_bytes + (size_t)s * SLICE + (size_t)(lane % LPR) * 16;
            }
            return base + (size_t)s * STAGE + (size_t)i * 1024 + (size_t)lane * 16;
        };
        u32x4 regs[PER_WAVE];
        auto issue = [&](int s, int buf) {
#pragma unroll
            for (int j = 0; j < PER_WAVE; j++) {
                const char *p = src_of(s, j);
                if (MODE < 2) {
                    char *dst = smem + (size_t)buf * STAGE + (size_t)(wave + NW * j) * 1024;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                } else {
                    regs[j] = *reinterpret_cast<const u32x4 *>(p);
                }
            }
        };
        auto land = [&](int buf) {  // mode 2/3: registers -> LDS
            if (MODE >= 2) {
#pragma unroll
                for (int j = 0; j < PER_WAVE; j++)
                    *reinterpret_cast<u32x4 *>(smem + (size_t)buf * STAGE + (size_t)(wave + NW * j) * 1024 + lane * 16) =
                        regs[j];
            }
        };
#pragma unroll
        for (int s0 = 0; s0 < NBUF - 1; s0++)
            if (s0 < nstages) {
                issue(s0, s0);
                if (MODE >= 2 && NBUF > 2) land(s0);
            }
        for (int s = 0; s < nstages; s++) {
            if (MODE >= 2) {
                // one stage of registers in flight: it lands in LDS now, the next one is requested
                if (NBUF == 2) land(s & 1);
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                if (s + 1 < nstages) issue(s + 1, (s + 1) & 1);
            } else {
                if (NBUF > 2 && s + NBUF - 2 < nstages)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PER_WAVE) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (s + NBUF - 1 < nstages) issue(s + NBUF - 1, (s + NBUF - 1) % NBUF);
            }
            // touch the stage like a consumer would (one 16-byte LDS read per lane)
            unsigned v;
            const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem +
                               (unsigned)((s % NBUF) * STAGE) + (unsigned)threadIdx.x * 16u % (unsigned)STAGE;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc ^= v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (threadIdx.x == 0) lds_task = next;
        __syncthreads();
    }
    if (acc == 0x12345678u && lds_pad == -1) sink[0] = acc;
}

template <int MODE, int SLICE, int THREADS, int NBUF>
void run(const char *name, const char *rows, size_t bytes, int row_bytes, int wg_per_cu, int num_cus, int *counter,
         unsigned *sink) {
    const int ntasks = (int)(bytes / ((size_t)kRows * row_bytes));
    const size_t stage = (size_t)kRows * SLICE;
    size_t lds = (size_t)NBUF * stage;
    // pad the dynamic LDS so that exactly wg_per_cu workgroups fit a CU (160 KB)
    const size_t want = (160 * 1024) / wg_per_cu - 512;
    if (lds > want) {
        printf("%-58s skipped (needs %zu B LDS, %d/CU allows %zu)\n", name, lds, wg_per_cu, want);
        return;
    }
    const size_t lds_total = want;
    auto k = stream_kernel<MODE, SLICE, THREADS, NBUF>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(counter, 0, sizeof(int)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(num_cus * wg_per_cu), dim3(THREADS), lds_total, 0, rows, ntasks, row_bytes, counter,
                           sink, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double gb = (double)ntasks * kRows * row_bytes / 1e9;
    printf("%-58s %d WG/CU x %d thr, %d buf x %3zu KB: %6.3f ms  %7.1f GB/s\n", name, wg_per_cu, THREADS, NBUF, stage / 1024,
           best, gb / (best / 1e3));
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int num_cus = prop.multiProcessorCount;
    const int row_bytes = 6144;
    const size_t bytes = (size_t)6 << 30;  // 6 GiB: well past L2 + MALL
    char *rows;
    int *counter;
    unsigned *sink;
    CK(hipMalloc(&rows, bytes));
    CK(hipMemset(rows, 1, bytes));
    CK(hipMalloc(&counter, 256));
    CK(hipMalloc(&sink, 256));
    printf("%s, %d CUs, %zu GiB streamed per launch, rows of %d bytes, tasks of %d rows\n", prop.name, num_cus,
           bytes >> 30, row_bytes, kRows);
#define RUN(MODE, SLICE, THREADS, NBUF, WG) \
    run<MODE, SLICE, THREADS, NBUF>("mode " #MODE " slice " #SLICE, rows, bytes, row_bytes, WG, num_cus, counter, sink)
    // K-sliced DMA (what mfma_scan_kernel does: 128-byte slices, 256 threads, 2 buffers)
    RUN(0, 128, 256, 2, 3);
    RUN(0, 128, 256, 2, 4);
    RUN(0, 128, 256, 3, 3);
    RUN(0, 128, 256, 4, 2);
    RUN(0, 128, 512, 2, 2);
    RUN(0, 256, 256, 2, 2);
    RUN(0, 256, 256, 2, 1);
    RUN(0, 256, 512, 2, 2);
    RUN(0, 512, 256, 2, 1);
    RUN(0, 512, 512, 2, 1);
    // linear DMA
    RUN(1, 128, 256, 2, 3);
    RUN(1, 128, 256, 2, 4);
    RUN(1, 256, 256, 2, 2);
    RUN(1, 256, 512, 2, 2);
    RUN(1, 512, 512, 2, 1);
    // through registers
    RUN(2, 128, 256, 2, 3);
    RUN(2, 128, 256, 2, 4);
    RUN(2, 256, 256, 2, 2);
    RUN(2, 256, 512, 2, 2);
    RUN(3, 128, 256, 2, 4);
    RUN(3, 256, 256, 2, 2);
    RUN(3, 256, 512, 2, 2);
    return 0;
}

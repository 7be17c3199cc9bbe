// tools/mfma_numerics.hip -- what ONE matrix instruction does to its K products and the C input, bit for bit.
// Every case is a K-vector a, a K-vector b and a scalar c; the instruction runs with every row of A = a, every column
// of B = b, every element of C = c, so all outputs are the same number: sum_k a_k b_k + c as the hardware rounds it.
// Cases come from a file written by tools/mfma_numerics.py (which also holds the candidate models the results are
// compared with); results go to a file of fp32 bit patterns.  One wavefront per case.
//   usage: mfma_numerics SHAPE cases.bin results.bin     SHAPE: h32 (32x32x16 f16)  h16 (16x16x32 f16)
//                                                               s32 (32x32x2 f32)   s16 (16x16x4 f32)
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// h32: lane l holds k = 8 * (l / 32) .. + 7      h16: lane l holds k = 8 * (l / 16) .. + 7
template <int K, int LANES_PER_K> __global__ void half_kernel(const _Float16 *a, const _Float16 *b, const float *c, float *out) {
    const int cs = blockIdx.x, lane = threadIdx.x;
    const int k0 = 8 * (lane / LANES_PER_K);
    f16x8 av, bv;
    for (int t = 0; t < 8; t++) {
        av[t] = a[(size_t)cs * K + k0 + t];
        bv[t] = b[(size_t)cs * K + k0 + t];
    }
    if constexpr (K == 16) {
        f32x16 acc;
        for (int r = 0; r < 16; r++) acc[r] = c[cs];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
        if (lane == 0) out[cs] = acc[0];
        if (lane == 37) out[gridDim.x + cs] = acc[11];   // another output element: must be the same number
    } else {
        f32x4 acc;
        for (int r = 0; r < 4; r++) acc[r] = c[cs];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
        if (lane == 0) out[cs] = acc[0];
        if (lane == 37) out[gridDim.x + cs] = acc[3];
    }
}
// s32: lane l holds k = l / 32 (K = 2)          s16: lane l holds k = l / 16 (K = 4)
template <int K, int LANES_PER_K> __global__ void single_kernel(const float *a, const float *b, const float *c, float *out) {
    const int cs = blockIdx.x, lane = threadIdx.x;
    const int k = lane / LANES_PER_K;
    const float av = a[(size_t)cs * K + k], bv = b[(size_t)cs * K + k];
    if constexpr (K == 2) {
        f32x16 acc;
        for (int r = 0; r < 16; r++) acc[r] = c[cs];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        if (lane == 0) out[cs] = acc[0];
        if (lane == 37) out[gridDim.x + cs] = acc[11];
    } else {
        f32x4 acc;
        for (int r = 0; r < 4; r++) acc[r] = c[cs];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        if (lane == 0) out[cs] = acc[0];
        if (lane == 37) out[gridDim.x + cs] = acc[3];
    }
}
int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: mfma_numerics h32|h16|s32|s16 cases.bin results.bin\n"); return 2; }
    const bool half = argv[1][0] == 'h';
    const int K = !strcmp(argv[1], "h32") ? 16 : !strcmp(argv[1], "h16") ? 32 : !strcmp(argv[1], "s32") ? 2 : 4;
    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    const size_t es = half ? 2 : 4, per = 2 * K * es + 4;
    const int n = (int)(bytes / per);
    std::vector<char> buf(bytes);
    if (fread(buf.data(), 1, bytes, f) != (size_t)bytes) return 2;
    fclose(f);
    // layout: a[n][K] | b[n][K] | c[n]
    void *da, *db; float *dc, *dout;
    hipMalloc(&da, (size_t)n * K * es); hipMalloc(&db, (size_t)n * K * es); hipMalloc(&dc, 4 * (size_t)n); hipMalloc(&dout, 8 * (size_t)n);
    hipMemcpy(da, buf.data(), (size_t)n * K * es, hipMemcpyHostToDevice);
    hipMemcpy(db, buf.data() + (size_t)n * K * es, (size_t)n * K * es, hipMemcpyHostToDevice);
    hipMemcpy(dc, buf.data() + 2 * (size_t)n * K * es, 4 * (size_t)n, hipMemcpyHostToDevice);
    if (K == 16) hipLaunchKernelGGL((half_kernel<16, 32>), dim3(n), dim3(64), 0, 0, (const _Float16 *)da, (const _Float16 *)db, dc, dout);
    else if (K == 32) hipLaunchKernelGGL((half_kernel<32, 16>), dim3(n), dim3(64), 0, 0, (const _Float16 *)da, (const _Float16 *)db, dc, dout);
    else if (K == 2) hipLaunchKernelGGL((single_kernel<2, 32>), dim3(n), dim3(64), 0, 0, (const float *)da, (const float *)db, dc, dout);
    else hipLaunchKernelGGL((single_kernel<4, 16>), dim3(n), dim3(64), 0, 0, (const float *)da, (const float *)db, dc, dout);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    std::vector<float> out(2 * (size_t)n);
    hipMemcpy(out.data(), dout, 8 * (size_t)n, hipMemcpyDeviceToHost);
    int differ = 0;
    for (int i = 0; i < n; i++) differ += memcmp(&out[i], &out[n + i], 4) != 0;
    f = fopen(argv[3], "wb");
    fwrite(out.data(), 4, n, f);
    fclose(f);
    printf("%s: %d cases, K = %d, %d cases where two output elements differ\n", argv[1], n, K, differ);
    return 0;
}

// kernels_mfma.hip -- rows x centers on the matrix cores: nearest center per row.
//
// The argmin loop of AddTupleToSort (src/ivfbuild.c:183-192) and the assignment half of a
// k-means iteration (src/ivfkmeans.c:391-451 without Elkan's pruning) are the two dense
// contractions of the path: n x k x dim multiply-adds.  kernels_pair.hip runs them on the
// vector ALUs in the reference's per-pair form; this file runs the inner products on MFMA:
//
//   vector   (fp32):  v_mfma_f32_32x32x2_f32  -- exact fp32 FMA chain, 157 TFLOP/s
//   halfvec  (fp16):  v_mfma_f32_32x32x16_f16 -- fp16 products are exact in fp32, fp32 accumulate
//
// * inner-product opclasses (FUNCTION 1 = -sum(a*b), spherical k-means = -clamp(sum(a*b))):
//   the MFMA result IS the reference's arithmetic (a different but equally valid summation
//   order, SURVEY hard part 3); the running lexicographic (value, id) minimum is kept in the
//   epilogue, so "first strictly smaller wins" (src/ivfbuild.c:187-191) holds exactly.
// * L2: sum((a-b)^2) is not a contraction.  |c|^2 - 2 a.c (the |a|^2 term is the same for
//   every center) is computed on MFMA as a PRE-FILTER that keeps the 4 best centers per row.
//   A row whose best pre-filter value is below its second best by more than twice the error
//   bound is decided there and then (the exact computation cannot order them differently);
//   for the others recheck_kernel evaluates the reference's exact sum((a-b)^2) of the 4 kept
//   centers and picks the lexicographic minimum.  The pre-filter's rounding error is at most
//   gamma * (|c|^2 + 2 |a||c|) with gamma = 8 * sqrt(dim + 4) * 2^-24 -- the probabilistic
//   bound of a length-dim fp32 summation (Higham & Mary 2019: lambda * sqrt(n) * u fails with
//   probability ~ exp(-lambda^2 / 2) per sum; lambda = 8), an order of magnitude tighter than
//   the worst-case n * u and still far above any error seen.  pgv_ctx_set_bound(PGV_BOUND_WORST_CASE)
//   replaces it by the deterministic gamma_(dim+1) (|c|^2 + 2 |a||c|) + gamma_(dim+2) d, the second term being the
//   reference's own fp32 rounding of d = sum((a-c)^2), relative to that distance (argmin_bound, pgv_internal.h).
//   A row whose 4th-best pre-filter
//   value is not more than twice that bound above its best (more than 4 centers could be the
//   true minimum) is put on a list and redone by the exact vector-ALU kernel.  Exact ties
//   (duplicate centers, integer-valued data) have a zero gap and therefore always reach the
//   exact kernels and resolve exactly like the reference; should the bound ever be exceeded,
//   the centre chosen differs from the reference's by less than the float tolerance of the
//   distances themselves (a float-level tie, which the reference's own summation order does
//   not pin either).
//
// Tiling: centers are the MFMA M dimension, data rows the N dimension, so a lane's 16
// accumulators of a 32x32 tile are 16 centers for ONE data row and the running minimum is
// lane-local.  A workgroup owns BN data rows and walks all centers in tiles of BM; 128-byte
// slices of BM + BN rows are brought into LDS by the asynchronous global->LDS DMA (next slice
// in flight while the current one is multiplied), stored XOR-swizzled in 16-byte slots so
// that the ds_read_b128 of 32 different rows is bank-conflict free.  Each lane's 16-byte read
// feeds 4 fp32 MFMAs (k-slots are permuted identically for A and B, which a dot product does
// not notice) or one fp16 MFMA.
#include "pgv_device.h"

#include <cfloat>

namespace pgv {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSliceBytes = 128;  // bytes of every row per pipeline stage = 8 slots of 16 B
#ifndef PGV_ARGMIN_AUX
#define PGV_ARGMIN_AUX 0   // cache policy of the assignment kernel's fills (A/B builds: 2 = nt, 16 = sc1, 17 = sc0 sc1)
#endif
constexpr int kCand = 4;          // centers a lane's (and a center part's) list keeps in the L2 pre-filter
constexpr int kWide = 8;          // centers the merged list of a row carries to the recheck (see finish_l2)

// waves along M (centers) x N (rows), 32x32 tiles per wave along M x N
template <typename T> struct MfmaCfg;
template <> struct MfmaCfg<float> {
    static constexpr int WM = 2, WN = 2, TM = 2, TN = 2;  // 128 x 128 per workgroup, 64 accumulators
};
template <> struct MfmaCfg<__half> {
    static constexpr int WM = 2, WN = 4, TM = 4, TN = 2;  // 256 x 256 per workgroup, 128 accumulators
};

// One step of a stage: this lane's 16-byte slot of TM center rows and TN data rows.  In asm,
// loads and wait in one statement (cdna_hip_programming.md 5.7 form i): hipcc cannot prove
// that these reads do not alias the stage the DMA is filling and would drain vmcnt(0) first.
template <int TM, int TN> struct OperandRead;
template <> struct OperandRead<2, 2> {
    static __device__ __forceinline__ void run(unsigned aa, unsigned ba, u32x4 (&a)[2], u32x4 (&b)[2]) {
        asm volatile(
            "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"
            "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:4096\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(b[0]), "=&v"(b[1])
            : "v"(aa), "v"(ba)
            : "memory");
    }
};
template <> struct OperandRead<4, 2> {
    static __device__ __forceinline__ void run(unsigned aa, unsigned ba, u32x4 (&a)[4], u32x4 (&b)[2]) {
        asm volatile(
            "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:4096\n\t"
            "ds_read_b128 %2, %6 offset:8192\n\tds_read_b128 %3, %6 offset:12288\n\t"
            "ds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:4096\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1])
            : "v"(aa), "v"(ba)
            : "memory");
    }
};

// The same reads WITHOUT the wait, and the wait on its own: the slice loop of the 4 x 2 form issues step c + 1's
// six reads before step c's eight MFMAs and waits (counted) only for step c's.  The "+v" operands of the wait tie
// every later use of the registers to it (an MFMA is register-only: a "memory" clobber alone does not order it).
struct OperandPipe42 {
    static __device__ __forceinline__ void issue(unsigned aa, unsigned ba, u32x4 (&a)[4], u32x4 (&b)[2]) {
        asm volatile(
            "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:4096\n\t"
            "ds_read_b128 %2, %6 offset:8192\n\tds_read_b128 %3, %6 offset:12288\n\t"
            "ds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:4096"
            : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1])
            : "v"(aa), "v"(ba)
            : "memory");
    }
    template <int PENDING> static __device__ __forceinline__ void wait(u32x4 (&a)[4], u32x4 (&b)[2]) {
        if constexpr (PENDING == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1])::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1])::"memory");
        __builtin_amdgcn_sched_barrier(0);  // hipcc would hoist the register-only MFMAs above the wait otherwise
    }
};

template <typename T> struct Mma;
template <> struct Mma<float> {
    // 4 k-slots per 16-byte operand: one 32x32x2 MFMA per float
    static __device__ __forceinline__ void run(f32x16 &acc, const u32x4 &a, const u32x4 &b) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
    }
};
template <> struct Mma<__half> {
    static __device__ __forceinline__ void run(f32x16 &acc, const u32x4 &a, const u32x4 &b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0,
                                                     0, 0);
    }
};

// The list scan's forms: FOUR independent accumulators per output (chain e gets every 4th k-group), summed as
// (c0 + c1) + (c2 + c3) in the epilogue.  Each chain adds up dim / 4 products, so the worst-case rounding error of the
// dot product is gamma_(dim/4 + 2) sum |a_i b_i| instead of gamma_dim -- which is what makes the DETERMINISTIC
// completeness bound of the L2 scan (scan_bound, pgv_internal.h) as tight as the statistical one was; the
// chains are also independent MFMA issue streams (no dependent-issue stalls).
template <typename T> struct Mma4;
template <> struct Mma4<float> {
    static __device__ __forceinline__ void run(f32x16 (&acc)[4], const u32x4 &a, const u32x4 &b, int) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc[e], 0, 0, 0);
    }
};
template <> struct Mma4<__half> {
    static __device__ __forceinline__ void run(f32x16 (&acc)[4], const u32x4 &a, const u32x4 &b, int c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[c],
                                                        0, 0, 0);
    }
};

// 16-wide forms: one A fragment against two B fragments (two independent accumulators hide the
// 40-cycle dependent-issue latency of the 16x16 shapes)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Mma16;
template <> struct Mma16<float> {
    static __device__ __forceinline__ void run(f32x4 &c0, f32x4 &c1, const u32x4 &a, const u32x4 &b0, const u32x4 &b1) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b0[e]), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b1[e]), c1, 0, 0, 0);
        }
    }
};
template <> struct Mma16<__half> {
    static __device__ __forceinline__ void run(f32x4 &c0, f32x4 &c1, const u32x4 &a, const u32x4 &b0, const u32x4 &b1) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b0), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b1), c1, 0, 0, 0);
    }
};

// ... and their four-chain variants for the list scan (chain = k-slot of the operand for fp32; read c + 2 x slice
// parity for fp16, whose 16-wide shape takes a whole 16-byte operand per instruction)
template <typename T> struct Mma16x4;
template <> struct Mma16x4<float> {
    static __device__ __forceinline__ void run(f32x4 (&c0)[4], f32x4 (&c1)[4], const u32x4 &a, const u32x4 &b0,
                                               const u32x4 &b1, int) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c0[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b0[e]), c0[e], 0, 0, 0);
            c1[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b1[e]), c1[e], 0, 0, 0);
        }
    }
};
template <> struct Mma16x4<__half> {
    static __device__ __forceinline__ void run(f32x4 (&c0)[4], f32x4 (&c1)[4], const u32x4 &a, const u32x4 &b0,
                                               const u32x4 &b1, int chain) {
        c0[chain] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b0),
                                                           c0[chain], 0, 0, 0);
        c1[chain] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b1),
                                                           c1[chain], 0, 0, 0);
    }
};

// sorted insertion into a lane's best-NC list; strict '<' keeps the earlier (lower) id first
// among equal values -- ids arrive in ascending order within a lane.  NaN and +inf never enter
// (not < DBL_MAX, src/ivfbuild.c:187).  `dropped` (L2 pre-filter only) is the smallest value this list ever turned away
// or pushed out: a row whose `dropped` is out of the error band's reach of its best has ALL its in-band centers in the
// lists, however many the lists hold.
template <int NC> __device__ __forceinline__ void keep_best(float (&tv)[NC], int (&ti)[NC], float &dropped, float s, int id) {
    if (!(s < tv[NC - 1])) {
        if constexpr (NC > 1) dropped = fminf(dropped, s);
        return;
    }
    if constexpr (NC > 1) dropped = fminf(dropped, tv[NC - 1]);
#pragma unroll
    for (int p = NC - 1; p >= 0; p--) {
        const bool shift = p > 0 && s < tv[p - 1];
        if (shift) {
            tv[p] = tv[p - 1];
            ti[p] = ti[p - 1];
        } else {
            tv[p] = s;
            ti[p] = id;
            break;
        }
    }
}

// the pre-filter's error band seen from a row's best value: twice the expansion's own error + the exact form's rounding,
// which is RELATIVE to the distance it rounds (all its terms are positive) -- the distance of the value `v` that is asked
// about, v + |a|^2; every center with a larger value is further still.  Monotone: once a value is out, all larger are.
__device__ __forceinline__ bool out_of_band(float v, float v0, float x2, float cm2, float gamma, float gamma_x) {
    if (!(v < INFINITY)) return true;  // (padding of a short list)
    const float cross = 2.f * sqrtf(x2 * cm2);
    return v - v0 > 2.f * gamma * (cm2 + cross) + 2.02f * gamma_x * fabsf(v + x2);
}

struct L2Lists {  // device buffers of the L2 pipeline
    const unsigned *cmax2 = nullptr;
    float gamma = 0.f, gamma_x = 0.f;
    int *u_count = nullptr;
    int32_t *u_rows = nullptr, *u_cand = nullptr;   // [n], [n x kWide]
    float *u_val = nullptr, *u_drop = nullptr, *u_x2 = nullptr;  // [n x kWide], [n], [n]
};

// inner-product modes: a row's best (value, id) is the answer
__device__ __forceinline__ void finish_ip(int64_t r, float v, int id, int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
    out_idx[r] = id;
    if (out_val) out_val[r] = v == INFINITY ? FLT_MAX : v;
}

// L2 pre-filter: a row's kWide smallest pre-filter values (ascending by (value, id)) and `dropped`, the smallest value
// that is in none of the lists the kWide came from.  The answer when the runner-up is out of the error band's reach of
// the best; else an entry of the recheck list, which evaluates the reference's exact form for the in-band candidates and
// sends the row on to the all-centers exact kernel only if the band holds more than the lists do (the kWide-th value or
// `dropped` inside it).
__device__ __forceinline__ void finish_l2(int64_t r, const float (&sv)[kWide], const int (&sid)[kWide], float dropped, float x2,
                                          int32_t *__restrict__ out_idx, const L2Lists &l2) {
    const float cm2 = __uint_as_float(*l2.cmax2);
    if (sv[0] < INFINITY && out_of_band(sv[1], sv[0], x2, cm2, l2.gamma, l2.gamma_x)) {
        out_idx[r] = sid[0];
        return;
    }
    const int p = atomicAdd(l2.u_count, 1);
    l2.u_rows[p] = (int32_t)r;
#pragma unroll
    for (int c = 0; c < kWide; c++) {
        l2.u_cand[(size_t)p * kWide + c] = sid[c];
        l2.u_val[(size_t)p * kWide + c] = sv[c];
    }
    l2.u_drop[p] = dropped;
    l2.u_x2[p] = x2;
}

// MODE 0: L2 pre-filter (bias - 2 ip, kCand kept)   1: -ip   3: -clamp(ip, -1, 1) (spherical k-means:
// same argmin as acos(ip)/pi including the ties its clamp creates, src/vector.c:703-722)
template <typename T, int MODE>
__global__ __launch_bounds__(MfmaCfg<T>::WM *MfmaCfg<T>::WN * 64, 2) void mfma_argmin_kernel(
    const char *__restrict__ rows, int64_t n, const char *__restrict__ centers, int k, int nvec,
    const float *__restrict__ bias, const char *__restrict__ zeros16, int32_t *__restrict__ out_idx,
    float *__restrict__ out_val, const L2Lists l2,
    int nparts, int part_tiles, float *__restrict__ part_val, int32_t *__restrict__ part_idx, float *__restrict__ part_x2,
    float *__restrict__ part_drop) {
    using C = MfmaCfg<T>;
    constexpr int TM = C::TM, TN = C::TN;
    constexpr int BM = C::WM * TM * 32, BN = C::WN * TN * 32;
    constexpr int STAGE = (BM + BN) * kSliceBytes;
    constexpr int NC = MODE == 0 ? kCand : 1;
    constexpr int NSRC = C::WM * 2;  // lists per data row before the final merge: waves along M x half-waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bias_lds = reinterpret_cast<float *>(smem + 2 * STAGE);

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned sw = (unsigned)(l31 >> 1) & 7u;
    // Which (row tile, center range) this workgroup owns.  nparts == 1: all centers for row tile blockIdx.x.
    // nparts > 1 (XCD-aware blocking, see launch_mfma_t): workgroup ids go round the 8 XCDs, so XCD x sees ids
    // x, x + 8, ...; its s-th workgroup takes center part s % nparts of its (s / nparts)-th row tile.  The ~32
    // workgroups an XCD runs at a time then share a few row tiles AND walk the same center tiles: both operands
    // are L2 hits for all but the first reader, instead of every row tile being re-read from MALL / HBM once per
    // center tile.
    int64_t row_tile = blockIdx.x;
    int part = 0;
    if (nparts > 1) {
        const int64_t w = blockIdx.x, x = w & 7, sq = w >> 3;
        part = (int)(sq % nparts);
        row_tile = x + 8 * (sq / nparts);
    }
    const int64_t row_base = row_tile * BN;
    if (row_base >= n) return;  // (whole workgroup: the padded tail of the XCD round-robin)
    const int cb_first = part * part_tiles * BM;
    const int cb_end = nparts > 1 ? (cb_first + part_tiles * BM < k ? cb_first + part_tiles * BM : k) : k;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int nslices = (nvec + 7) / 8;

    // DMA role of this wavefront: 64 consecutive rows of [BM center rows | BN data rows], 8 per
    // instruction; a lane brings slot (lane & 7) of row (lane >> 3) of the group
    const int crow0 = wave * 64;
    const bool fills_a = crow0 < BM;
    const int drow = lane >> 3, dpos = lane & 7;
    const char *src_base = fills_a ? centers : rows + (size_t)row_base * row_bytes;
    const int src_limit = fills_a ? k : (int)(n - row_base < BN ? n - row_base : BN);
    auto src_row = [&](int cb, int j) {  // row (relative to src_base) instruction j reads; past the end: the last row
        const int i = (fills_a ? cb + crow0 : crow0 - BM) + 8 * j + drow;
        return i < src_limit ? i : src_limit - 1;
    };
    auto issue_stage = [&](int cb, int sl, int buf, int j0 = 0, int j1 = 8) {
#pragma unroll
        for (int j = j0; j < j1; j++) {
            // slot position p of row i holds 16-byte vector p ^ ((i >> 1) & 7) of the slice
            const int v = dpos ^ ((4 * j + (drow >> 1)) & 7);
            const int vi = sl * 8 + v;
            const char *p = vi < nvec ? src_base + (size_t)src_row(cb, j) * row_bytes + (size_t)vi * sizeof(Raw16) : zeros16;
            char *dst = smem + (size_t)buf * STAGE + (size_t)(crow0 + 8 * j) * kSliceBytes;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, PGV_ARGMIN_AUX);
        }
    };

    float tv[TN][NC];
    int ti[TN][NC];
    float dm[TN];  // L2: the smallest value this lane's list of a row turned away (keep_best)
#pragma unroll
    for (int t = 0; t < TN; t++) {
        dm[t] = INFINITY;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            tv[t][c] = INFINITY;
            ti[t][c] = 0;
        }
    }

    float xx[TN];  // L2 only: this lane's share of |row|^2 (its half-wave's k-slots), gathered during the first center tile
#pragma unroll
    for (int t = 0; t < TN; t++) xx[t] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned a_lane = (unsigned)(wm * TM * 32 + l31) * kSliceBytes;
    const unsigned b_lane = (unsigned)(BM + wn * TN * 32 + l31) * kSliceBytes;

    for (int cb = cb_first; cb < cb_end; cb += BM) {
        f32x16 acc[TM][TN];
        if constexpr (MODE == 0) {
            // L2: the accumulators START at -|c|^2 / 2, so that -2 acc = |c|^2 - 2 a.c comes out of the matrix pipeline
            // itself (the epilogue is a multiplication by -2, no bias reads, no fma; the rounding bound of the chain is
            // unchanged: gamma x (|c|^2 / 2 + |a||c|) on acc)
            for (int i = threadIdx.x; i < BM; i += blockDim.x) bias_lds[i] = cb + i < k ? -0.5f * bias[cb + i] : 0.f;
            __syncthreads();
#pragma unroll
            for (int a = 0; a < TM; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float nb = bias_lds[wm * TM * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                    for (int b = 0; b < TN; b++) acc[a][b][r] = nb;
                }
        } else {
#pragma unroll
            for (int a = 0; a < TM; a++)
#pragma unroll
                for (int b = 0; b < TN; b++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
        }

        issue_stage(cb, 0, 0);
        for (int sl = 0; sl < nslices; sl++) {
            // slice sl has landed for every wavefront and nobody still reads the other buffer.  The
            // wait is spelled out: __syncthreads() alone does NOT drain an LDS-DMA (hipcc emits only
            // lgkmcnt(0) before the barrier here; with the short fp16 stages the reads then overtook
            // the fill)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            constexpr bool kPipe = TM == 4 && TN == 2;  // fp16 (its L2 form takes |row|^2 from row_norms_kernel: part_x2)
            const bool more = sl + 1 < nslices;
            if (more) issue_stage(cb, sl + 1, (sl + 1) & 1);  // all eight pieces at once: spreading them behind the MFMA groups
                                                                // was 9 % SLOWER (28.2 -> 30.9 ms) -- the fill is what the barrier waits for
            const unsigned sbase = lds0 + (unsigned)(sl & 1) * STAGE;
            if constexpr (kPipe) {
                // two operand sets: step c + 1's reads are in flight while step c multiplies (one exposed LDS latency
                // per slice instead of four)
                u32x4 a0[4], b0[2], a1[4], b1[2];
                const unsigned x0 = (((unsigned)(0 + half)) ^ sw) << 4, x1 = (((unsigned)(2 + half)) ^ sw) << 4,
                               x2 = (((unsigned)(4 + half)) ^ sw) << 4, x3 = (((unsigned)(6 + half)) ^ sw) << 4;
                OperandPipe42::issue(sbase + a_lane + x0, sbase + b_lane + x0, a0, b0);
                OperandPipe42::issue(sbase + a_lane + x1, sbase + b_lane + x1, a1, b1);
                OperandPipe42::wait<6>(a0, b0);
#pragma unroll
                for (int tm = 0; tm < 4; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) Mma<T>::run(acc[tm][tn], a0[tm], b0[tn]);
                __builtin_amdgcn_sched_barrier(0);
                OperandPipe42::issue(sbase + a_lane + x2, sbase + b_lane + x2, a0, b0);
                OperandPipe42::wait<6>(a1, b1);
#pragma unroll
                for (int tm = 0; tm < 4; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) Mma<T>::run(acc[tm][tn], a1[tm], b1[tn]);
                __builtin_amdgcn_sched_barrier(0);
                OperandPipe42::issue(sbase + a_lane + x3, sbase + b_lane + x3, a1, b1);
                OperandPipe42::wait<6>(a0, b0);
#pragma unroll
                for (int tm = 0; tm < 4; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) Mma<T>::run(acc[tm][tn], a0[tm], b0[tn]);
                __builtin_amdgcn_sched_barrier(0);
                OperandPipe42::wait<0>(a1, b1);
#pragma unroll
                for (int tm = 0; tm < 4; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) Mma<T>::run(acc[tm][tn], a1[tm], b1[tn]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const unsigned x = (((unsigned)(2 * c + half)) ^ sw) << 4;
                u32x4 a[TM], b[TN];
                OperandRead<TM, TN>::run(sbase + a_lane + x, sbase + b_lane + x, a, b);
                if (MODE == 0 && nparts == 1 && cb == 0) {  // (split over center parts: |row|^2 comes from row_norms_kernel)
#pragma unroll
                    for (int tn = 0; tn < TN; tn++) {
                        Raw16 raw;
#pragma unroll
                        for (int w = 0; w < 4; w++) raw.w[w] = b[tn][w];
                        xx[tn] = accum_slice<T, 1>(xx[tn], raw, raw);
                    }
                }
#pragma unroll
                for (int tm = 0; tm < TM; tm++)
#pragma unroll
                    for (int tn = 0; tn < TN; tn++) Mma<T>::run(acc[tm][tn], a[tm], b[tn]);
            }
        }

        // fold this tile of centers into the lane-local running minima
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int tm = 0; tm < TM; tm++) {
                __builtin_amdgcn_sched_barrier(0);  // one tile at a time: keeps the epilogue's live values few
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int cl = wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int cid = cb + cl;
                    float s = acc[tm][tn][r];
                    if constexpr (MODE == 0) {
                        s = -2.f * s;
                    } else if constexpr (MODE == 3) {
                        s = s > 1.f ? 1.f : (s < -1.f ? -1.f : s);
                        s = -s;
                    } else {
                        s = -s;
                    }
                    if (cid < k) keep_best<NC>(tv[tn], ti[tn], dm[tn], s, cid);
                }
            }
        __syncthreads();  // the last slice and the bias tile are free to be overwritten
    }

    // merge the NSRC lists of every data row (LDS reuses the stage memory)
    constexpr int W = NC > 1 ? kWide : 1;   // what a row's merged list holds when this workgroup saw every center
    float *mv = reinterpret_cast<float *>(smem);
    int *mi = reinterpret_cast<int *>(smem + sizeof(float) * BN * NSRC * NC);
    float *mx = reinterpret_cast<float *>(smem + (sizeof(float) + sizeof(int)) * BN * NSRC * NC);  // [BN][2]
    float *md = mx + BN * 2;                                                                        // [BN][NSRC]
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const int j = wn * TN * 32 + tn * 32 + l31;
        const int s = wm * 2 + half;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            mv[(j * NSRC + s) * NC + c] = tv[tn][c];
            mi[(j * NSRC + s) * NC + c] = ti[tn][c];
        }
        if (MODE == 0 && wm == 0) mx[j * 2 + half] = xx[tn];
        if (MODE == 0) md[j * NSRC + s] = dm[tn];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < BN; j += blockDim.x) {
        const int64_t r = row_base + j;
        if (r >= n) continue;
        const float *v = mv + j * NSRC * NC;
        const int *id = mi + j * NSRC * NC;
        // rounds of lexicographic (value, id) minimum over the entries not taken yet: NC when the parts of a row are
        // merged later (argmin_merge_kernel), W when this is the whole row; one more round gives the smallest entry left
        const int rounds = nparts > 1 ? NC : W;
        float sv[W];
        int sid[W];
        float last_v = -INFINITY;
        int last_id = -1;
        bool first = true;
        float dropped = INFINITY;
        for (int c = 0; c <= rounds; c++) {
            float bv = INFINITY;
            int bid = 0x7fffffff;
            for (int e = 0; e < NSRC * NC; e++) {
                const float ev = v[e];
                const int eid = id[e];
                const bool after = first || ev > last_v || (ev == last_v && eid > last_id);
                if (after && (ev < bv || (ev == bv && eid < bid))) {
                    bv = ev;
                    bid = eid;
                }
            }
            if (bid == 0x7fffffff) {  // nothing left (fewer finite candidates than asked for)
                bv = INFINITY;
                bid = 0;
            }
            if (c == rounds) {
                dropped = bv;
                break;
            }
#pragma unroll
            for (int w = 0; w < W; w++)
                if (w == c) {
                    sv[w] = bv;
                    sid[w] = bid;
                }
            if (bv == INFINITY) {  // the rest is padding
#pragma unroll
                for (int w = 0; w < W; w++)
                    if (w > c) {
                        sv[w] = INFINITY;
                        sid[w] = 0;
                    }
                break;
            }
            last_v = bv;
            last_id = bid;
            first = false;
        }
        if constexpr (MODE == 0)
            for (int s = 0; s < NSRC; s++) dropped = fminf(dropped, md[j * NSRC + s]);
        if (nparts > 1) {
            // this part's best NC of its centers: argmin_merge_kernel folds the parts of a row
#pragma unroll
            for (int c = 0; c < NC; c++) {
                part_val[((size_t)r * nparts + part) * NC + c] = sv[c];
                part_idx[((size_t)r * nparts + part) * NC + c] = sid[c];
            }
            if constexpr (MODE == 0) part_drop[(size_t)r * nparts + part] = dropped;
            continue;
        }
        if constexpr (NC > 1) {
            const float x2 = sizeof(T) == 2 ? part_x2[r] : mx[j * 2] + mx[j * 2 + 1];
            finish_l2(r, sv, sid, dropped, x2, out_idx, l2);
        } else {
            finish_ip(r, sv[0], sid[0], out_idx, out_val);
        }
    }
}

// the parts of every row (nparts x NC candidates each, ascending by (value, id)) -> the row's NC best -> finish_row.
// G = nparts rounded up to a power of two lanes per row: a lane holds one part's list in registers (coalesced
// 32-byte loads), the row's merge is NC rounds of "smallest head of the G lists" by shuffles; the owner of the
// smallest head advances.  (One thread per row walking the parts' lists in memory was 17 ms for 1 M rows: every
// load instruction touched 64 cache lines.)
template <int NC>
__global__ __launch_bounds__(256) void argmin_merge_kernel(int64_t n, int nparts, int lg, const float *__restrict__ part_val,
                                                           const int32_t *__restrict__ part_idx,
                                                           const float *__restrict__ part_x2,
                                                           const float *__restrict__ part_drop, int32_t *__restrict__ out_idx,
                                                           float *__restrict__ out_val, const L2Lists l2) {
    constexpr int W = NC > 1 ? kWide : 1;
    const int G = 1 << lg;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> lg;
    const int p = (int)(t & (G - 1));
    const bool live = r < n && p < nparts;
    unsigned long long list[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        list[c] = ~0ull;
        if (live) {
            const size_t at = ((size_t)r * nparts + p) * NC + c;
            const float v = part_val[at];
            // +inf entries are a part's padding (fewer finite candidates than NC): never candidates
            if (v < INFINITY) list[c] = ((unsigned long long)float_to_key(v) << 32) | (unsigned)part_idx[at];
        }
    }
    float sv[W];
    int sid[W];
#pragma unroll
    for (int c = 0; c < W; c++) {
        unsigned long long best = list[0];
        for (int m = 1; m < G; m <<= 1) {
            const unsigned long long o = __shfl_xor(best, m);
            best = o < best ? o : best;
        }
        if (best == ~0ull) {
            sv[c] = INFINITY;
            sid[c] = 0;
        } else {
            sv[c] = key_to_float((unsigned)(best >> 32));
            sid[c] = (int)(unsigned)(best & 0xffffffffu);
        }
        if (list[0] == best && best != ~0ull) {  // ids are distinct: exactly one lane owns the smallest head
#pragma unroll
            for (int j = 0; j + 1 < NC; j++) list[j] = list[j + 1];
            list[NC - 1] = ~0ull;
        }
    }
    if constexpr (NC > 1) {
        // the smallest value in none of the W: what a part turned away, or a part's head that is still waiting
        float dropped = live ? part_drop[(size_t)r * nparts + p] : INFINITY;
        if (list[0] != ~0ull) dropped = fminf(dropped, key_to_float((unsigned)(list[0] >> 32)));
        for (int m = 1; m < G; m <<= 1) dropped = fminf(dropped, __shfl_xor(dropped, m));
        if (r < n && p == 0) finish_l2(r, sv, sid, dropped, part_x2[r], out_idx, l2);
    } else {
        if (r < n && p == 0) finish_ip(r, sv[0], sid[0], out_idx, out_val);
    }
}

// bias[c] = |c|^2 in fp32; *cmax2 = max_c |c|^2 (as ordered uint bits; a NaN norm ends up
// above every number and sends all rows to the exact kernel)
template <typename T>
__global__ __launch_bounds__(256) void center_norms_kernel(const char *__restrict__ centers, int k, int nvec,
                                                           float *__restrict__ bias, unsigned *__restrict__ cmax2) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= k) return;
    const int lane = threadIdx.x & (kWave - 1);
    const char *row = centers + (size_t)c * nvec * sizeof(Raw16);
    float acc = 0.f;
    for (int v = lane; v < nvec; v += kWave) {
        const Raw16 x = load16(row + (size_t)v * sizeof(Raw16));
        acc = accum_slice<T, 1>(acc, x, x);
    }
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) {
        bias[c] = acc;
        atomicMax(cmax2, __float_as_uint(acc));
    }
}

// The reference's exact sum((a-b)^2) for the in-band candidates (<= kWide) of the rows the pre-filter left undecided
// (one wavefront per listed row), lexicographic minimum.  A row whose band holds more than the lists do -- its kWide-th
// value, or the smallest value the lists turned away, is not clear of the best -- goes on to the redo list.
template <typename T>
__global__ __launch_bounds__(256) void recheck_kernel(const char *__restrict__ rows, const char *__restrict__ centers,
                                                      int k, int nvec, const L2Lists l2, int32_t *__restrict__ out_idx,
                                                      int *__restrict__ fb_count,
                                                      int32_t *__restrict__ fb_rows,
                                                      unsigned long long *__restrict__ packed) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= *l2.u_count) return;
    const int64_t r = l2.u_rows[p];
    const int lane = threadIdx.x & (kWave - 1);
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const char *row = rows + (size_t)r * row_bytes;
    const float cm2 = __uint_as_float(*l2.cmax2), x2 = l2.u_x2[p];
    const float v0 = l2.u_val[p * kWide];
    // the candidates inside the band: a prefix of the ascending list (at least the best two)
    int ncheck = 2;
    while (ncheck < kWide && !out_of_band(l2.u_val[p * kWide + ncheck], v0, x2, cm2, l2.gamma, l2.gamma_x)) ncheck++;
    float bv = INFINITY;
    int bid = 0x7fffffff;
    for (int c0 = 0; c0 < ncheck; c0 += kCand) {   // (wavefront-uniform)
        int id[kCand];
        const char *cp[kCand];
#pragma unroll
        for (int c = 0; c < kCand; c++) {
            id[c] = l2.u_cand[p * kWide + (c0 + c < ncheck ? c0 + c : c0)];
            cp[c] = centers + (size_t)id[c] * row_bytes;
        }
        float d[kCand] = {0.f, 0.f, 0.f, 0.f};
        for (int v = lane; v < nvec; v += kWave) {
            const Raw16 x = load16(row + (size_t)v * sizeof(Raw16));
#pragma unroll
            for (int c = 0; c < kCand; c++) d[c] = accum_slice<T, 0>(d[c], x, load16(cp[c] + (size_t)v * sizeof(Raw16)));
        }
        for (int m = 32; m > 0; m >>= 1) {
#pragma unroll
            for (int c = 0; c < kCand; c++) d[c] += __shfl_xor(d[c], m);
        }
#pragma unroll
        for (int c = 0; c < kCand; c++)
            if (c0 + c < ncheck && l2.u_val[p * kWide + c0 + c] < INFINITY &&
                (d[c] < bv || (d[c] == bv && d[c] < INFINITY && id[c] < bid))) {
                bv = d[c];
                bid = id[c];
            }
    }
    if (lane != 0) return;
    out_idx[r] = bid == 0x7fffffff ? 0 : bid;
    if (k > kWide) {
        // complete when the first value NOT evaluated is out of the band: the next of the list, and what the lists dropped
        const bool list_ok = ncheck < kWide;   // (ncheck == kWide: the last of the list is inside -- there may be more)
        if (!(list_ok && out_of_band(l2.u_drop[p], v0, x2, cm2, l2.gamma, l2.gamma_x))) {
            packed[r] = ~0ull;
            fb_rows[atomicAdd(fb_count, 1)] = (int32_t)r;
        }
    }
}

// the redo list's rows: (distance key, center id) minima gathered by the chunked exact kernel -> list ids
__global__ __launch_bounds__(256) void redo_finish_kernel(const int *__restrict__ fb_count,
                                                          const int32_t *__restrict__ fb_rows,
                                                          const unsigned long long *__restrict__ packed,
                                                          int32_t *__restrict__ out_idx) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= *fb_count) return;
    const int64_t r = fb_rows[p];
    const unsigned long long e = packed[r];
    out_idx[r] = e == ~0ull ? 0 : (int32_t)(unsigned)(e & 0xffffffffu);  // nothing < DBL_MAX: list 0 (src/ivfbuild.c:183-192)
}

// FUNCTION 1 distance of every row to the center chosen for it (the value AddTupleToSort's loop
// ends with), one wavefront per row
template <typename T>
__global__ __launch_bounds__(256) void chosen_distance_kernel(const char *__restrict__ rows, int64_t n,
                                                              const char *__restrict__ centers, int nvec,
                                                              const int32_t *__restrict__ idx,
                                                              float *__restrict__ out_val) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & (kWave - 1);
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const char *row = rows + (size_t)r * row_bytes;
    const char *cp = centers + (size_t)idx[r] * row_bytes;
    float d0 = 0.f, d1 = 0.f;
    int v = lane;
    for (; v + kWave < nvec; v += 2 * kWave) {
        const Raw16 x0 = load16(row + (size_t)v * sizeof(Raw16)), x1 = load16(row + (size_t)(v + kWave) * sizeof(Raw16));
        const Raw16 c0 = load16(cp + (size_t)v * sizeof(Raw16)), c1 = load16(cp + (size_t)(v + kWave) * sizeof(Raw16));
        d0 = accum_slice<T, 0>(d0, x0, c0);
        d1 = accum_slice<T, 0>(d1, x1, c1);
    }
    if (v < nvec) d0 = accum_slice<T, 0>(d0, load16(row + (size_t)v * sizeof(Raw16)), load16(cp + (size_t)v * sizeof(Raw16)));
    float d = d0 + d1;
    for (int m = 32; m > 0; m >>= 1) d += __shfl_xor(d, m);
    if (lane == 0) out_val[r] = d < INFINITY ? d : FLT_MAX;
}

// profiling only: rows redone by the exact kernel, rows assigned, rows rechecked -- accumulated on the device
__global__ void add_count_kernel(const int *u_count, const int *fb_count, double *acc, double rows) {
    acc[0] += (double)*fb_count;
    acc[1] += rows;
    acc[2] += (double)*u_count;
}

template <typename T, int MODE>
int launch_mfma_t(pgv_ctx *ctx, const RowGeom &g, const void *rows, int64_t n, const void *centers, int k,
                  const float *bias, int32_t *out_idx, float *out_val, const L2Lists &l2 = L2Lists()) {
    using C = MfmaCfg<T>;
    constexpr int BM = C::WM * C::TM * 32, BN = C::WN * C::TN * 32;
    constexpr int threads = C::WM * C::WN * 64;
    constexpr int NC = MODE == 0 ? kCand : 1;
    const size_t lds = 2 * (size_t)(BM + BN) * kSliceBytes + sizeof(float) * BM;
    auto kern = mfma_argmin_kernel<T, MODE>;
    PGV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t row_tiles = (n + BN - 1) / BN;
    // XCD-aware blocking: one center tile per workgroup when there are several.  Every row tile is then read once
    // per XCD round instead of once per center tile (the center tiles are shared by the workgroups of the round),
    // and a short n (a Lloyd iteration over 50 k samples) still fills the device.  PGV_ARGMIN_PART_TILES overrides
    // the center tiles per workgroup (0: all of them, the one-workgroup-per-row-tile form).
    static const int env_tiles = [] {
        const char *e = getenv("PGV_ARGMIN_PART_TILES");
        return e ? atoi(e) : -1;
    }();
    const int center_tiles = (k + BM - 1) / BM;
    // measured (profiles/r03_assign_blocking.md): the split wins 22 % for fp16 inner product (1.25 M x 4096 x 3072:
    // 35.6 -> 29.1 ms), 3-6 % for fp16 L2, fp32 inner product and short n; the fp32 L2 pre-filter at 1 M x 1000 is
    // compute-bound already (70 % of the fp32 MFMA peak) and loses 6-10 % to the extra norm pass and the merge
    const bool split = MODE != 0 || sizeof(T) == 2 || row_tiles < 2 * (int64_t)ctx->num_cus;
    // center tiles per workgroup when split (measured, 1.25 M x 4096 x 3072 fp16: L2 pre-filter 743 / 791 / 809 TF at
    // 1 / 2 / 4 tiles -- fewer epilogues and parts to merge --, inner product 1092 / 1124 / 1113; fp32 indifferent;
    // a short n wants the most workgroups: 1)
    const int auto_tiles = row_tiles < 2 * (int64_t)ctx->num_cus ? 1 : (sizeof(T) == 2 ? (MODE == 0 ? 4 : 2) : 1);
    int part_tiles = env_tiles < 0 ? (split ? auto_tiles : center_tiles) : (env_tiles == 0 ? center_tiles : env_tiles);
    int nparts = (center_tiles + part_tiles - 1) / part_tiles;
    if (nparts > 64) {
        part_tiles = (center_tiles + 63) / 64;
        nparts = (center_tiles + part_tiles - 1) / part_tiles;
    }
    float *part_val = nullptr, *part_x2 = nullptr, *part_drop = nullptr;
    int32_t *part_idx = nullptr;
    int64_t grid = row_tiles;
    const bool norms_pass = MODE == 0 && (nparts > 1 || sizeof(T) == 2);  // (the fp16 kernel never gathers |row|^2 itself)
    if (nparts > 1 || norms_pass) {
        const size_t pv = sizeof(float) * (size_t)n * nparts * NC;
        const size_t pd = sizeof(float) * (size_t)n * nparts;
        PGV_TRY(ctx->mf_d.ensure(2 * pv + pd + sizeof(float) * (size_t)n + 64));
        part_val = ctx->mf_d.as<float>();
        part_idx = reinterpret_cast<int32_t *>(ctx->mf_d.as<char>() + pv);
        part_drop = reinterpret_cast<float *>(ctx->mf_d.as<char>() + 2 * pv);
        part_x2 = reinterpret_cast<float *>(ctx->mf_d.as<char>() + 2 * pv + pd);
        if (nparts > 1) grid = 8 * ((row_tiles + 7) / 8) * nparts;
    }
    if (grid > 0x7fffffff) PGV_FAIL(PGV_ERR_ARG, "assignment: too many workgroups");
    if (norms_pass) {
        // |row|^2 for the error bound of the pre-filter: one streaming pass (the one-workgroup-per-row-tile form gathers
        // it from the operands it reads anyway; here 1 / nparts of the workgroups would have to, and wait for it)
        unsigned *max_bits = reinterpret_cast<unsigned *>(part_x2 + n);
        PGV_HIP(hipMemsetAsync(max_bits, 0, sizeof(unsigned), ctx->stream));
        PGV_TRY(launch_row_norms(ctx, sizeof(T) == 4 ? PGV_F32 : PGV_F16, g, rows, n, part_x2, max_bits));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, ctx->stream, static_cast<const char *>(rows), n,
                       static_cast<const char *>(centers), k, g.nvec, bias, static_cast<const char *>(ctx->zeros.p),
                       out_idx, out_val, l2, nparts, part_tiles, part_val, part_idx, part_x2, part_drop);
    PGV_HIP(hipGetLastError());
    if (nparts > 1) {
        int lg = 0;
        while ((1 << lg) < nparts) lg++;
        const int64_t threads_total = n << lg;
        hipLaunchKernelGGL(argmin_merge_kernel<NC>, dim3((unsigned)((threads_total + 255) / 256)), dim3(256), 0, ctx->stream, n,
                           nparts, lg, part_val, part_idx, part_x2, part_drop, out_idx, out_val, l2);
        PGV_HIP(hipGetLastError());
    }
    return PGV_OK;
}

template <typename T>
int launch_mfma_mode(pgv_ctx *ctx, int mode, const RowGeom &g, const void *rows, int64_t n, const void *centers, int k,
                     int32_t *out_idx, float *out_val) {
    if (mode == 1) return launch_mfma_t<T, 1>(ctx, g, rows, n, centers, k, nullptr, out_idx, out_val);
    if (mode == 3) return launch_mfma_t<T, 3>(ctx, g, rows, n, centers, k, nullptr, out_idx, out_val);

    // L2: norms -> MFMA pre-filter (decides most rows) -> exact recheck of the undecided -> exact redo of
    // the still ambiguous (chunked over the centers) -> distances to the chosen centers if asked for.
    // mf_a: bias[k] | cmax2 | u_count | fb_count      mf_b: u_cand[n x kWide] | u_val[n x kWide] | u_drop[n] | u_x2[n]
    // mf_c: u_rows[n] | fb_rows[n] | packed[n] (uint64)
    const pgv_dtype dtype = sizeof(T) == 4 ? PGV_F32 : PGV_F16;
    PGV_TRY(ctx->mf_a.ensure(sizeof(float) * (size_t)k + 64));
    PGV_TRY(ctx->mf_b.ensure((sizeof(int32_t) + sizeof(float)) * (size_t)n * kWide + 2 * sizeof(float) * (size_t)n));
    PGV_TRY(ctx->mf_c.ensure((2 * sizeof(int32_t) + sizeof(unsigned long long)) * (size_t)n + 16));
    float *bias = ctx->mf_a.as<float>();
    unsigned *cmax2 = reinterpret_cast<unsigned *>(bias + k);
    int *u_count = reinterpret_cast<int *>(cmax2 + 1);
    int *fb_count = u_count + 1;
    int32_t *u_cand = ctx->mf_b.as<int32_t>();
    float *u_val = reinterpret_cast<float *>(u_cand + (size_t)n * kWide);
    unsigned long long *packed = ctx->mf_c.as<unsigned long long>();
    int32_t *u_rows = reinterpret_cast<int32_t *>(packed + n);
    int32_t *fb_rows = u_rows + n;
    PGV_HIP(hipMemsetAsync(cmax2, 0, 3 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(center_norms_kernel<T>, dim3((k + 3) / 4), dim3(256), 0, ctx->stream,
                       static_cast<const char *>(centers), k, g.nvec, bias, cmax2);
    L2Lists l2;
    l2.cmax2 = cmax2;
    const ExpansionBound eb = argmin_bound(ctx, g.ld);  // 8 sqrt(dim + 4) 2^-24 (see the header), or the worst case
    l2.gamma = eb.gamma;
    l2.gamma_x = eb.gamma_exact;
    l2.u_count = u_count;
    l2.u_rows = u_rows;
    l2.u_cand = u_cand;
    l2.u_val = u_val;
    l2.u_drop = u_val + (size_t)n * kWide;
    l2.u_x2 = l2.u_drop + n;
    PGV_TRY((launch_mfma_t<T, 0>(ctx, g, rows, n, centers, k, bias, out_idx, nullptr, l2)));
    // the lists' lengths stay on the device: grids cover the worst case, surplus workgroups leave at once
    hipLaunchKernelGGL(recheck_kernel<T>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream,
                       static_cast<const char *>(rows), static_cast<const char *>(centers), k, g.nvec, l2, out_idx, fb_count,
                       fb_rows, packed);
    PGV_HIP(hipGetLastError());
    if (k > kWide) {
        PGV_TRY(launch_argmin_listed(ctx, 0, dtype, g, rows, n, centers, k, fb_rows, fb_count, packed));
        hipLaunchKernelGGL(redo_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, fb_count,
                           fb_rows, packed, out_idx);
        PGV_HIP(hipGetLastError());
    }
    if (out_val) {
        hipLaunchKernelGGL(chosen_distance_kernel<T>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream,
                           static_cast<const char *>(rows), n, static_cast<const char *>(centers), g.nvec, out_idx,
                           out_val);
        PGV_HIP(hipGetLastError());
    }
    if (ctx->profiling && ctx->stats_dev.p) {
        hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, ctx->stream, u_count, fb_count,
                           ctx->stats_dev.as<double>() + 2, (double)n);
        PGV_HIP(hipGetLastError());
    }
    return PGV_OK;
}

// =====================================================================================
// The batched list scan on the matrix cores.
//
// tile_scan_kernel scores a row against 16 queries with sub + fma per element on the vector ALUs
// and is bound by their issue rate, not by HBM.  A task here is <= 128 rows of
// one list x <= 32 queries probing it: queries are the MFMA M dimension, rows the N dimension,
// both K-streamed through LDS in 128-byte slices by the same DMA + swizzle as above, one 32x32
// tile (32 queries x 32 rows) per wavefront.  A lane ends up with 16 queries' values for ONE row,
// and the 32 lanes of a half-wave hold 32 consecutive rows of one query: 128-byte stores into
// the queries' output segments.
//   METRIC 1 (negative inner product): the value is the reference's arithmetic.
//   METRIC 0 (L2): |x|^2 - 2 q.x with the rows' precomputed norms (no |q|^2: it shifts all values of a query alike,
//   so the selections run without it) -- an APPROXIMATION of
//   sum((q-x)^2) (cancellation), used only to pick candidates; pgv_abi_ivf.hip's scan_batch_dev
//   re-evaluates the exact form for the k' best and checks that nothing outside them can matter.
constexpr int kScanQueries = 32;   // queries per task

constexpr int kScanWaves = 4;      // one 32 x 32 tile each: a task is 128 rows

// NT: the list rows are fetched with the non-temporal policy (an index far larger than the caches is read once per
// batch: +3.5 % on the 6 GB headline index, 1.145 -> 1.102 ms per batch, same box, alternating runs); the query rows,
// which every task of a group re-reads, keep the default policy
// NQ: queries per task, 32 or 64.  64 (round 5) is for batches whose lists are probed by more than ~12 queries on average
// (configs[2] / [4] at their real shape: 1024 queries x 64 probes over 4096 lists): a list probed by 33 .. 64 queries is
// streamed ONCE instead of twice -- a wavefront then holds two 32 x 32 tiles (queries 0-31 and 32-63 against its 32 rows).
// NS: LDS stages, 2 or 3.  3 (round 6) is for launches with fewer tasks than workgroup slots -- the batch's center ranking
// (1000 centers x 1024 queries = 256 tasks), small batches: one workgroup per CU cannot hide a stage fill (an L2 / MALL
// round trip, ~1 us) behind the others' streaming, so the 32-query form keeps TWO fills in flight (slice sl + 2 is issued
// while slice sl is multiplied, the wait is vmcnt(5): this wavefront's five DMA instructions of the younger fill may
// still be out).  60 KB of LDS: two workgroups per CU, which such a launch does not have anyway.
template <typename T, int METRIC, int NW, bool NT, int NQ, int NS = 2>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NS == 3 ? 2 : 3, NS == 3 ? 2 : 3))) void mfma_scan_kernel(
    const char *__restrict__ rows, const char *__restrict__ queries, const ScanTask *__restrict__ tasks,
    const int *__restrict__ ntasks_ptr, int *__restrict__ task_counter, const ScanPair *__restrict__ pairs,
    const float *__restrict__ row_norms, int nvec, const char *__restrict__ zeros16, float *__restrict__ out) {
    constexpr int ROWS = 32 * NW;
    constexpr int NGROUPS = (NQ + ROWS) / 8;          // DMA instructions per stage (8 rows each)
    constexpr int NDMA = (NGROUPS + NW - 1) / NW;               // ... per wavefront, at most
    constexpr int STAGE = (NQ + ROWS) * kSliceBytes;  // 20 KB (NQ 32) / 24 KB (NQ 64)
    static_assert(NS == 2 || (NS == 3 && NQ == 32 && NGROUPS % NW == 0), "three stages: the 32-query form only");
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
    __shared__ int64_t pair_rel[NQ];
    __shared__ int lds_task;

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned sw = (unsigned)(l31 >> 1) & 7u;
    const size_t row_bytes = (size_t)nvec * sizeof(Raw16);
    const int nslices = (nvec + 7) / 8;
    const int ntasks = *ntasks_ptr;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    // DMA: groups of 8 rows of [32 query rows | ROWS list rows]; wavefront w issues groups w, w + NW, ...;
    // a lane brings slot (lane & 7) of row (lane >> 3) of its group
    const int drow = lane >> 3, dpos = lane & 7;

    if (threadIdx.x == 0) lds_task = atomicAdd(task_counter, 1);
    __syncthreads();
    for (;;) {
        const int t = lds_task;
        if (t >= ntasks) {
            // the last workgroup out leaves the two words zero for the next launch (no memset in front of it)
            if (threadIdx.x == 0 && atomicAdd(task_counter + 1, 1) == (int)gridDim.x - 1) {
                task_counter[0] = 0;
                task_counter[1] = 0;
            }
            return;
        }
        // the task after this one is claimed now: its round trip runs under this task's streaming
        int next = 0;
        if (threadIdx.x == 0) next = atomicAdd(task_counter, 1);
        const ScanTask task = tasks[t];
        const int np = task.npairs;
        if ((int)threadIdx.x < NQ) {
            // slots past the task's last query repeat that query
            const ScanPair pr = pairs[task.pair0 + ((int)threadIdx.x < np ? (int)threadIdx.x : np - 1)];
            pair_rel[threadIdx.x] = pr.out_rel + task.row0;
        }
        const char *src[NDMA];
#pragma unroll
        for (int j = 0; j < NDMA; j++) {
            const int crow = (wave + NW * j) * 8 + drow;  // row of the concatenation
            if (crow < NQ) {
                const ScanPair pr = pairs[task.pair0 + (crow < np ? crow : np - 1)];
                src[j] = queries + (size_t)pr.query * row_bytes;
            } else {
                const int r = crow - NQ;
                src[j] = rows + ((size_t)task.row0 + (size_t)(r < task.nrows ? r : task.nrows - 1)) * row_bytes;
            }
        }
        // tasks of <= 16 queries (most of them at ~10 queries per list) take the 16-wide MFMA shapes: half
        // the matrix-core time and half the query bytes through the DMA
        const bool narrow = np <= 16;
        const bool upper = NQ > 32 && np > 32;  // the task's queries 32 .. 63 exist: the second tile of every wavefront
        // rows that another task streams again (a list probed by more than 32 queries of the batch) are fetched with
        // the default policy, so that the later pass can find them in the MALL; everything else past the caches
        // (headline, alternating runs on one box: scan 1.149 / 1.148 -> 1.107 / 1.129 ms)
        const bool past_caches = NT && task.pad == 0;
        auto issue_stage = [&](int sl, int buf) {
#pragma unroll
            for (int j = 0; j < NDMA; j++) {
                const int g8 = wave + NW * j;  // group of 8 rows
                if (NGROUPS % NW != 0 && g8 >= NGROUPS) break;  // (uniform per wavefront)
                if (narrow && g8 >= 2 && g8 < NQ / 8) continue;    // query rows 16 .. NQ - 1: not used
                if (NQ > 32 && !upper && g8 >= 4 && g8 < NQ / 8) continue;  // query rows 32 .. 63: not used
                const int v = dpos ^ ((4 * (g8 & 1) + (drow >> 1)) & 7);  // slot p of row i holds vector p ^ ((i >> 1) & 7)
                const int vi = sl * 8 + v;
                const char *p = vi < nvec ? src[j] + (size_t)vi * sizeof(Raw16) : zeros16;
                char *dst = smem + (size_t)buf * STAGE + (size_t)g8 * 8 * kSliceBytes;
                if (NT && past_caches && g8 >= NQ / 8)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 2);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            }
        };
        // pair table to registers (after the loop: a store followed by an LDS read makes hipcc wait for the store)
        const int j32 = wave * 32 + l31;
        if constexpr (NQ > 32) {
          if (!narrow && upper) {
            // The 64-query form.  Two tiles per wavefront (queries 0-31 and, when the task has them, 32-63, against its 32
            // rows) leave no room for four interleaved accumulators per output (2 x 64 registers: spills at three waves per
            // SIMD): the four chains are the row's slices in four consecutive QUARTERS instead -- one accumulator per tile,
            // closed into a second register set at every quarter's end, like mfma_dense_kernel.  Each chain holds at most
            // ceil(slices / 4) x 32 (fp32) products: the bound of the candidates is scan_bound_chain (pgv_abi_ivf.hip).
            f32x16 accl, accu, suml, sumu;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                accl[r] = 0.f;
                accu[r] = 0.f;
                suml[r] = 0.f;
                sumu[r] = 0.f;
            }
            const int quarter = (nslices + 3) >> 2;
            int in_quarter = 0;
            const unsigned a_lane = (unsigned)l31 * kSliceBytes;                         // query l31 (upper tile: + 32 rows)
            const unsigned b_lane = (unsigned)(NQ + wave * 32 + l31) * kSliceBytes;      // row wave * 32 + l31
            issue_stage(0, 0);
            for (int sl = 0; sl < nslices; sl++) {
                __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                if (sl + 1 < nslices) issue_stage(sl + 1, (sl + 1) & 1);
                const unsigned sbase = lds0 + (unsigned)(sl & 1) * STAGE;
                if (upper) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const unsigned x = (((unsigned)(2 * c + half)) ^ sw) << 4;
                        u32x4 a, au, b;
                        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:4096\n\tds_read_b128 %2, %4\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&v"(a), "=&v"(au), "=&v"(b)
                                     : "v"(sbase + a_lane + x), "v"(sbase + b_lane + x)
                                     : "memory");
                        Mma<T>::run(accl, a, b);
                        Mma<T>::run(accu, au, b);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const unsigned x = (((unsigned)(2 * c + half)) ^ sw) << 4;
                        u32x4 a, b;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(a), "=&v"(b)
                                     : "v"(sbase + a_lane + x), "v"(sbase + b_lane + x)
                                     : "memory");
                        Mma<T>::run(accl, a, b);
                    }
                }
                if (++in_quarter == quarter) {
                    in_quarter = 0;
                    suml += accl;
                    sumu += accu;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        accl[r] = 0.f;
                        accu[r] = 0.f;
                    }
                }
            }
            suml += accl;  // (the open chain; zeros when the last quarter closed)
            sumu += accu;
            const int j32w = wave * 32 + l31;
            if (j32w < task.nrows) {
                const float rn = METRIC == 0 ? row_norms[task.row0 + j32w] : 0.f;
                {
                    int64_t rel[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) rel[r] = pair_rel[(r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                    for (int r = 0; r < 16; r++) out[rel[r] + j32w] = METRIC == 0 ? fmaf(-2.f, suml[r], rn) : -suml[r];
                }
                if (upper) {
                    // (slots past the task's last query repeat that query: same value, same address)
                    int64_t rel[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) rel[r] = pair_rel[32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                    for (int r = 0; r < 16; r++) out[rel[r] + j32w] = METRIC == 0 ? fmaf(-2.f, sumu[r], rn) : -sumu[r];
                }
            }
          }
        }
        // (the 64-query kernel's tasks of 17 .. 32 queries take this form too: the same code as the 32-query kernel's)
        if (!narrow && !upper) {
            f32x16 acc4[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc4[e][r] = 0.f;
            const unsigned a_lane = (unsigned)l31 * kSliceBytes;                                   // query l31
            const unsigned b_lane = (unsigned)(NQ + wave * 32 + l31) * kSliceBytes;      // row wave * 32 + l31
            issue_stage(0, 0);
            if (NS == 3 && nslices > 1) issue_stage(1, 1);
            int buf = 0;  // stage of slice sl (NS == 3: a ring; NS == 2: sl & 1)
            for (int sl = 0; sl < nslices; sl++) {
                // the slice has landed (a bare s_barrier: __syncthreads() does not reliably drain an LDS-DMA).  Three
                // stages: the first slice of a task waits for everything (the previous task's output stores are counted
                // in vmcnt too, and loads and stores need not retire in order); from then on only this wavefront's five
                // DMA instructions of the NEXT slice may still be out
                if (NS == 3 && sl > 0 && sl + 1 < nslices)
                    __builtin_amdgcn_s_waitcnt(0x0075);  // vmcnt(5) lgkmcnt(0)
                else
                    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                if (NS == 3) {
                    if (sl + 2 < nslices) issue_stage(sl + 2, buf == 0 ? 2 : buf - 1);   // (sl + 2) % 3
                } else {
                    if (sl + 1 < nslices) issue_stage(sl + 1, (sl + 1) & 1);
                }
                const unsigned sbase = lds0 + (unsigned)(NS == 3 ? buf : (sl & 1)) * STAGE;
                if (NS == 3) buf = buf == 2 ? 0 : buf + 1;
                if (NS == 3) {
                    // one workgroup per CU: nobody else's MFMAs cover this wavefront's LDS round trips -- step c + 1's two
                    // operand reads go out before step c's MFMAs (two register sets, a counted wait)
                    u32x4 pa[2], pb[2];
                    const unsigned x0 = (((unsigned)half) ^ sw) << 4;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"
                                 : "=&v"(pa[0]), "=&v"(pb[0])
                                 : "v"(sbase + a_lane + x0), "v"(sbase + b_lane + x0)
                                 : "memory");
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (c + 1 < 4) {
                            const unsigned x = (((unsigned)(2 * (c + 1) + half)) ^ sw) << 4;
                            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"
                                         : "=&v"(pa[(c + 1) & 1]), "=&v"(pb[(c + 1) & 1])
                                         : "v"(sbase + a_lane + x), "v"(sbase + b_lane + x)
                                         : "memory");
                            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(pa[c & 1]), "+v"(pb[c & 1])::"memory");
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pa[c & 1]), "+v"(pb[c & 1])::"memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);  // (the register-only MFMAs must stay behind the wait)
                        Mma4<T>::run(acc4, pa[c & 1], pb[c & 1], c);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const unsigned x = (((unsigned)(2 * c + half)) ^ sw) << 4;
                        u32x4 a, b;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(a), "=&v"(b)
                                     : "v"(sbase + a_lane + x), "v"(sbase + b_lane + x)
                                     : "memory");
                        Mma4<T>::run(acc4, a, b, c);
                    }
                }
            }
            const f32x16 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            // lane: row j32 of the task; register r: query (r & 3) + 8 (r >> 2) + 4 half.  Slots past the
            // task's last query hold copies of that query (same operands, same value, same address): they
            // are stored too, so that the 16 stores are one straight run -- a branch per store makes hipcc
            // wait for the previous store each time
            int64_t rel[16];
#pragma unroll
            for (int r = 0; r < 16; r++) rel[r] = pair_rel[(r & 3) + 8 * (r >> 2) + 4 * half];
            if (j32 < task.nrows) {
                const float rn = METRIC == 0 ? row_norms[task.row0 + j32] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++)
                    out[rel[r] + j32] = METRIC == 0 ? fmaf(-2.f, acc[r], rn) : -acc[r];
            }
        }
        if (narrow) {
            // 16 x 16 tiles: lane = (row or query l15, k-group kg); two row halves per wavefront
            const int l15 = lane & 15, kg = lane >> 4;
            const unsigned sw15 = (unsigned)(l15 >> 1) & 7u;
            f32x4 c04[4], c14[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                c04[e] = f32x4{0.f, 0.f, 0.f, 0.f};
                c14[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const unsigned a_lane = (unsigned)l15 * kSliceBytes;
            const unsigned b_lane = (unsigned)(NQ + wave * 32 + l15) * kSliceBytes;  // second half: + 16 rows
            issue_stage(0, 0);
            for (int sl = 0; sl < nslices; sl++) {
                __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                if (sl + 1 < nslices) issue_stage(sl + 1, (sl + 1) & 1);
                const unsigned sbase = lds0 + (unsigned)(sl & 1) * STAGE;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const unsigned x = (((unsigned)(kg + 4 * c)) ^ sw15) << 4;
                    u32x4 a, b0, b1;
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:2048\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(a), "=&v"(b0), "=&v"(b1)
                                 : "v"(sbase + a_lane + x), "v"(sbase + b_lane + x)
                                 : "memory");
                    // (a uniform branch, not an index computed at run time: the accumulators stay in registers)
                    if (sl & 1)
                        Mma16x4<T>::run(c04, c14, a, b0, b1, c + 2);
                    else
                        Mma16x4<T>::run(c04, c14, a, b0, b1, c);
                }
            }
            const f32x4 c0 = (c04[0] + c04[1]) + (c04[2] + c04[3]), c1 = (c14[0] + c14[1]) + (c14[2] + c14[3]);
            // lane: rows wave * 32 + l15 (c0) and + 16 (c1); register r: query 4 kg + r
            int64_t rel[4];
#pragma unroll
            for (int r = 0; r < 4; r++) rel[r] = pair_rel[4 * kg + r];
            const int ja = wave * 32 + l15, jb = ja + 16;
            const float rna = (METRIC == 0 && ja < task.nrows) ? row_norms[task.row0 + ja] : 0.f;
            const float rnb = (METRIC == 0 && jb < task.nrows) ? row_norms[task.row0 + jb] : 0.f;
            if (ja < task.nrows) {
#pragma unroll
                for (int r = 0; r < 4; r++) out[rel[r] + ja] = METRIC == 0 ? fmaf(-2.f, c0[r], rna) : -c0[r];
            }
            if (jb < task.nrows) {
#pragma unroll
                for (int r = 0; r < 4; r++) out[rel[r] + jb] = METRIC == 0 ? fmaf(-2.f, c1[r], rnb) : -c1[r];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the pair table and the task id have been read ...
        __builtin_amdgcn_s_barrier();        // ... by everyone; they and the LDS slices are free again
        if (threadIdx.x == 0) lds_task = next;
        __syncthreads();
    }
}

// |row|^2 in fp32, one wavefront per row (the L2 scan's norms; also the queries' of a batch)
template <typename T>
__global__ __launch_bounds__(256) void row_norms_kernel(const char *__restrict__ rows, int64_t n, int nvec,
                                                        float *__restrict__ out, unsigned *__restrict__ max_bits) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & (kWave - 1);
    const char *row = rows + (size_t)r * nvec * sizeof(Raw16);
    float acc = 0.f;
    for (int v = lane; v < nvec; v += kWave) {
        const Raw16 x = load16(row + (size_t)v * sizeof(Raw16));
        acc = accum_slice<T, 1>(acc, x, x);
    }
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) {
        out[r] = acc;
        // a million atomics on one word take 11 ms; nearly every row is below the maximum seen so far and only looks
        // (NaN bits compare above every number, as wanted)
        if (max_bits && __float_as_uint(acc) > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(max_bits, __float_as_uint(acc));
    }
}

}  // namespace

// the matrix-core path pays from a few dozen centers on (a tile is 128 / 256 centers wide)
bool mfma_argmin_supported(int mode, int64_t n, int k) { return (mode == 0 || mode == 1 || mode == 3) && k >= 64 && n >= 256; }

int launch_argmin_mfma(pgv_ctx *ctx, int mode, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n,
                       const void *centers, int k, int32_t *out_idx, float *out_val) {
    if (n <= 0) return PGV_OK;
    if (!ctx->zeros.p) {
        PGV_TRY(ctx->zeros.ensure(256));
        PGV_HIP(hipMemsetAsync(ctx->zeros.p, 0, 256, ctx->stream));
    }
    if (dtype == PGV_F32) return launch_mfma_mode<float>(ctx, mode, g, rows, n, centers, k, out_idx, out_val);
    return launch_mfma_mode<__half>(ctx, mode, g, rows, n, centers, k, out_idx, out_val);
}

}  // namespace pgv

namespace pgv {

int mfma_scan_rows_per_task() { return 32 * kScanWaves; }
int mfma_scan_queries_per_task() { return kScanQueries; }
// ... or twice that, when the lists of a batch are probed by more than ~12 queries on average (the 64-query form)
int mfma_scan_queries_per_task_wide() { return 2 * kScanQueries; }

int launch_row_norms(pgv_ctx *ctx, pgv_dtype dtype, const RowGeom &g, const void *rows, int64_t n, float *out,
                     unsigned *max_bits) {
    if (n <= 0) return PGV_OK;
    const dim3 grid((unsigned)((n + 3) / 4));
    if (dtype == PGV_F32)
        hipLaunchKernelGGL(row_norms_kernel<float>, grid, dim3(256), 0, ctx->stream, static_cast<const char *>(rows), n,
                           g.nvec, out, max_bits);
    else
        hipLaunchKernelGGL(row_norms_kernel<__half>, grid, dim3(256), 0, ctx->stream, static_cast<const char *>(rows), n,
                           g.nvec, out, max_bits);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int launch_mfma_scan(pgv_ctx *ctx, pgv_metric metric, pgv_dtype dtype, const RowGeom &g, const void *rows,
                     const void *queries, const ScanTask *tasks, const int *ntasks_dev, int ntasks_bound,
                     const ScanPair *pairs, const float *row_norms, const float *query_norms, float *out,
                     bool stream_rows, int queries_per_task) {
    if (ntasks_bound <= 0) return PGV_OK;
    if (queries_per_task != 32 && queries_per_task != 64) PGV_FAIL(PGV_ERR_ARG, "mfma scan: %d queries per task", queries_per_task);
    if (metric != PGV_L2SQ && metric != PGV_NEG_IP) PGV_FAIL(PGV_ERR_ARG, "mfma scan: L2 / inner product only");
    if (!ctx->zeros.p) {
        PGV_TRY(ctx->zeros.ensure(256));
        PGV_HIP(hipMemsetAsync(ctx->zeros.p, 0, 256, ctx->stream));
    }
    PGV_TRY(ctx->counters.ensure(256));
    if (!ctx->counters_clean) {
        PGV_HIP(hipMemsetAsync(ctx->counters.p, 0, 256, ctx->stream));
        ctx->counters_clean = true;
    }
    int *counter = ctx->counters.as<int>() + 8;  // words 8, 9: claimed tasks, workgroups done (the kernel re-zeroes them)
    int grid = ctx->num_cus * 3;  // 41 KB of LDS per workgroup: three per CU
    if (grid > ntasks_bound) grid = ntasks_bound;
    // fewer tasks than two per CU: nothing else streams through a workgroup's fill latency -> three LDS stages
    // (PGV_SCAN_DEEP = 0 switches it off: A/B)
    static const int deep_env = [] {
        const char *e = getenv("PGV_SCAN_DEEP");
        return e ? atoi(e) : -1;
    }();
    const bool deep = queries_per_task == 32 && deep_env != 0 && ntasks_bound <= 2 * ctx->num_cus;
#define PGV_MSCAN_Q(T, M, NT, Q, S)                                                                                     \
    hipLaunchKernelGGL((mfma_scan_kernel<T, M, kScanWaves, NT, Q, S>), dim3(grid), dim3(kScanWaves * 64), 0, ctx->stream, \
                       static_cast<const char *>(rows), static_cast<const char *>(queries), tasks, ntasks_dev, counter, \
                       pairs, row_norms, g.nvec, static_cast<const char *>(ctx->zeros.p), out)
#define PGV_MSCAN_NT(T, M, NT)                   \
    do {                                         \
        if (queries_per_task == 64)              \
            PGV_MSCAN_Q(T, M, NT, 64, 2);         \
        else if (deep)                           \
            PGV_MSCAN_Q(T, M, NT, 32, 3);         \
        else                                     \
            PGV_MSCAN_Q(T, M, NT, 32, 2);         \
    } while (0)
#define PGV_MSCAN(T, M)              \
    do {                             \
        if (stream_rows)             \
            PGV_MSCAN_NT(T, M, true); \
        else                         \
            PGV_MSCAN_NT(T, M, false); \
    } while (0)
    if (dtype == PGV_F32) {
        if (metric == PGV_L2SQ)
            PGV_MSCAN(float, 0);
        else
            PGV_MSCAN(float, 1);
    } else {
        if (metric == PGV_L2SQ)
            PGV_MSCAN(__half, 0);
        else
            PGV_MSCAN(__half, 1);
    }
#undef PGV_MSCAN
#undef PGV_MSCAN_NT
#undef PGV_MSCAN_Q
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

}  // namespace pgv

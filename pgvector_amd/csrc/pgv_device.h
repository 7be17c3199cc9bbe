// pgv_device.h -- device-side helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "pgv_internal.h"

namespace pgv {

constexpr int kWave = 64;  // CDNA wavefront

// ---- 16-byte row vectors -------------------------------------------------
// A row in HBM is a whole number of 16-byte vectors: 4 floats or 8 halves.
template <typename T> struct VecTraits;
template <> struct VecTraits<float> {
    static constexpr int N = 4;
};
template <> struct VecTraits<__half> {
    static constexpr int N = 8;
};

struct alignas(16) Raw16 {
    uint32_t w[4];
};

__device__ __forceinline__ Raw16 raw16_zero() {
    Raw16 r;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0u;
    return r;
}

__device__ __forceinline__ Raw16 load16(const void *p) {
    return *reinterpret_cast<const Raw16 *>(p);
}

// unpack to fp32 lanes (exact for binary16: src/halfutils.h:62-141)
template <typename T> struct Unpacked;
template <> struct Unpacked<float> {
    float v[4];
    __device__ __forceinline__ explicit Unpacked(const Raw16 &r) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = __uint_as_float(r.w[i]);
    }
};
template <> struct Unpacked<__half> {
    float v[8];
    __device__ __forceinline__ explicit Unpacked(const Raw16 &r) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __half2 h = *reinterpret_cast<const __half2 *>(&r.w[i]);
            float2 f = __half22float2(h);
            v[2 * i] = f.x;
            v[2 * i + 1] = f.y;
        }
    }
};

// ---- the reference's per-element updates ----------------------------------
// METRIC 0: sum (a-b)^2 (src/vector.c:566-571)   1: sum a*b (:613-614)   2: sum |a-b| (:731-732)
template <int METRIC> __device__ __forceinline__ float accum(float acc, float a, float b) {
    if constexpr (METRIC == 0) {
        float d = a - b;
        return fmaf(d, d, acc);
    } else if constexpr (METRIC == 1) {
        return fmaf(a, b, acc);
    } else {
        return acc + fabsf(a - b);
    }
}

// Same updates straight from binary16 operands.  Written as fma(ext(a), 1, -ext(b)) /
// fma(ext(a), ext(b), acc) so that hipcc can select v_fma_mix_f32 (fp16 sources converted
// inside the FMA, fp32 result): no separate v_cvt per element.  Bit-identical to
// converting first: a*1 - b and a - b round once, from the same exact value.
template <int METRIC> __device__ __forceinline__ float accum_h(float acc, __half a, __half b) {
    if constexpr (METRIC == 1) {
        return __builtin_fmaf(__half2float(a), __half2float(b), acc);
    } else {
        const float d = __builtin_fmaf(__half2float(a), 1.0f, -__half2float(b));
        if constexpr (METRIC == 0)
            return __builtin_fmaf(d, d, acc);
        else
            return acc + fabsf(d);
    }
}

// one 16-byte slice against another: N products folded into acc
template <typename T, int METRIC> __device__ __forceinline__ float accum_slice(float acc, const Raw16 &a, const Raw16 &b);
template <> __device__ __forceinline__ float accum_slice<float, 0>(float acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc = accum<0>(acc, __uint_as_float(a.w[i]), __uint_as_float(b.w[i]));
    return acc;
}
template <> __device__ __forceinline__ float accum_slice<float, 1>(float acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc = accum<1>(acc, __uint_as_float(a.w[i]), __uint_as_float(b.w[i]));
    return acc;
}
template <> __device__ __forceinline__ float accum_slice<float, 2>(float acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc = accum<2>(acc, __uint_as_float(a.w[i]), __uint_as_float(b.w[i]));
    return acc;
}
// fp16 difference a - b of the low / high halves of two packed words, as ONE v_fma_mix_f32
// (a * 1.0 + (-b), fp16 sources extended inside the FMA): hipcc folds the source-level
// fma(ext(a), 1, -ext(b)) back into two conversions and a subtract, so it is spelled in asm.
__device__ __forceinline__ float half_diff_lo(uint32_t wa, uint32_t wb) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(wa), "v"(wb));
    return d;
}
__device__ __forceinline__ float half_diff_hi(uint32_t wa, uint32_t wb) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(wa), "v"(wb));
    return d;
}
// acc + a * b on the low / high fp16 halves, also one v_fma_mix_f32 (the fp16 x fp16 product is
// exact in fp32, so this is the reference's convert-then-FMA, src/halfutils.c:81-122)
__device__ __forceinline__ float half_fma_lo(uint32_t wa, uint32_t wb, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(wa), "v"(wb), "v"(acc));
    return d;
}
__device__ __forceinline__ float half_fma_hi(uint32_t wa, uint32_t wb, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(wa), "v"(wb), "v"(acc));
    return d;
}
template <> __device__ __forceinline__ float accum_slice<__half, 0>(float acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float d0 = half_diff_lo(a.w[i], b.w[i]);
        acc = __builtin_fmaf(d0, d0, acc);
        const float d1 = half_diff_hi(a.w[i], b.w[i]);
        acc = __builtin_fmaf(d1, d1, acc);
    }
    return acc;
}
template <> __device__ __forceinline__ float accum_slice<__half, 1>(float acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        acc = half_fma_lo(a.w[i], b.w[i], acc);
        acc = half_fma_hi(a.w[i], b.w[i], acc);
    }
    return acc;
}
template <> __device__ __forceinline__ float accum_slice<__half, 2>(float acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        acc += fabsf(half_diff_lo(a.w[i], b.w[i]));
        acc += fabsf(half_diff_hi(a.w[i], b.w[i]));
    }
    return acc;
}

// Two-lane accumulators: the fp32 vector ALU reaches its full rate only through the packed
// instructions (v_pk_add_f32 / v_pk_fma_f32: two elements per lane per issue), so the
// compute-bound kernels keep even and odd elements in the two halves of a float2 and add the
// halves at the end (a different but equally valid summation order, SURVEY hard part 3).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T, int METRIC>
__device__ __forceinline__ void accum_slice2(f32x2 &acc, const Raw16 &a, const Raw16 &b);

#define PGV_SLICE2_F(M)                                                                               \
    template <> __device__ __forceinline__ void accum_slice2<float, M>(f32x2 & acc, const Raw16 &a,  \
                                                                       const Raw16 &b) {             \
        _Pragma("unroll") for (int i = 0; i < 4; i += 2) {                                           \
            const f32x2 va = {__uint_as_float(a.w[i]), __uint_as_float(a.w[i + 1])};                 \
            const f32x2 vb = {__uint_as_float(b.w[i]), __uint_as_float(b.w[i + 1])};                 \
            if (M == 1) {                                                                            \
                acc = __builtin_elementwise_fma(va, vb, acc);                                        \
            } else {                                                                                 \
                const f32x2 d = va - vb;                                                             \
                if (M == 0)                                                                          \
                    acc = __builtin_elementwise_fma(d, d, acc);                                      \
                else                                                                                 \
                    acc += __builtin_elementwise_abs(d);                                             \
            }                                                                                        \
        }                                                                                            \
    }
PGV_SLICE2_F(0)
PGV_SLICE2_F(1)
PGV_SLICE2_F(2)
#undef PGV_SLICE2_F

template <> __device__ __forceinline__ void accum_slice2<__half, 0>(f32x2 &acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f32x2 d = {half_diff_lo(a.w[i], b.w[i]), half_diff_hi(a.w[i], b.w[i])};
        acc = __builtin_elementwise_fma(d, d, acc);
    }
}
template <> __device__ __forceinline__ void accum_slice2<__half, 1>(f32x2 &acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        acc.x = half_fma_lo(a.w[i], b.w[i], acc.x);
        acc.y = half_fma_hi(a.w[i], b.w[i], acc.y);
    }
}
template <> __device__ __forceinline__ void accum_slice2<__half, 2>(f32x2 &acc, const Raw16 &a, const Raw16 &b) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        acc.x += fabsf(half_diff_lo(a.w[i], b.w[i]));
        acc.y += fabsf(half_diff_hi(a.w[i], b.w[i]));
    }
}

// kernel value -> FUNCTION 1 value: negative inner product is negated (src/vector.c:646)
template <int METRIC> __device__ __forceinline__ float finish(float acc) {
    if constexpr (METRIC == 1)
        return -acc;
    else
        return acc;
}

// ---- wave64 DPP reduction --------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// Sum over groups of (1 << lg) adjacent lanes; the total lands in the LAST lane
// of each group (other lanes hold partial sums).
__device__ __forceinline__ float group_sum_to_last(float v, int lg) {
    if (lg >= 1) v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    if (lg >= 2) v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    if (lg >= 3) v += dpp_mov<0x114>(v);         // row_shr:4
    if (lg >= 4) v += dpp_mov<0x118>(v);         // row_shr:8
    if (lg >= 5) v += dpp_mov<0x142, 0xa>(v);    // row_bcast:15 -> rows 1,3
    if (lg >= 6) v += dpp_mov<0x143, 0xc>(v);    // row_bcast:31 -> rows 2,3
    return v;
}

// Sums of TWO per-lane values over the 64 lanes for about the price of one reduction: the first
// butterfly step also folds the two values into one register (even lanes carry a, odd lanes b),
// the row rotations and the gfx950 lane-swap instructions finish both at once.  Even lanes
// return the total of a, odd lanes the total of b.
__device__ __forceinline__ float wave_sum2(float a, float b) {
    const bool odd = __lane_id() & 1u;
    float v = odd ? b : a;
    const float send = odd ? a : b;
    v += dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);    // row_ror:4
    v += dpp_mov<0x128>(v);    // row_ror:8
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

// minimum of an unsigned over the 64 lanes, in every lane (same network as wave_sum2)
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = umin32(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = umin32(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = umin32(v, dpp_u32<0x124>(v));  // row_ror:4
    v = umin32(v, dpp_u32<0x128>(v));  // row_ror:8
    const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = umin32(r16[0], r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return umin32(r32[0], r32[1]);
}

// ---- ordered-uint keys for float selection ---------------------------------
// ascending float order == ascending unsigned order; every NaN maps above +inf,
// like PostgreSQL's float8 ordering puts NaN last.
__device__ __forceinline__ uint32_t float_to_key(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN
    if (u == 0x80000000u) u = 0u;                              // -0 == +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    if (k == 0xffffffffu) return __uint_as_float(0x7fc00000u);
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

}  // namespace pgv

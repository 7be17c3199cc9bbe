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

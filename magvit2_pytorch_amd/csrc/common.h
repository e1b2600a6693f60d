// Shared device helpers for the mv2 HIP kernels (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mv2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define MV2_WAVE 64

// element-type tags: storage type of activations / packed weights
struct BF16 {
    typedef unsigned short T;
    static constexpr int kBytes = 2;
    static constexpr int kVec = 8;   // elements per 16-byte chunk
    static constexpr int kDtype = MV2_BF16;
};
struct F32 {
    typedef float T;
    static constexpr int kBytes = 4;
    static constexpr int kVec = 4;
    static constexpr int kDtype = MV2_F32;
};

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;   // round-to-nearest-even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned short, b);
}

// unpack one 16-byte chunk into fp32 lanes (kVec of them)
template <typename E> struct Chunk;
template <> struct Chunk<BF16> {
    static __device__ __forceinline__ void unpack(const i32x4& v, float* f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned w = (unsigned)v[j];
            f[2 * j] = __uint_as_float(w << 16);
            f[2 * j + 1] = __uint_as_float(w & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ i32x4 pack(const float* f) {
        i32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (int)((unsigned)f2bf(f[2 * j]) | ((unsigned)f2bf(f[2 * j + 1]) << 16));
        return v;
    }
};
template <> struct Chunk<F32> {
    static __device__ __forceinline__ void unpack(const i32x4& v, float* f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __int_as_float(v[j]);
    }
    static __device__ __forceinline__ i32x4 pack(const float* f) {
        i32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __float_as_int(f[j]);
        return v;
    }
};

// scalar element load/store by dtype
template <typename E> __device__ __forceinline__ float ld_elem(const void* p, int64_t i);
template <> __device__ __forceinline__ float ld_elem<BF16>(const void* p, int64_t i) { return bf2f(((const unsigned short*)p)[i]); }
template <> __device__ __forceinline__ float ld_elem<F32>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <typename E> __device__ __forceinline__ void st_elem(void* p, int64_t i, float v);
template <> __device__ __forceinline__ void st_elem<BF16>(void* p, int64_t i, float v) { ((unsigned short*)p)[i] = f2bf(v); }
template <> __device__ __forceinline__ void st_elem<F32>(void* p, int64_t i, float v) { ((float*)p)[i] = v; }

// 4 consecutive elements (8 B bf16 / 16 B fp32), address must be naturally aligned
template <typename E> __device__ __forceinline__ void ld4(const void* p, int64_t i, float* f);
template <> __device__ __forceinline__ void ld4<BF16>(const void* p, int64_t i, float* f) {
    i32x2 v = *(const i32x2*)((const unsigned short*)p + i);
    f[0] = __uint_as_float(((unsigned)v[0]) << 16); f[1] = __uint_as_float(((unsigned)v[0]) & 0xffff0000u);
    f[2] = __uint_as_float(((unsigned)v[1]) << 16); f[3] = __uint_as_float(((unsigned)v[1]) & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld4<F32>(const void* p, int64_t i, float* f) {
    f32x4 v = *(const f32x4*)((const float*)p + i);
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <typename E> __device__ __forceinline__ void st4(void* p, int64_t i, const float* f);
template <> __device__ __forceinline__ void st4<BF16>(void* p, int64_t i, const float* f) {
    i32x2 v;
    v[0] = (int)((unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16));
    v[1] = (int)((unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16));
    *(i32x2*)((unsigned short*)p + i) = v;
}
template <> __device__ __forceinline__ void st4<F32>(void* p, int64_t i, const float* f) {
    f32x4 v = {f[0], f[1], f[2], f[3]};
    *(f32x4*)((float*)p + i) = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float act_elu(float v) { return v > 0.f ? v : expm1f(v); }
__device__ __forceinline__ float act_silu(float v) { return v / (1.f + expf(-v)); }
__device__ __forceinline__ float act_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

// dtype-aware activations: the fp32 (id-parity) mode keeps libm accuracy; the bf16 mode (8 mantissa bits
// on store) uses hardware exp / rcp forms that are ~10x cheaper in the conv epilogues.
template <typename E> struct Act;
template <> struct Act<F32> {
    static __device__ __forceinline__ float elu(float v) { return act_elu(v); }
    static __device__ __forceinline__ float silu(float v) { return act_silu(v); }
    static __device__ __forceinline__ float gelu(float v) { return act_gelu(v); }
};
template <> struct Act<BF16> {
    static __device__ __forceinline__ float elu(float v) { return v > 0.f ? v : __expf(v) - 1.f; }
    // v_rcp_f32 (1 ulp) instead of a division: without fast-math __fdividef compiles to the full IEEE sequence (2 v_div_scale, v_rcp, Newton
    // steps, v_div_fmas, v_div_fixup: ~10 VALU slots per value in an epilogue whose wave has the SIMD to itself)
    static __device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
    // gelu(v) = max(v, 0) - 0.5 r(|v|), r(a) = a erfc(a / sqrt 2): the bump r as a degree-10 polynomial in t = 2 a / 4.5 - 1 on [0, 4.5]
    // (clamped beyond, where r < 1e-5): |error| <= 1.5e-5 ABSOLUTE in float32 Horner form (tools/gelu_bump_fit.py; coefficients < 1 in
    // magnitude, no cancellation): <= 2^-8 relative (half a bf16 ulp) where |gelu| >= 4e-3, and an absolute 1.5e-5 below that - the
    // negative tail comes out as a constant ~ -1.5e-5 for gates below -4.5 where the true value tends to 0 (an absolute, not a
    // relative bound; the bf16 parity gates are anchored to the reference's own bf16 drift, 1e-3 .. 7e-3).  No v_exp / v_rcp (quarter rate): 2 + 12
    // packable operations per value against 2 transcendentals + 17 of the Abramowitz-Stegun erf form this replaces (the GEGLU epilogues
    // are VALU-bound: profiles/r05_findings.md).  The fp32 (id-parity) mode keeps libm's erf.
    static __device__ __forceinline__ float gelu(float v) {
        const float a = fminf(fabsf(v), 4.5f);
        const float t = a * 0.44444444444444444f - 1.f;
        float p = 4.258678854e-02f;
        p = p * t + -2.178250334e-02f;
        p = p * t + -1.763804275e-01f;
        p = p * t + 1.763150563e-01f;
        p = p * t + 2.049072476e-01f;
        p = p * t + -4.347814101e-01f;
        p = p * t + 2.029682756e-01f;
        p = p * t + 1.472158170e-01f;
        p = p * t + -2.465923949e-01f;
        p = p * t + 1.330395059e-01f;
        p = p * t + -2.749713406e-02f;
        return fmaxf(v, 0.f) + p;
    }
};

#define MV2_LAUNCH_CHECK()                                 \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return (int)e_;                \
    } while (0)

// weasal_amd/csrc/ws_bf16.h -- feature-row element types of the gather / pooling / GEMM kernels.
//
// BASELINE config 5 keeps the feature rows (activations, their gradients, the weighted features wf) in HBM as
// bf16 and accumulates in fp32; geometry stays fp32.  The kernels are templated on the row element type T
// (float or bf16_t) and touch rows only through ld4 / st4 (four consecutive channels: 16 bytes of f32, 8 bytes
// of bf16), so the lane <-> channel mapping is the same for both.  f32 -> bf16 is the hardware's round-to-
// nearest-even (v_cvt_pk_bf16_f32; keeps NaN a NaN), bf16 -> f32 is exact.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ws_bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float ws_bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned ws_pack_bf2(float a, float b)
{
    const bf16x2_t p = {(bf16_t)a, (bf16_t)b};
    return *reinterpret_cast<const unsigned*>(&p);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p)
{
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(ws_bf_lo(v.x), ws_bf_hi(v.x), ws_bf_lo(v.y), ws_bf_hi(v.y));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v)
{
    *reinterpret_cast<uint2*>(p) = make_uint2(ws_pack_bf2(v.x, v.y), ws_pack_bf2(v.z, v.w));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return ws_bf_lo(*reinterpret_cast<const unsigned short*>(p)); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)v; }

// rows of T are "vector" rows when four channels can be moved at once
template <typename T> struct ws_row_align { static constexpr unsigned mask = 15u; };
template <> struct ws_row_align<bf16_t> { static constexpr unsigned mask = 7u; };
template <typename T> static inline bool ws_row_aligned(const void* p)
{
    return (reinterpret_cast<uintptr_t>(p) & ws_row_align<T>::mask) == 0;
}

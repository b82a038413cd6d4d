// Storage element type of the activation stacks / finalised output gradients in HBM.
//
// The default mode keeps every stack in fp32.  The bf16-storage mode (rln_set_storage, BASELINE.json configs[1]/[3]: the
// reference reaches mixed precision through Lightning's --precision 16, train.py:100-101) keeps the stacks of the levels
// that carry the traffic (row width >= 40: 97.5 % of the activation elements at 120x160) as bf16 NCHW planes: every kernel
// that reads or writes a stack is a template over the element type of each side.  Arithmetic stays what it is everywhere:
// bf16 values are widened on load (exact), BatchNorm / ReLU / statistics / accumulation run in fp32, results are rounded
// to nearest-even once when they are stored.
//
// Kernel parameter structs keep `const float*` fields for the stacks; with ST_BF16 such a pointer addresses bf16 elements
// and only ever passes through SP<ST>, which does the pointer arithmetic in ELEMENTS of the real type.
#pragma once
#include "common.h"

namespace rln {

enum { ST_F32 = 0, ST_BF16 = 1 };
__host__ __device__ constexpr int st_bytes(int st) { return st == ST_BF16 ? 2 : 4; }

typedef __bf16 st_bf16x2 __attribute__((ext_vector_type(2)));
typedef float st_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// two floats -> packed bf16 pair (round to nearest even, v_cvt_pk_bf16_f32), first value in the low half
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  union { st_bf16x2 h; unsigned u; } c;
  st_f32x2 v = {a, b};
  c.h = __builtin_convertvector(v, st_bf16x2);
  return c.u;
}

template <int ST>
struct SRaw;  // the registers a load of 1 / 2 / 4 consecutive elements occupies before it is widened
// rq = ONE 16-byte load (a full-width request per lane: half-width loads leave the per-CU request queues as the limit
// and buy no bandwidth); it holds QN consecutive elements, q(v, i) widens element i (i a compile-time constant after
// unrolling).
template <>
struct SRaw<ST_F32> {
  typedef float r1;
  typedef float2 r2;
  typedef float4 r4;
  typedef float4 rq;
  static constexpr int QN = 4;
  static __device__ __forceinline__ float q(const rq& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
  static __device__ __forceinline__ float w1(r1 v) { return v; }
  static __device__ __forceinline__ float2 w2(r2 v) { return v; }
  static __device__ __forceinline__ float4 w4(r4 v) { return v; }
};
template <>
struct SRaw<ST_BF16> {
  typedef unsigned short r1;
  typedef unsigned r2;
  typedef uint2 r4;
  typedef uint4 rq;
  static constexpr int QN = 8;
  static __device__ __forceinline__ float q(const rq& v, int i) {
    const unsigned w = (i >> 1) == 0 ? v.x : (i >> 1) == 1 ? v.y : (i >> 1) == 2 ? v.z : v.w;
    return (i & 1) ? bf16_hi(w) : bf16_lo(w);
  }
  static __device__ __forceinline__ float w1(r1 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
  static __device__ __forceinline__ float2 w2(r2 v) { return make_float2(bf16_lo(v), bf16_hi(v)); }
  static __device__ __forceinline__ float4 w4(r4 v) {
    return make_float4(bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y));
  }
};

// pointer to storage elements; + counts elements
template <int ST>
struct SP {
  typedef SRaw<ST> R;
  static constexpr int ES = ST == ST_BF16 ? 2 : 4;
  const unsigned char* p;
  __device__ __forceinline__ SP() : p(nullptr) {}
  __device__ __forceinline__ explicit SP(const void* q) : p(reinterpret_cast<const unsigned char*>(q)) {}
  __device__ __forceinline__ SP operator+(long long n) const {
    SP r;
    r.p = p + n * ES;
    return r;
  }
  // raw loads (keep the narrow registers until the value is needed) ...
  __device__ __forceinline__ typename R::r1 raw1(long long i = 0) const {
    return *reinterpret_cast<const typename R::r1*>(p + i * ES);
  }
  __device__ __forceinline__ typename R::r2 raw2(long long i = 0) const {
    return *reinterpret_cast<const typename R::r2*>(p + i * ES);
  }
  __device__ __forceinline__ typename R::r4 raw4(long long i = 0) const {
    return *reinterpret_cast<const typename R::r4*>(p + i * ES);
  }
  __device__ __forceinline__ typename R::rq rawq(long long i = 0) const {  // 16 bytes = R::QN elements
    return *reinterpret_cast<const typename R::rq*>(p + i * ES);
  }
  // ... or widened at once
  __device__ __forceinline__ float ld1(long long i = 0) const { return R::w1(raw1(i)); }
  __device__ __forceinline__ float2 ld2(long long i = 0) const { return R::w2(raw2(i)); }
  __device__ __forceinline__ float4 ld4(long long i = 0) const { return R::w4(raw4(i)); }
  // stores (the pointer is const-qualified for convenience only: output views go through the same type)
  __device__ __forceinline__ void st1(long long i, float v) const {
    if constexpr (ST == ST_BF16) {
      *reinterpret_cast<unsigned short*>(const_cast<unsigned char*>(p) + i * ES) = (unsigned short)(pack_bf16x2(v, 0.f) & 0xffffu);
    } else {
      *reinterpret_cast<float*>(const_cast<unsigned char*>(p) + i * ES) = v;
    }
  }
  __device__ __forceinline__ void st2(long long i, float a, float b) const {
    if constexpr (ST == ST_BF16) {
      *reinterpret_cast<unsigned*>(const_cast<unsigned char*>(p) + i * ES) = pack_bf16x2(a, b);
    } else {
      *reinterpret_cast<float2*>(const_cast<unsigned char*>(p) + i * ES) = make_float2(a, b);
    }
  }
  __device__ __forceinline__ void st4(long long i, float a, float b, float c, float d) const {
    if constexpr (ST == ST_BF16) {
      *reinterpret_cast<uint2*>(const_cast<unsigned char*>(p) + i * ES) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
    } else {
      *reinterpret_cast<float4*>(const_cast<unsigned char*>(p) + i * ES) = make_float4(a, b, c, d);
    }
  }
};

// what a kernel that stores ST elements sees of its own output when it also accumulates statistics of it: the value as
// the consumers will read it (the rounded one), so that the batch statistics describe the stored tensor exactly
template <int ST>
__device__ __forceinline__ float st_round(float v) {
  if constexpr (ST == ST_BF16) return bf16_lo(pack_bf16x2(v, 0.f) & 0xffffu);
  else return v;
}

// host side: byte address of element `idx` of a stack that stores `st` elements
inline const float* st_at(const void* base, long long idx, int st) {
  return reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(base) + idx * st_bytes(st));
}
inline float* st_at(void* base, long long idx, int st) {
  return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(base) + idx * st_bytes(st));
}

}  // namespace rln

// Device-side helpers shared by the 16-bit decoder kernels (conv_bf16.hip, resblock16.hip):
// storage conversions (bfloat16 / IEEE half <-> f32, round to nearest even) and the 32x32x16 MFMA.
#pragma once
#include <hip/hip_runtime.h>

namespace wetts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// 16-bit storage type selected at compile time: F16 = IEEE half, else bfloat16
template <bool F16>
__device__ __forceinline__ float cv_in(unsigned short h) {
  if (F16) return (float)__builtin_bit_cast(_Float16, h);
  return bf2f(h);
}
template <bool F16>
__device__ __forceinline__ unsigned short cv_out(float f) {
  if (F16) return __builtin_bit_cast(unsigned short, (_Float16)f);
  return f2bf(f);
}
// two f32 -> one packed dword (a in the low half); bf16 uses v_cvt_pk_bf16_f32 (RNE)
template <bool F16>
__device__ __forceinline__ unsigned pk2(float a, float b) {
  if (F16) {
    // the values must exist as rounded f32 before the conversion: without this the compiler may
    // fold the producing add/mul into v_fma_mixlo_f16 in one kernel and not in another, and the
    // fused / unfused decoder paths would stop being bit-identical
    asm volatile("" : "+v"(a), "+v"(b));
    // gfx950 has the packed conversion for IEEE half too (v_cvt_pk_f16_f32, round to nearest even per component): one
    // instruction where two v_cvt_f16_f32 and a pack were three -- the f16 lines ran 5 % behind the bf16 ones for it
    typedef _Float16 f16x2_pk __attribute__((ext_vector_type(2)));
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_pk));
  }
  f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool F16>
__device__ __forceinline__ float lo16(unsigned w) {
  if (F16) return cv_in<true>((unsigned short)(w & 0xffffu));
  return __uint_as_float(w << 16);
}
template <bool F16>
__device__ __forceinline__ float hi16(unsigned w) {
  if (F16) return cv_in<true>((unsigned short)(w >> 16));
  return __uint_as_float(w & 0xffff0000u);
}
// leaky-relu of a packed pair, computed in f32 and rounded back (the 16-bit numerics spec).
// max(x, slope * x) == (x > 0 ? x : slope * x) for 0 <= slope <= 1 (also for +-0): one v_max_f32 per element
// instead of a compare + select (+ the wait states between them) -- these sit in the staging and epilogue phases
// of every 16-bit conv, where the vector ALU work is as long as the MFMA work for the k = 3 shapes
__device__ __forceinline__ float lrelu_max(float x, float slope) {
  const float m = x * slope;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
  return r;
}
template <bool F16>
__device__ __forceinline__ unsigned lrelu_pk(unsigned w, float slope) {
  const float a = lrelu_max(lo16<F16>(w), slope), c = lrelu_max(hi16<F16>(w), slope);
  return pk2<F16>(a, c);
}
// ---- pairs (round 6): the epilogues and the staging of the fused 16-bit kernels are VALU-bound (SQ counters: 7-15 vector
// instructions per MFMA), and gfx950 executes v_pk_mul / add / fma_f32 on two f32 values per lane at the rate of one:
// bias adds, slope products and the MRF quotient run on pairs.  Same IEEE operations per element, same results.
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ f32x2v unpack2(unsigned w) { return f32x2v{lo16<F16>(w), hi16<F16>(w)}; }
// leaky-relu of a pair: one packed product, two v_max (max(x, slope x), see lrelu_max)
__device__ __forceinline__ f32x2v lrelu2v(f32x2v v, f32x2v slope2) {
  const f32x2v m = v * slope2;
  f32x2v r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r.x) : "v"(v.x), "v"(m.x));
  asm("v_max_f32 %0, %1, %2" : "=v"(r.y) : "v"(v.y), "v"(m.y));
  return r;
}
template <bool F16>
__device__ __forceinline__ unsigned lrelu_pk2(unsigned w, f32x2v slope2) {
  const f32x2v r = lrelu2v(unpack2<F16>(w), slope2);
  return pk2<F16>(r.x, r.y);
}
// common.h: div_small_const on a pair (c2 = {1/d, 1/d}, nd2 = {-d, -d})
__device__ __forceinline__ f32x2v div_small_const2(f32x2v v, f32x2v nd2, f32x2v c2) {
  const f32x2v q = v * c2;
  const f32x2v r = __builtin_elementwise_fma(nd2, q, v);
  return __builtin_elementwise_fma(r, c2, q);
}
// a [T][C] 16-bit plane as a raw buffer (gfx9 family descriptor: 32-bit data format, no swizzle): 16-byte loads at byte
// offsets outside [0, bytes) -- rows before the utterance (negative offsets are huge unsigned ones) or behind it -- return
// zeros, i.e. the convs' zero padding (lrelu(0) = 0) without a bounds test, exec mask or branch per piece
constexpr int kBufRsrcRaw16 = 0x00020000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const unsigned short* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0, bytes, kBufRsrcRaw16);
}
__device__ __forceinline__ uint4 plane_load16(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma16(uint4 a, uint4 b, f32x16 c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace wetts

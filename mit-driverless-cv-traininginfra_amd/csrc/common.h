// Shared device helpers for the MDCV gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "tune.h"
#include <atomic>
#include <mutex>

typedef unsigned short bf16_t;   // raw bf16 bits; all arithmetic is done in fp32
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MDCV_OK 0
#define MDCV_EARG (-1)
#define MDCV_F32 0
#define MDCV_BF16 1

// Every kernel launch of the library goes through MDCV_LAUNCH.  When the in-library profiler is on (mdcv_profile_begin, runtime.hip) the
// kernel is launched with hipExtLaunchKernelGGL and a start / stop event pair: the pair carries the dispatch's OWN begin / end
// timestamps (what a rocprofv3 kernel trace reports), adds no packets to the stream, and is remembered with the kernel's host function,
// from which the symbol rocprofv3 prints is recovered.  Off: one predictable branch per launch.
extern int mdcv_g_prof;
void mdcv_prof_new(const void* fn, hipEvent_t* e0, hipEvent_t* e1);
extern thread_local hipEvent_t mdcv_t_arm;      // mdcv_stream_fork_arm: the event the NEXT launch carries as its stop event (cleared by that launch)
#define MDCV_LAUNCH(kern, grid, block, lds, st, ...)                                                      \
  do {                                                                                                    \
    if (mdcv_g_prof) {                                                                                    \
      hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                          \
      mdcv_prof_new(reinterpret_cast<const void*>(kern), &e0__, &e1__);                                   \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0__, e1__, 0, __VA_ARGS__);                      \
      if (mdcv_t_arm) { (void)hipEventRecord(mdcv_t_arm, st); mdcv_t_arm = nullptr; }   /* (profiling: a plain record behind it) */ \
    } else if (mdcv_t_arm) {                                                                              \
      hipEvent_t ea__ = mdcv_t_arm;                                                                       \
      mdcv_t_arm = nullptr;                                                                               \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, nullptr, ea__, 0, __VA_ARGS__);                   \
    } else {                                                                                              \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                        \
    }                                                                                                     \
  } while (0)

#define MDCV_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return (int)e__;                   \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// float -> bf16, round-to-nearest-even, NaN preserved: gfx950 has a packed hardware convert (v_cvt_pk_bf16_f32)
typedef __bf16 mdcv_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float mdcv_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const mdcv_f32x2_t v = {lo, hi};
  const mdcv_bf16x2_t b = __builtin_convertvector(v, mdcv_bf16x2_t);
  return *reinterpret_cast<const unsigned*>(&b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, f) & 0xffffu); }

// element traits: VEC = elements per 16-byte vector
template <typename T> struct ET;
template <> struct ET<float> {
  static constexpr int VEC = 4;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
    f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y); f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct ET<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
    f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
    f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xffff0000u);
    f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xffff0000u);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  }
};

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a function ON a device.  One cache per call site (a `static DynLds` next to the launch):
// per device the largest size already granted; the slow path (first launch of a kernel on a device, or a larger size) is serialised, so two threads
// cannot leave the attribute below what either of them asked for.  (Until round 5 the call sites kept one `static bool` per process: a second GPU in
// the same process never got the attribute and its launches above 64 KiB of LDS failed -- ADVICE r4.)
struct DynLds { std::atomic<int> granted[64]; };
inline hipError_t mdcv_dyn_lds(DynLds& c, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::atomic<int>& g = c.granted[dev & 63];
  if (g.load(std::memory_order_acquire) >= bytes) return hipSuccess;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (g.load(std::memory_order_relaxed) >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) g.store(bytes, std::memory_order_release);
  return e;
}


// Operands a pass reads for the LAST time before they are long dead (raw conv outputs y on the way forward; dz and y in the BatchNorm-backward apply;
// the residual of a fused shortcut; weight-gradient slabs in their reduce) are loaded NON-TEMPORALLY: such a pass streams tensors of 11-350 MB through
// the L2 / MALL that the co-running weight gradient and the next data gradient re-read their operand rows from.  Same box, whole YOLOv3 step, two
// processes each (scripts/ab_step.py, MDCV_LIB builds): 13.16 / 13.20 -> 13.03 / 13.05 ms with the loads of the two apply passes, 13.04 / 13.07 with the
// residual's too; RektNet 7.59 / 7.86 -> 7.45 / 7.67 ms.  The OUTPUTS stay ordinary stores -- their readers are the very next kernels (non-temporal stores
// gave the gain back: 13.16 / 13.15); the LDS-DMA loads of the 1x1 backward's x / addsrc tiles with the nt policy and the producer loads of the fused
// 1x1 forward block (pw_block.hip) measured level (12.80 vs 12.81; 12.85 / 12.84 / 12.83 vs 12.85 / 12.88 / 12.82) and keep the default policy.
typedef unsigned mdcv_u32x4_nt_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 mdcv_ld_stream(const void* p) {
  const mdcv_u32x4_nt_t v = __builtin_nontemporal_load(reinterpret_cast<const mdcv_u32x4_nt_t*>(p));
  return uint4{v.x, v.y, v.z, v.w};
}

// BatchNorm(+activation)-backward apply of ONE element: dy = cA * g + cB * y + cC with g = dz * act'(scale * y + shift).  Explicit fused
// multiply-adds in a fixed order: the apply pass (elementwise.hip) and the kernels that form dy in their operand load (conv_igemm.hip BNA)
// must round identically -- left to -ffp-contract, two kernels contract the same expression differently and differ in the last fp32 bit.
__device__ __forceinline__ float mdcv_bn_bwd_dy(float dz, float y, float scale, float shift, float cA, float cB, float cC, int act, float slope) {
  const float pre = __builtin_fmaf(y, scale, shift);
  const float g = dz * (act == 0 ? 1.f : (pre > 0.f ? 1.f : slope));
  return __builtin_fmaf(cA, g, __builtin_fmaf(cB, y, cC));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

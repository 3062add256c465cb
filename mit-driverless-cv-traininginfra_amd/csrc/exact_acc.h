// Exact, order-independent accumulation of fp32 partial sums (gfx950): BatchNorm statistics without partial rows and without a finalize launch.
//
// A conv kernel used to write one partial row [2][Nout] (sum, sum of squares) per 128 output positions and mdcv_bn_stats_finalize summed the
// rows in a launch of its own between the conv and the BatchNorm-apply pass: a 6 us kernel that costs 10-12 us of critical path, 72 times per
// YOLOv3 forward (what-if timing).  An in-launch fold of the rows (round 4, removed) paid the same as a hand-off.  Here the
// epilogue ADDS its partial sums to a per-layer accumulator with fire-and-forget agent-scope integer atomics: nobody waits, nobody polls, and
// the consumer kernel (after the kernel boundary) reads the totals in its prologue.
//
// Integer addition is associative, so the totals do not depend on the order the atomics land in: bit-reproducible, unlike float atomics.
// And it is exact: an fp32 value m * 2^e (24-bit m) is split over XACC_DIGITS signed 64-bit words that hold 40-bit digits of a fixed-point
// number with quantum 2^-70 -- digit k counts units of 2^(-70 + 40 k) -- so that any value from 2^-70 to 2^49 is added without rounding (at
// most two digits are touched) and 2^20 additions cannot overflow a word.  Magnitudes below 2^-46 lose their low bits (truncation toward
// zero of the shifted mantissa; nothing in a BatchNorm statistic is that small and matters), magnitudes above 2^49 saturate, and a
// non-finite value poisons the top word (atomic max with INT64_MAX; xacc_value returns NaN) so that a diverged run still shows NaN.
//
// Same-address atomics retire at ~10 ns each on MI355X whoever issues them (scripts/probes/atomics_probe.hip), different lines in parallel:
// a layer's rows are therefore spread over `reps` replicas (row % reps) so that a word sees at most a few hundred additions per launch; the
// consumer adds the replicas (integers again: exact, any order).
//
// Layout of one layer's accumulator: long long [reps][XACC_DIGITS][nsums][C] (C contiguous: a wave's 64 channels are 512 contiguous bytes
// per digit).  Zeroed by one memset per forward (engine.Plan holds all layers in one arena).
#pragma once
#include <hip/hip_runtime.h>

#define XACC_DIGITS 3

struct XAccArgs {
  long long* acc;      // NULL: off (partial rows + finalize launch)
  int reps;            // power of two
};

// add v to the element whose digit-0 word is p; ds = distance between consecutive digits in words (nsums * C)
__device__ __forceinline__ void xacc_add(long long* p, size_t ds, float v) {
  const unsigned u = __float_as_uint(v);
  const int e = (int)((u >> 23) & 0xffu);
  unsigned long long m = u & 0x7fffffu;
  if (e == 255) {                                    // inf / nan
    __hip_atomic_fetch_max(p + 2 * ds, 0x7fffffffffffffffLL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  int s;                                             // v = m * 2^(s - 70)
  if (e == 0) s = -149 + 70; else { m |= 0x800000u; s = e - 150 + 70; }
  if (s < 0) { m = s > -24 ? m >> (-s) : 0; s = 0; }
  if (s > 95) { m = 0xffffffu; s = 95; }             // saturate: 2^49
  if (m == 0) return;
  const int k = s / 40, r = s - k * 40;              // m << r < 2^64 (r < 40, m < 2^24)
  const unsigned long long w = m << r;
  long long lo = (long long)(w & 0xffffffffffULL), hi = (long long)(w >> 40);
  if (u >> 31) { lo = -lo; hi = -hi; }
  if (lo) __hip_atomic_fetch_add(p + (size_t)k * ds, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (hi) __hip_atomic_fetch_add(p + (size_t)(k + 1) * ds, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (k + 1 <= 2: s <= 95 keeps w >> 40 inside digit 2 when k == 2 -> hi == 0 there)
}

// A top-digit word of ONE replica that a non-finite value poisoned.  Consumers must look at every replica's word with this BEFORE adding
// the replicas (INT64_MAX + INT64_MAX wraps to -2: a finite-looking sum) and carry XACC_POISON instead of the sum.
#define XACC_POISON 0x7fffffffffffffffLL
__device__ __forceinline__ bool xacc_poisoned(long long d2) { return d2 > (1LL << 60) || d2 < -(1LL << 60); }

// the digits of one element (replicas added by the caller, poison carried as XACC_POISON) -> double
__device__ __forceinline__ double xacc_value(long long d0, long long d1, long long d2) {
  if (xacc_poisoned(d2)) return __builtin_nan("");
  return (double)d0 * 0x1p-70 + (double)d1 * 0x1p-30 + (double)d2 * 0x1p10;
}

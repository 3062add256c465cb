// BatchNorm-backward partial sums fused into the epilogue of the data-gradient kernel that produces dz.
//
// The BN(+activation) backward of layer L needs  sum_p g  and  sum_p g * (y - mean)  per channel, with
// g = dz * act'(scale*y + shift).  dz is the output of the NEXT layer's data gradient (after its addsrc), so the
// store loop of that kernel already holds dz in registers: it only has to load the matching y vector.  One partial row
// [2][C] is written per 128 output positions (same granularity as the forward statistics); bn_colfinal_kernel sums them.
// This removes the stand-alone bn_act_bwd_reduce pass (one read of dz and one launch per BatchNorm).
#pragma once
#include "common.h"

// Inference epilogue of a forward conv: out = act(acc * oscale[n] + bias[n]) (+ addsrc) -- BatchNorm with running statistics and the
// activation folded into the conv's store path (mdcv_conv2d_affine_act).  oscale == NULL: plain bias epilogue.
struct EpiArgs { const float* oscale; int act; float slope; };

struct BnFuseArgs {
  const void* y;            // raw conv output of the BatchNorm being differentiated, same [pixel][channel] indexing as dz; NULL = off
  const float* scale; const float* shift; const float* mean;
  float* partial;           // [rows][2][C] fp32
  int ldy, act, row_base;   // act: 0 none, 1 leaky, 2 relu (slope 0)
  float slope;
};

// Per-thread state of the store loop.  Every thread keeps a FIXED 8-channel vector (cv) and walks rows row0, row0+RPP, ...
template <typename T, int BN, int NT>
struct BnFuseAcc {
  static constexpr int VEC = ET<T>::VEC;
  static constexpr int VPRO = BN / VEC;          // channel vectors per tile row
  static constexpr int RPP = NT / VPRO;          // rows covered by one pass of the block
  static_assert(NT % VPRO == 0 && RPP <= 128 && 128 % RPP == 0 && VPRO <= 64, "store-loop geometry");
  float fs[VEC], fb[VEC], fm[VEC], sg[VEC], sx[VEC];

  __device__ __forceinline__ void init(const BnFuseArgs& f, int n0, int Nout) {   // n0 = first channel of this thread's vector
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool ok = f.y && n0 + e < Nout;
      fs[e] = ok ? f.scale[n0 + e] : 0.f; fb[e] = ok ? f.shift[n0 + e] : 0.f; fm[e] = ok ? f.mean[n0 + e] : 0.f;
      sg[e] = 0.f; sx[e] = 0.f;
    }
  }
  // dv: the VEC values of dz just stored (as rounded to T); yq: the 16-byte y vector of the same pixel / channels
  __device__ __forceinline__ void add(const BnFuseArgs& f, const float (&dv)[VEC], const uint4& yq) {
    float yv[VEC];
    ET<T>::unpack(yq, yv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float pre = yv[e] * fs[e] + fb[e];
      const float g = (f.act != 0 && !(pre > 0.f)) ? dv[e] * f.slope : dv[e];
      sg[e] += g;
      sx[e] += g * (yv[e] - fm[e]);
    }
  }
  // Block-wide: fold the threads that share a channel vector and write one partial row.  red: NT/64 * BN floats of LDS.
  // Must be called by every thread of the block (uniform control flow).  Resets the accumulators.
  __device__ __forceinline__ void flush(const BnFuseArgs& f, float* red, int tid, int n_tile0, int Nout, int row) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off >= VPRO; off >>= 1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { sg[e] += __shfl_xor(sg[e], off, 64); sx[e] += __shfl_xor(sx[e], off, 64); }
    }
    // two rounds through the same NT/64 * BN floats of scratch (sum g, then sum g*(y-mean)): no LDS beyond the statistics area
    float tot[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (lane < VPRO) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[wave * BN + lane * VEC + e] = k == 0 ? sg[e] : sx[e];
      }
      __syncthreads();
      if (tid < BN) {
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot[k] += red[w * BN + tid];
      }
      __syncthreads();
    }
    if (tid < BN) {
      const int n = n_tile0 + tid;
      if (n < Nout) {
        float* prow = f.partial + (size_t)(f.row_base + row) * 2 * Nout;
        prow[n] = tot[0];
        prow[Nout + n] = tot[1];
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sg[e] = 0.f; sx[e] = 0.f; }
  }
};

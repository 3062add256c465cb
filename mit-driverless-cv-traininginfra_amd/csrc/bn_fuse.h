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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for every global store the thread
// has in flight to be acknowledged; inside a store loop (flush() below runs once per 128 stored rows) that exposes the store latency
// once per group and tile.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Cross-lane sums without LDS round trips (gfx950).  swap_add16(u, v): v_permlane16_swap exchanges the odd 16-lane rows of u with the even
// rows of v; the sum of the two results is rows [u0+u1, v0+v1, u2+u3, v2+v3].  swap_add32: the same for the 32-lane halves,
// [u_lo + u_hi, v_lo + v_hi].  ror_add<CTRL>: x + (x rotated inside its 16-lane row), CTRL = 0x120 | lanes.
__device__ __forceinline__ float swap_add16(float u, float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, u), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float swap_add32(float u, float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, u), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
template <int CTRL> __device__ __forceinline__ float ror_add(float x) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

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
      const float pre = __builtin_fmaf(yv[e], fs[e], fb[e]);   // (the sign at the activation boundary must be the apply pass's: common.h mdcv_bn_bwd_dy)
      const float g = (f.act != 0 && !(pre > 0.f)) ? dv[e] * f.slope : dv[e];
      sg[e] += g;
      sx[e] += g * (yv[e] - fm[e]);
    }
  }
  // ---- bf16 form (VEC = 8, 16 values per thread): no barrier inside the store loop.
  // The lanes that share a channel vector sit VPRO apart.  Two rounds of gfx950 row swaps fold the four 16-lane rows AND halve the
  // register count each time (swap(u, v) + add leaves u's pair sums in the even rows and v's in the odd rows), so 16 values cost
  // 12 swaps + 12 adds instead of 32 ds_bpermute round trips; row r of x[n] then holds value 4n + r.  Lanes closer than a row
  // (VPRO < 16) finish with DPP row rotations.  The wave leaves its 2*BN sums in `ws`, LDS that only this wave touches -- the kernels
  // pass the staging rows the wave itself read in the first pass of the group, which are dead by then -- and the waves meet ONCE,
  // after the tile's last store (write_row below).  With a block-wide flush after every 128 rows the four barriers made every wave
  // wait for the slowest one's loads in the middle of the store loop: 7..17 us per launch on the 1x1 data gradients of YOLOv3.
  static constexpr bool kWaveFold = VEC == 8 && VPRO <= 16;
  static constexpr int WROWS = 64 / VPRO;                  // staging rows one wave reads per pass: consecutive, private to the wave
  __device__ __forceinline__ void fold_wave(float* ws, int lane) {
    float w[8], x[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { w[m] = swap_add16(sg[2 * m], sg[2 * m + 1]); w[4 + m] = swap_add16(sx[2 * m], sx[2 * m + 1]); }
#pragma unroll
    for (int n = 0; n < 4; ++n) x[n] = swap_add32(w[2 * n], w[2 * n + 1]);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if constexpr (VPRO <= 8) x[n] = ror_add<0x128>(x[n]);
      if constexpr (VPRO <= 4) x[n] = ror_add<0x124>(x[n]);
      if constexpr (VPRO <= 2) x[n] = ror_add<0x122>(x[n]);
    }
    const int rho = lane >> 4, c16 = lane & 15;            // x[0]: sum g of channel 8 cv + rho, x[1]: 8 cv + 4 + rho ; x[2], x[3]: sum g (y - mean)
    if (c16 < VPRO) {
      float* r0 = ws + c16 * VEC + rho;
      r0[0] = x[0]; r0[4] = x[1]; r0[BN] = x[2]; r0[BN + 4] = x[3];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sg[e] = 0.f; sx[e] = 0.f; }
  }
  // after a barrier: element t of [2][BN] of one group's partial row = sum over the waves' scratches (ws0 + w * wstride floats)
  static __device__ __forceinline__ void write_row(const BnFuseArgs& f, const float* ws0, int wstride, int t, int n_tile0, int Nout, int row) {
    const int which = t / BN, c = t - which * BN;
    float tot = 0.f;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) tot += ws0[wv * wstride + t];
    const int n = n_tile0 + c;
    if (n < Nout) f.partial[(size_t)(f.row_base + row) * 2 * Nout + (size_t)which * Nout + n] = tot;
  }

  // ---- generic form (fp32 parity mode): block-wide fold after every group.  red: NT/64 * BN floats of LDS.
  // Must be called by every thread of the block (uniform control flow).  Resets the accumulators.
  __device__ __forceinline__ void flush(const BnFuseArgs& f, float* red, int tid, int n_tile0, int Nout, int row) {
    const int lane = tid & 63, wave = tid >> 6;
    {
#pragma unroll
      for (int off = 32; off >= VPRO; off >>= 1) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { sg[e] += __shfl_xor(sg[e], off, 64); sx[e] += __shfl_xor(sx[e], off, 64); }
      }
      float tot[2] = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 2; ++k) {                          // two rounds through the first NT/64 * BN floats of the scratch
        if (lane < VPRO) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) red[wave * BN + lane * VEC + e] = k == 0 ? sg[e] : sx[e];
        }
        lds_only_barrier();
        if (tid < BN) {
#pragma unroll
          for (int w = 0; w < NT / 64; ++w) tot[k] += red[w * BN + tid];
        }
        lds_only_barrier();
      }
      if (tid < BN) {
        const int n = n_tile0 + tid;
        if (n < Nout) {
          float* prow = f.partial + (size_t)(f.row_base + row) * 2 * Nout;
          prow[n] = tot[0];
          prow[Nout + n] = tot[1];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sg[e] = 0.f; sx[e] = 0.f; }
  }
};

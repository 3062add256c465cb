// HBM-bound NHWC elementwise / per-channel-reduction kernels for gfx950.
//
// Replaces nn.BatchNorm2d (train + eval), nn.LeakyReLU / nn.ReLU, the shortcut add, nn.Upsample(nearest) and
// their autograd backward on the reference hot path (CVC-YOLOv3/models.py:66-71,86-88,325-327;
// RektNet/resnet.py:22-27, keypoint_net.py:59).
//
// All kernels are "strip" kernels: a block owns a contiguous strip of pixels, a thread owns one 16-byte channel
// vector (8 bf16 / 4 fp32) and walks down the strip, so every access is a coalesced 16-byte load/store and per-channel
// reductions stay in registers until one LDS fold + one fp64 atomic per channel per block.
#include "common.h"
#include "exact_acc.h"

namespace {

using ::mdcv_ld_stream;
__device__ __forceinline__ uint4 ld_stream(const void* p) { return mdcv_ld_stream(p); }     // (common.h: non-temporal 16-byte load of a last-use operand)

struct Strip {
  int CV, PPI, PB;   // vectors per pixel, pixels per block-iteration, pixels per block
};
template <typename T> static Strip make_strip(int M, int C, int target_blocks, int min_iters = 1) {
  Strip s;
  s.CV = C / ET<T>::VEC;
  s.PPI = 256 / s.CV; if (s.PPI < 1) s.PPI = 1;
  long long pb = ((long long)M + target_blocks - 1) / target_blocks;
  pb = ((pb + s.PPI - 1) / s.PPI) * s.PPI;
  if (pb < (long long)s.PPI * min_iters) pb = (long long)s.PPI * min_iters;   // amortise the per-block prologue / fold
  s.PB = (int)pb;
  return s;
}

// reduction kernels end in one fp64 atomic per channel per block: keep (blocks x channels x sums) within a budget
static int reduce_blocks(int C, int nsums) {
  (void)C; (void)nsums;
  return 1024;     // partial rows are plain stores now (no atomics): cap only the row count the follow-up tree reduce has to read
}

// fold the per-thread channel-vector sums of a block: lanes that own the same channel vector are CV apart.
// Requires 256 % CV == 0.  red: 256*VEC floats.  Result: threads tid < CV hold the block total for vector tid in v[].
template <int VEC>
__device__ __forceinline__ void block_fold(float (&v)[VEC], int CV, float* red, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  if (CV < 64) {
    for (int off = 32; off >= CV; off >>= 1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] += __shfl_xor(v[e], off, 64);
    }
  }
  const int span = CV < 64 ? CV : 64;
  __syncthreads();
  if (lane < span) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[(wave * 64 + lane) * VEC + e] = v[e];
  }
  __syncthreads();
  if (tid < CV) {
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    for (int w = 0; w < 4; ++w) {
      if (CV > 64 && ((w * 64) % CV) != (tid & ~63)) continue;
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] += red[(w * 64 + (tid & 63)) * VEC + e];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = s[e];
  }
}

// VEC consecutive per-channel coefficients as 16-byte loads (arrays are 16-byte aligned, c0 is a multiple of VEC)
template <int VEC>
__device__ __forceinline__ void ldcoef(const float* __restrict__ p, int c0, float (&o)[VEC], float dflt) {
  if (!p) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = dflt;
    return;
  }
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(p + c0 + 4 * q);
    o[4 * q] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
  }
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope) { return act == 0 ? v : (v > 0.f ? v : v * slope); }
__device__ __forceinline__ float act_grad(float pre, int act, float slope) { return act == 0 ? 1.f : (pre > 0.f ? 1.f : slope); }

// ---------------------------------------------------------------- layout conversion
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W, int ldc, int Cpad) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = Cpad / VEC;
  const long long total = (long long)B * H * W * CV;
  const int HW = H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long pix = i / CV;
    const int b = (int)(pix / HW), hw = (int)(pix - (long long)b * HW);
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = cv * VEC + e;
      v[e] = c < C ? src[((size_t)b * C + c) * HW + hw] : 0.f;
    }
    *reinterpret_cast<uint4*>(dst + (size_t)pix * ldc + cv * VEC) = ET<T>::pack(v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W, int ldc) {
  const long long total = (long long)B * C * H * W;
  const int HW = H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int hw = (int)(i % HW);
    const long long bc = i / HW;
    const int c = (int)(bc % C), b = (int)(bc / C);
    dst[i] = ET<T>::ld(src + ((size_t)b * HW + hw) * ldc + c);
  }
}

// ---------------------------------------------------------------- BatchNorm statistics
// partial[rows][nsums][C] (fp32, from the conv epilogue) -> accum[nsums][C] (fp64)
// block = 64 columns x 4 row lanes, each block covers 128 rows: coalesced 256-byte row reads, one atomic per column per block
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partial, int rows, int cols, double* __restrict__ accum) {
  __shared__ double red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * 128;
  double s = 0.0;
  if (c < cols) {
    const int r1 = min(rows, r0 + 128);
    for (int r = r0 + ry; r < r1; r += 4) s += (double)partial[(size_t)r * cols + c];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < cols) atomicAdd(&accum[c], red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

// accum[0]=sum, accum[1]=sumsq over `count` samples per channel -> batch mean / biased var -> scale, shift ; running stats
__global__ void bn_finalize_kernel(double* __restrict__ accum, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double s = accum[c], q = accum[C + c];
  accum[c] = 0.0; accum[C + c] = 0.0;                 // ready for the next step
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma[c], b = beta[c];
  scale[c] = g * invstd;
  shift[c] = b - (float)mean * g * invstd;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                      const float* __restrict__ rv, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rv[c] + eps);
  scale[c] = gamma[c] * is;
  shift[c] = beta[c] - rm[c] * gamma[c] * is;
}
// the same with the conv's own bias folded in: BN(conv + b) = conv * scale + (shift + b * scale)
__global__ void bn_eval_coeffs_bias_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                           const float* __restrict__ rv, float eps, const float* __restrict__ conv_bias,
                                           float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc + (conv_bias ? conv_bias[c] * sc : 0.f);
}

// ---------------------------------------------------------------- fused BN-apply + activation (+ second BN branch) (+ residual)
struct BnActArgs {
  const void* y1; const void* y2; const void* resid; void* out;
  const float* s1; const float* b1; const float* s2; const float* b2;
  int ld1, ld2, ldr, ldo, M, C, act, PB, CV, PPI;
  float slope;
};
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnActArgs a) {
  constexpr int VEC = ET<T>::VEC;
  const int tid = threadIdx.x;
  if (tid >= a.PPI * a.CV) return;
  const int cv = tid % a.CV, pi = tid / a.CV;
  float s1[VEC], b1[VEC], s2[VEC], b2[VEC];
  ldcoef<VEC>(a.s1, cv * VEC, s1, 1.f); ldcoef<VEC>(a.s1 ? a.b1 : nullptr, cv * VEC, b1, 0.f);
  ldcoef<VEC>(a.y2 ? a.s2 : nullptr, cv * VEC, s2, 0.f); ldcoef<VEC>(a.y2 ? a.b2 : nullptr, cv * VEC, b2, 0.f);
  const long long p0 = (long long)blockIdx.x * a.PB;
  const long long p1 = min((long long)a.M, p0 + a.PB);
  const T* y1 = reinterpret_cast<const T*>(a.y1);
  const T* y2 = reinterpret_cast<const T*>(a.y2);
  const T* rs = reinterpret_cast<const T*>(a.resid);
  T* out = reinterpret_cast<T*>(a.out);
  for (long long pb = p0 + pi; pb < p1; pb += 4 * a.PPI) {
    uint4 q1[4], q2[4], qr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long p = pb + (long long)u * a.PPI;
      if (p < p1) {
        q1[u] = ld_stream(y1 + p * a.ld1 + cv * VEC);
        if (y2) q2[u] = ld_stream(y2 + p * a.ld2 + cv * VEC);
        if (rs) qr[u] = ld_stream(rs + p * a.ldr + cv * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long p = pb + (long long)u * a.PPI;
      if (p >= p1) break;
      float v[VEC], w[VEC];
      ET<T>::unpack(q1[u], v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = v[e] * s1[e] + b1[e];
      if (y2) {
        ET<T>::unpack(q2[u], w);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += w[e] * s2[e] + b2[e];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = act_fwd(v[e], a.act, a.slope);
      if (rs) {
        ET<T>::unpack(qr[u], w);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += w[e];
      }
      *reinterpret_cast<uint4*>(out + p * a.ldo + cv * VEC) = ET<T>::pack(v);
    }
  }
}


// The same pass with the statistics read from exact accumulators (exact_acc.h): the conv ADDED its per-tile sums to [reps][3][2][C] 64-bit
// words with fire-and-forget integer atomics; every workgroup here adds the replicas (integers: exact, any order), converts and forms
// scale / shift for all C channels in LDS, with its first strip of y already requested; workgroup 0 also publishes scale / shift / mean /
// invstd for the backward and updates the running statistics.  No rows, no finalize launch, no hand-off inside a launch.
// Thread layout of the prologue: C <= 256: the 256 threads are G = 256 / Cp groups (Cp = C rounded up to a power of two), group g adds
// replicas [g rp, (g + 1) rp) of channel tid % Cp and the groups meet in LDS; C > 256: thread t owns channels t + 256 k, k < NCH, all replicas.
// NCH * rp <= 4: at most 24 words per thread are in flight together (one memory round trip) and the kernel keeps the register budget -- i.e.
// the occupancy -- of the plain apply pass (with 48 words: 138 registers, three waves per SIMD instead of four, +7 us on the 104^2 tensors).
struct BnXAccArgs {
  const long long* acc; int reps; double count, inv_count;
  const float* gamma; const float* beta; float* rm; float* rv; float momentum, eps;
  float* scale; float* shift; float* mean; float* invstd;
  int Cp, rp;                               // C <= 256: Cp = pow2 >= C, rp = max(1, reps / (256 / Cp)); C > 256: Cp = 256, rp = reps
};
template <typename T, int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void bn_act_fwd_xacc_kernel(BnActArgs a, BnXAccArgs f) {
  // (waves_per_eu(4, 4): with a bare minimum of four the scheduler went for five waves and paid for the registers by waiting for every
  //  strip load before issuing the next one and for every store before the next -- 28 us where the plain pass takes 23)
  constexpr int VEC = ET<T>::VEC;
  constexpr int RPMAX = 4 / NCH;
  __shared__ float cs[1024], cb[1024];
  __shared__ long long sd[NCH == 1 ? 6 * 256 : 1];
  const int tid = threadIdx.x;
  const int cl = tid & (f.Cp - 1), g = tid / f.Cp;                     // (NCH > 1: Cp = 256, g = 0)
  const size_t ds = 2 * (size_t)a.C, rs = (size_t)XACC_DIGITS * 2 * a.C;
  long long w[NCH][RPMAX][6];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = cl + 256 * k;
#pragma unroll
    for (int j = 0; j < RPMAX; ++j) {
      const int rep = g * f.rp + j;
      const bool ok = c < a.C && j < f.rp && rep < f.reps;
      const long long* p = f.acc + (size_t)(ok ? rep : 0) * rs + (ok ? c : 0);
#pragma unroll
      for (int d = 0; d < XACC_DIGITS; ++d) {
        w[k][j][2 * d + 0] = ok ? p[d * ds] : 0;
        w[k][j][2 * d + 1] = ok ? p[d * ds + a.C] : 0;
      }
    }
  }
  // gamma / beta (and, in the publishing workgroup, the running statistics) are requested NOW, beside the accumulator words: behind the
  // statistics arithmetic each would be one more dependent memory round trip in front of every workgroup's first store.
  const bool publisher = blockIdx.x == gridDim.x - 1;                  // one extra workgroup with an empty strip: nobody's strip waits for the publishing stores
  float gmv[NCH], btv[NCH], rmv[NCH], rvv[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = cl + 256 * k;
    const bool ok = g == 0 && c < a.C;
    gmv[k] = ok ? f.gamma[c] : 0.f; btv[k] = ok ? f.beta[c] : 0.f;
    rmv[k] = ok && publisher && f.rm ? f.rm[c] : 0.f; rvv[k] = ok && publisher && f.rm ? f.rv[c] : 0.f;
  }
  const bool active = tid < a.PPI * a.CV;
  const int cv = active ? tid % a.CV : 0, pi = active ? tid / a.CV : 0;
  const long long p0 = (long long)blockIdx.x * a.PB;
  const long long p1 = active ? min((long long)a.M, p0 + a.PB) : 0;    // (inactive threads and the publisher: an empty strip)
  const T* y1 = reinterpret_cast<const T*>(a.y1);
  const T* rsd = reinterpret_cast<const T*>(a.resid);
  T* out = reinterpret_cast<T*>(a.out);
  // the first strip is requested above the prologue
  uint4 q1[4], qr[4];
  long long pb = p0 + pi;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long p = pb + (long long)u * a.PPI;
    if (p < p1) {
      q1[u] = ld_stream(y1 + p * a.ld1 + cv * VEC);
      if (rsd) qr[u] = ld_stream(rsd + p * a.ldr + cv * VEC);
    }
  }
  // A non-finite partial sum poisoned the top digit of ITS replica (atomic max with INT64_MAX, exact_acc.h).  The poison is looked for per
  // replica BEFORE the replicas are added: two poisoned words sum to -2, thirty-two to -32 -- finite garbage.  A poisoned
  // channel keeps XACC_POISON through both additions and xacc_value turns it into NaN.
  long long t[NCH][6];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < RPMAX; ++j) bad = bad || xacc_poisoned(w[k][j][4]) || xacc_poisoned(w[k][j][5]);
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      long long x = 0;
#pragma unroll
      for (int j = 0; j < RPMAX; ++j) x += (e >= 4 && xacc_poisoned(w[k][j][e])) ? 0 : w[k][j][e];
      t[k][e] = (e >= 4 && bad) ? XACC_POISON : x;
    }
  }
  if constexpr (NCH == 1) {
    if (f.Cp < 256) {                                                  // (uniform) the replica groups meet
#pragma unroll
      for (int e = 0; e < 6; ++e) sd[e * 256 + tid] = t[0][e];
      __syncthreads();
      if (g == 0) {
        const int ng = 256 / f.Cp;
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          long long x = 0;
          for (int gg = 0; gg < ng; ++gg) {
            const long long v = sd[e * 256 + gg * f.Cp + cl];
            if (e >= 4 && xacc_poisoned(v)) bad = true; else x += v;
          }
          t[0][e] = x;
        }
        if (bad) { t[0][4] = XACC_POISON; t[0][5] = XACC_POISON; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = cl + 256 * k;
    if (g == 0 && c < a.C) {
      const double s0 = xacc_value(t[k][0], t[k][2], t[k][4]), s1 = xacc_value(t[k][1], t[k][3], t[k][5]);
      const double mean = s0 * f.inv_count;                               // (no fp64 division / square root in every workgroup's prologue)
      double var = s1 * f.inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = 1.0f / sqrtf((float)var + f.eps);
      const float gm = gmv[k], b = btv[k];
      const float sc = gm * invstd, sh = b - (float)mean * gm * invstd;
      cs[c] = sc; cb[c] = sh;
      if (publisher) {
        f.scale[c] = sc; f.shift[c] = sh; f.mean[c] = (float)mean; f.invstd[c] = invstd;
        if (f.rm) {
          const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
          f.rm[c] = (1.f - f.momentum) * rmv[k] + f.momentum * (float)mean;
          f.rv[c] = (1.f - f.momentum) * rvv[k] + f.momentum * (float)unbiased;
        }
      }
    }
  }
  if (publisher) return;                                               // (uniform: its strip is empty)
  __syncthreads();
  if (!active) return;
  float s1[VEC], b1[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s1[e] = cs[cv * VEC + e]; b1[e] = cb[cv * VEC + e]; }
  // (only the FIRST strip is requested ahead; from the second on this is the loop of bn_act_fwd_kernel: load four, compute, store four)
  for (bool first = true; pb < p1; pb += 4 * a.PPI, first = false) {
    if (!first) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long p = pb + (long long)u * a.PPI;
        if (p < p1) {
          q1[u] = ld_stream(y1 + p * a.ld1 + cv * VEC);
          if (rsd) qr[u] = ld_stream(rsd + p * a.ldr + cv * VEC);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long p = pb + (long long)u * a.PPI;
      if (p >= p1) break;
      float v[VEC], x[VEC];
      ET<T>::unpack(q1[u], v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = act_fwd(v[e] * s1[e] + b1[e], a.act, a.slope);
      if (rsd) {
        ET<T>::unpack(qr[u], x);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += x[e];
      }
      *reinterpret_cast<uint4*>(out + p * a.ldo + cv * VEC) = ET<T>::pack(v);
    }
  }
}

// backward, pass 1: g = dout * act'(pre);  accum += [sum g, sum g*xhat1, (sum g*xhat2)]   (fp64 atomics, one per channel per block)
struct BnBwdArgs {
  const void* dout; const void* y1; const void* y2; void* dy1; void* dy2;
  const float* s1; const float* b1; const float* m1; const float* is1;
  const float* s2; const float* b2; const float* m2; const float* is2;
  const float* cA1; const float* cB1; const float* cC1; const float* cA2; const float* cB2; const float* cC2;
  double* accum; float* partial;
  int ldd, ld1, ld2, ldy1, ldy2, M, C, act, PB, CV, PPI;
  float slope;
};
template <typename T, bool DUAL>     // DUAL: see bn_act_bwd_apply_kernel
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(BnBwdArgs a) {
  constexpr int VEC = ET<T>::VEC;
  constexpr int ND = DUAL ? VEC : 1;
  __shared__ float red[256 * VEC];
  const int tid = threadIdx.x;
  const bool active = tid < a.PPI * a.CV;
  const int cv = active ? tid % a.CV : 0, pi = active ? tid / a.CV : 0;
  constexpr int nsum = DUAL ? 3 : 2;
  float s1[VEC], b1[VEC], m1[VEC], i1[VEC], s2[ND], b2[ND], m2[ND], i2[ND];
  ldcoef<VEC>(a.s1, cv * VEC, s1, 1.f); ldcoef<VEC>(a.b1, cv * VEC, b1, 0.f); ldcoef<VEC>(a.m1, cv * VEC, m1, 0.f); ldcoef<VEC>(a.is1, cv * VEC, i1, 0.f);
  if constexpr (DUAL) {
    ldcoef<VEC>(a.s2, cv * VEC, s2, 0.f); ldcoef<VEC>(a.b2, cv * VEC, b2, 0.f);
    ldcoef<VEC>(a.m2, cv * VEC, m2, 0.f); ldcoef<VEC>(a.is2, cv * VEC, i2, 0.f);
  }
  float sg[VEC], sx1[VEC], sx2[ND];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sg[e] = 0.f; sx1[e] = 0.f; }
#pragma unroll
  for (int e = 0; e < ND; ++e) sx2[e] = 0.f;
  const long long p0 = (long long)blockIdx.x * a.PB;
  const long long p1 = min((long long)a.M, p0 + a.PB);
  const T* dout = reinterpret_cast<const T*>(a.dout);
  const T* y1 = reinterpret_cast<const T*>(a.y1);
  const T* y2 = reinterpret_cast<const T*>(a.y2);
  if (active)
    for (long long pb = p0 + pi; pb < p1; pb += 4 * a.PPI) {
      uint4 qd[4], qv[4], qw[DUAL ? 4 : 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                     // issue all loads of 4 pixels before touching any
        const long long p = pb + (long long)u * a.PPI;
        if (p < p1) {
          qd[u] = *reinterpret_cast<const uint4*>(dout + p * a.ldd + cv * VEC);
          qv[u] = *reinterpret_cast<const uint4*>(y1 + p * a.ld1 + cv * VEC);
          if constexpr (DUAL) qw[u] = *reinterpret_cast<const uint4*>(y2 + p * a.ld2 + cv * VEC);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
      if (pb + (long long)u * a.PPI >= p1) break;
      float d[VEC], v[VEC], w[ND];
      ET<T>::unpack(qd[u], d);
      ET<T>::unpack(qv[u], v);
      if constexpr (DUAL) ET<T>::unpack(qw[u], w);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float pre = v[e] * s1[e] + b1[e];
        if constexpr (DUAL) pre += w[e] * s2[e] + b2[e];
        const float g = d[e] * act_grad(pre, a.act, a.slope);
        sg[e] += g;
        sx1[e] += g * (v[e] - m1[e]) * i1[e];
        if constexpr (DUAL) sx2[e] += g * (w[e] - m2[e]) * i2[e];
      }
      }
    }
  // fold the PPI pixel-lanes that share a channel vector; one partial row [nsum][C] per block (summed by partial_reduce)
  float* prow = a.partial + (size_t)blockIdx.x * nsum * a.C;
  block_fold<VEC>(sg, a.CV, red, tid);
  if (tid < a.CV) *reinterpret_cast<uint4*>(prow + tid * VEC) = ET<float>::pack(sg), (VEC == 8 ? (void)(*reinterpret_cast<uint4*>(prow + tid * VEC + 4) = ET<float>::pack(sg + 4)) : (void)0);
  block_fold<VEC>(sx1, a.CV, red, tid);
  if (tid < a.CV) *reinterpret_cast<uint4*>(prow + a.C + tid * VEC) = ET<float>::pack(sx1), (VEC == 8 ? (void)(*reinterpret_cast<uint4*>(prow + a.C + tid * VEC + 4) = ET<float>::pack(sx1 + 4)) : (void)0);
  if constexpr (DUAL) {
    block_fold<VEC>(sx2, a.CV, red, tid);
    if (tid < a.CV) *reinterpret_cast<uint4*>(prow + 2 * a.C + tid * VEC) = ET<float>::pack(sx2), (VEC == 8 ? (void)(*reinterpret_cast<uint4*>(prow + 2 * a.C + tid * VEC + 4) = ET<float>::pack(sx2 + 4)) : (void)0);
  }
}

// backward finalize: accum[0]=sum g, accum[kx]=sum g*xhat  ->  dgamma, dbeta and the per-channel coefficients of
//    dy = cA*g + cB*y + cC   ( = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) )
__global__ void bn_bwd_finalize_kernel(double* __restrict__ accum, int kx, int zero_after, double count, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ cA, float* __restrict__ cB, float* __restrict__ cC,
                                       int C, int nsums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double sg = accum[c], sgx = accum[(size_t)kx * C + c];
  if (zero_after) for (int k = 0; k < nsums; ++k) accum[(size_t)k * C + c] = 0.0;
  dgamma[c] = (float)sgx;
  dbeta[c] = (float)sg;
  const double g = gamma[c], is = invstd[c], mu = mean[c];
  const double mg = sg / count, mgx = sgx / count;
  cA[c] = (float)(g * is);
  cB[c] = (float)(-g * is * is * mgx);
  cC[c] = (float)(-g * is * mg + g * is * is * mu * mgx);
}

// ---------------------------------------------------------------- column-owner reduce + finalize
// One launch instead of (partial-row reduce with fp64 atomics -> finalize): a workgroup OWNS 16 channels, sums their partial rows
// itself (64 row lanes x 16 channels, 64-byte segments per row) and finalizes them.  No atomics, no second dependent launch,
// deterministic.  Used while the partial buffer is small (rows <= COLFIN_MAX_ROWS); the 208^2 / 416^2 layers keep the two-stage path.
constexpr int COLFIN_MAX_ROWS = 4096;
struct ColFinArgs {
  const float* partial; int rows, nsums, C; double count;
  // MODE 0: forward statistics -> scale / shift / running stats
  const float* gamma; const float* beta; float* rm; float* rv; float momentum, eps; float* scale; float* shift; float* mean; float* invstd;
  // MODE 1: backward sums -> dgamma, dbeta, coefficient vectors; BN #1 uses sums (0, 1), BN #2 (fused residual pair) sums (0, 2)
  const float* g1; const float* mean1; const float* is1; float* dg1; float* db1; float* cA1; float* cB1; float* cC1;
  const float* g2; const float* mean2; const float* is2; float* dg2; float* db2; float* cA2; float* cB2; float* cC2;
};

__device__ __forceinline__ void bwd_coeffs(double sg, double sgx, double count, float gamma, float mean, float invstd, float* dg, float* db,
                                           float* cA, float* cB, float* cC, int c) {
  dg[c] = (float)sgx;
  db[c] = (float)sg;
  const double g = gamma, is = invstd, mu = mean;
  const double mg = sg / count, mgx = sgx / count;
  cA[c] = (float)(g * is);
  cB[c] = (float)(-g * is * is * mgx);
  cC[c] = (float)(-g * is * mg + g * is * is * mu * mgx);
}

// More than COLFIN_MAX_ROWS partial rows (RektNet's 80x80 x 256-image tensors: 12 800 rows per layer; YOLOv3's 208^2 / 416^2 layers): fold them
// to COLFIN_FOLD_ROWS rows first, IN PLACE and without atomics: block (column group, j) sums the rows j, j + R, j + 2R, ... of its 16 columns
// and writes row j -- it is the only block that reads the rows it writes.  Fixed order -> deterministic.  (The earlier path, 128-row blocks +
// one fp64 atomic per column per block + a finalize launch, took 37 us per RektNet layer.)
constexpr int COLFIN_FOLD_ROWS = 64;
__global__ __launch_bounds__(1024) void rows_fold_kernel(float* __restrict__ partial, int rows, int cols) {
  __shared__ double red[64][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx, j = blockIdx.y;
  double s = 0.0;
  if (c < cols) {
#pragma unroll 4
    for (int r = j + COLFIN_FOLD_ROWS * ry; r < rows; r += COLFIN_FOLD_ROWS * 64) s += (double)partial[(size_t)r * cols + c];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry < 16) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) t += red[ry * 4 + k][cx];
    red[ry * 4][cx] = t;
  }
  __syncthreads();
  if (ry == 0 && c < cols) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 4][cx];
    partial[(size_t)j * cols + c] = (float)t;
  }
}

// rows -> at most COLFIN_MAX_ROWS rows (returns the new row count); `partial` is scratch of the caller and is consumed
static int fold_rows(float* partial, int rows, int cols, hipStream_t st) {
  if (rows <= COLFIN_MAX_ROWS) return rows;
  MDCV_LAUNCH(rows_fold_kernel, dim3((unsigned)cdiv(cols, 16), (unsigned)COLFIN_FOLD_ROWS), dim3(1024), 0, st, partial, rows, cols);
  return COLFIN_FOLD_ROWS;
}

template <int MODE>
__global__ __launch_bounds__(1024) void bn_colfinal_kernel(ColFinArgs a) {
  __shared__ double red[3][64][16];
  __shared__ double tot[3][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  if (c < a.C) {
    const size_t stride = (size_t)a.nsums * a.C;
    const float* p = a.partial + c;
    if (a.nsums == 1) {
#pragma unroll 4
      for (int r = ry; r < a.rows; r += 64) s0 += (double)p[r * stride];
    } else if (a.nsums == 2) {
#pragma unroll 4
      for (int r = ry; r < a.rows; r += 64) { s0 += (double)p[r * stride]; s1 += (double)p[r * stride + a.C]; }
    } else {
#pragma unroll 4
      for (int r = ry; r < a.rows; r += 64) { s0 += (double)p[r * stride]; s1 += (double)p[r * stride + a.C]; s2 += (double)p[r * stride + 2 * a.C]; }
    }
  }
  red[0][ry][cx] = s0; red[1][ry][cx] = s1; red[2][ry][cx] = s2;
  __syncthreads();
  if (ry < 3) {
    double t = 0.0;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) t += red[ry][k][cx];
    tot[ry][cx] = t;
  }
  __syncthreads();
  if (ry != 0 || c >= a.C) return;
  if (MODE == 0) {
    const double mean = tot[0][cx] / a.count;
    double var = tot[1][cx] / a.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float g = a.gamma[c], b = a.beta[c];
    a.scale[c] = g * invstd;
    a.shift[c] = b - (float)mean * g * invstd;
    a.mean[c] = (float)mean;
    a.invstd[c] = invstd;
    if (a.rm) {
      const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
      a.rm[c] = (1.f - a.momentum) * a.rm[c] + a.momentum * (float)mean;
      a.rv[c] = (1.f - a.momentum) * a.rv[c] + a.momentum * (float)unbiased;
    }
  } else if (MODE == 2) {
    a.scale[c] = (float)tot[0][cx];            // plain column sum (bias gradient)
  } else if (MODE == 3) {                      // sums from the fused data gradient: second sum is sum g*(y - mean), not yet / std
    bwd_coeffs(tot[0][cx], tot[1][cx] * (double)a.is1[c], a.count, a.g1[c], a.mean1[c], a.is1[c], a.dg1, a.db1, a.cA1, a.cB1, a.cC1, c);
  } else {
    bwd_coeffs(tot[0][cx], tot[1][cx], a.count, a.g1[c], a.mean1[c], a.is1[c], a.dg1, a.db1, a.cA1, a.cB1, a.cC1, c);
    if (a.nsums == 3) bwd_coeffs(tot[0][cx], tot[2][cx], a.count, a.g2[c], a.mean2[c], a.is2[c], a.dg2, a.db2, a.cA2, a.cB2, a.cC2, c);
  }
}

// backward, pass 2: dy_i = cA_i*g + cB_i*y_i + cC_i
// DUAL: two BatchNorms feed one activation (RektNet's residual blocks).  A template parameter, not a runtime test of y2: the single form
// then carries 40 coefficient registers instead of 80 (214 -> ~120 VGPRs), so four blocks fit on a CU instead of two -- the kernel shares
// the chip with the side stream's weight gradients, whose blocks leave no registers on the CUs they occupy.
template <typename T, bool DUAL>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(BnBwdArgs a) {
  constexpr int VEC = ET<T>::VEC;
  constexpr int ND = DUAL ? VEC : 1;
  const int tid = threadIdx.x;
  if (tid >= a.PPI * a.CV) return;
  const int cv = tid % a.CV, pi = tid / a.CV;
  float s1[VEC], b1[VEC], A1[VEC], B1[VEC], C1[VEC], s2[ND], b2[ND], A2[ND], B2[ND], C2[ND];
  ldcoef<VEC>(a.s1, cv * VEC, s1, 1.f); ldcoef<VEC>(a.b1, cv * VEC, b1, 0.f);
  ldcoef<VEC>(a.cA1, cv * VEC, A1, 0.f); ldcoef<VEC>(a.cB1, cv * VEC, B1, 0.f); ldcoef<VEC>(a.cC1, cv * VEC, C1, 0.f);
  if constexpr (DUAL) {
    ldcoef<VEC>(a.s2, cv * VEC, s2, 0.f); ldcoef<VEC>(a.b2, cv * VEC, b2, 0.f);
    ldcoef<VEC>(a.cA2, cv * VEC, A2, 0.f); ldcoef<VEC>(a.cB2, cv * VEC, B2, 0.f); ldcoef<VEC>(a.cC2, cv * VEC, C2, 0.f);
  }
  const long long p0 = (long long)blockIdx.x * a.PB;
  const long long p1 = min((long long)a.M, p0 + a.PB);
  const T* dout = reinterpret_cast<const T*>(a.dout);
  const T* y1 = reinterpret_cast<const T*>(a.y1);
  const T* y2 = reinterpret_cast<const T*>(a.y2);
  T* dy1 = reinterpret_cast<T*>(a.dy1);
  T* dy2 = reinterpret_cast<T*>(a.dy2);
  for (long long pb = p0 + pi; pb < p1; pb += 4 * a.PPI) {
    uint4 qd[4], qv[4], qw[DUAL ? 4 : 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long p = pb + (long long)u * a.PPI;
      if (p < p1) {
        qd[u] = ld_stream(dout + p * a.ldd + cv * VEC);
        qv[u] = ld_stream(y1 + p * a.ld1 + cv * VEC);
        if constexpr (DUAL) qw[u] = ld_stream(y2 + p * a.ld2 + cv * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long p = pb + (long long)u * a.PPI;
      if (p >= p1) break;
      float d[VEC], v[VEC], w[ND], o1[VEC], o2[ND];
      ET<T>::unpack(qd[u], d);
      ET<T>::unpack(qv[u], v);
      if constexpr (DUAL) ET<T>::unpack(qw[u], w);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if constexpr (!DUAL) {
          o1[e] = mdcv_bn_bwd_dy(d[e], v[e], s1[e], b1[e], A1[e], B1[e], C1[e], a.act, a.slope);     // (shared with the operand-load forms: bit-identical)
        } else {
          float pre = v[e] * s1[e] + b1[e];
          pre += w[e] * s2[e] + b2[e];
          const float g = d[e] * act_grad(pre, a.act, a.slope);
          o1[e] = A1[e] * g + B1[e] * v[e] + C1[e];
          o2[e] = A2[e] * g + B2[e] * w[e] + C2[e];
        }
      }
      *reinterpret_cast<uint4*>(dy1 + p * a.ldy1 + cv * VEC) = ET<T>::pack(o1);
      if constexpr (DUAL) *reinterpret_cast<uint4*>(dy2 + p * a.ldy2 + cv * VEC) = ET<T>::pack(o2);
    }
  }
}

// ---------------------------------------------------------------- per-channel column sum (bias gradients of BN-less convs)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ldc, int M, int C, double* __restrict__ accum, int PB, int CV, int PPI) {
  constexpr int VEC = ET<T>::VEC;
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const bool active = tid < PPI * CV;
  const int cv = active ? tid % CV : 0, pi = active ? tid / CV : 0;
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  const long long p0 = (long long)blockIdx.x * PB, p1 = min((long long)M, p0 + PB);
  if (active)
    for (long long p = p0 + pi; p < p1; p += PPI) {
      float v[VEC];
      ET<T>::unpack(*reinterpret_cast<const uint4*>(x + p * ldc + cv * VEC), v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[tid * 8 + e] = s[e];
  __syncthreads();
  if (active && pi == 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = 0.f;
      for (int q = 0; q < PPI; ++q) t += red[(q * CV + cv) * 8 + e];
      atomicAdd(&accum[cv * VEC + e], (double)t);
    }
  }
}
// same strip walk, but every block stores its [C] partial row (no atomics); bn_colfinal_kernel<2> sums the rows
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* __restrict__ x, int ldc, int M, int C, float* __restrict__ partial, int PB, int CV, int PPI) {
  constexpr int VEC = ET<T>::VEC;
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const bool active = tid < PPI * CV;
  const int cv = active ? tid % CV : 0, pi = active ? tid / CV : 0;
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  const long long p0 = (long long)blockIdx.x * PB, p1 = min((long long)M, p0 + PB);
  if (active)
    for (long long p = p0 + pi; p < p1; p += PPI) {
      float v[VEC];
      ET<T>::unpack(*reinterpret_cast<const uint4*>(x + p * ldc + cv * VEC), v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[tid * 8 + e] = s[e];
  __syncthreads();
  if (active && pi == 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = 0.f;
      for (int q = 0; q < PPI; ++q) t += red[(q * CV + cv) * 8 + e];
      partial[(size_t)blockIdx.x * C + cv * VEC + e] = t;
    }
  }
}
__global__ void accum_to_f32_kernel(double* __restrict__ accum, float* __restrict__ out, int n, int zero_after) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = (float)accum[i];
  if (zero_after) accum[i] = 0.0;
}

// ---------------------------------------------------------------- nearest x2 upsample
template <typename T>
__global__ void upsample2x_fwd_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo, int B, int H, int W, int C) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const long long total = (long long)B * H * W * 4 * CV;       // one thread per OUTPUT vector
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long op = i / CV;
    const int W2 = 2 * W, H2 = 2 * H;
    const int ow = (int)(op % W2);
    const long long t = op / W2;
    const int oh = (int)(t % H2), b = (int)(t / H2);
    const long long ip = ((long long)b * H + (oh >> 1)) * W + (ow >> 1);
    *reinterpret_cast<uint4*>(out + op * ldo + cv * VEC) = *reinterpret_cast<const uint4*>(in + ip * ldi + cv * VEC);
  }
}
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dout, int ldo, T* __restrict__ din, int ldi, int B, int H, int W, int C) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const long long total = (long long)B * H * W * CV;           // one thread per INPUT vector: sum of its 2x2 outputs
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long ip = i / CV;
    const int w = (int)(ip % W);
    const long long t = ip / W;
    const int h = (int)(t % H), b = (int)(t / H);
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const long long op = ((long long)b * 2 * H + 2 * h + dy) * 2 * W + 2 * w + dx;
        float v[VEC];
        ET<T>::unpack(*reinterpret_cast<const uint4*>(dout + op * ldo + cv * VEC), v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += v[e];
      }
    *reinterpret_cast<uint4*>(din + ip * ldi + cv * VEC) = ET<T>::pack(s);
  }
}

// ---------------------------------------------------------------- 2x2 max-pool (yolo_baseline_tiny.cfg): stride 2, or stride 1 on a
// bottom/right zero-padded input (nn.ZeroPad2d((0,1,0,1)) + nn.MaxPool2d(2,1), reference models.py:74-84).  idx = winning window
// position (kh*2+kw; first maximum wins like torch; 4 = the zero padding won) so that backward is a pure gather.
template <typename T>
__global__ void maxpool2x2_fwd_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo, unsigned char* __restrict__ idx,
                                      int B, int H, int W, int C, int stride) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const int Ho = stride == 2 ? H / 2 : H, Wo = stride == 2 ? W / 2 : W;
  const long long total = (long long)B * Ho * Wo * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long op = i / CV;
    const int ow = (int)(op % Wo);
    const long long t = op / Wo;
    const int oh = (int)(t % Ho), b = (int)(t / Ho);
    float best[VEC]; unsigned char bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int h = oh * stride + (k >> 1), w = ow * stride + (k & 1);
      float v[VEC];
      const bool inside = h < H && w < W;
      if (inside) ET<T>::unpack(*reinterpret_cast<const uint4*>(in + (((long long)b * H + h) * W + w) * ldi + cv * VEC), v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float x = inside ? v[e] : 0.f;                    // zero padding takes part in the max
        if (x > best[e]) { best[e] = x; bi[e] = inside ? (unsigned char)k : (unsigned char)4; }
      }
    }
    *reinterpret_cast<uint4*>(out + op * ldo + cv * VEC) = ET<T>::pack(best);
#pragma unroll
    for (int e = 0; e < VEC; ++e) idx[op * C + cv * VEC + e] = bi[e];
  }
}
template <typename T>
__global__ void maxpool2x2_bwd_kernel(const T* __restrict__ dout, int ldo, const unsigned char* __restrict__ idx, T* __restrict__ din, int ldi,
                                      int B, int H, int W, int C, int stride) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const int Ho = stride == 2 ? H / 2 : H, Wo = stride == 2 ? W / 2 : W;
  const long long total = (long long)B * H * W * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long ip = i / CV;
    const int w = (int)(ip % W);
    const long long t = ip / W;
    const int h = (int)(t % H), b = (int)(t / H);
    float g[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = 0.f;
    const int nwin = stride == 2 ? 1 : 4;
    for (int q = 0; q < nwin; ++q) {
      int oh, ow, pos;
      if (stride == 2) { oh = h >> 1; ow = w >> 1; pos = (h & 1) * 2 + (w & 1); }
      else { oh = h - (q >> 1); ow = w - (q & 1); pos = (q >> 1) * 2 + (q & 1); }
      if (oh < 0 || ow < 0 || oh >= Ho || ow >= Wo) continue;
      const long long op = ((long long)b * Ho + oh) * Wo + ow;
      float d[VEC];
      ET<T>::unpack(*reinterpret_cast<const uint4*>(dout + op * ldo + cv * VEC), d);
#pragma unroll
      for (int e = 0; e < VEC; ++e) if (idx[op * C + cv * VEC + e] == pos) g[e] += d[e];
    }
    *reinterpret_cast<uint4*>(din + ip * ldi + cv * VEC) = ET<T>::pack(g);
  }
}

// ---------------------------------------------------------------- generic max-pool and nearest upsample (reference models.py:74-88 builds
// nn.MaxPool2d(size, stride, (size - 1) // 2) and nn.Upsample(scale_factor = stride) for ANY size / stride; the bundled cfgs only use the
// 2x2 pools and the x2 upsample above).  Padding never wins (-inf); the first maximum in (kh, kw) scan order wins like torch;
// idx = kh*k + kw (k <= 15), so backward is a gather over the windows that contain the input pixel: no atomics, deterministic.
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo, unsigned char* __restrict__ idx,
                                   int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const long long total = (long long)B * Ho * Wo * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long op = i / CV;
    const int ow = (int)(op % Wo);
    const long long t = op / Wo;
    const int oh = (int)(t % Ho), b = (int)(t / Ho);
    float best[VEC]; unsigned char bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    for (int kh = 0; kh < k; ++kh) {
      const int h = oh * stride - pad + kh;
      if (h < 0 || h >= H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int w = ow * stride - pad + kw;
        if (w < 0 || w >= W) continue;
        float v[VEC];
        ET<T>::unpack(*reinterpret_cast<const uint4*>(in + (((long long)b * H + h) * W + w) * ldi + cv * VEC), v);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (v[e] > best[e]) { best[e] = v[e]; bi[e] = (unsigned char)(kh * k + kw); }
      }
    }
    *reinterpret_cast<uint4*>(out + op * ldo + cv * VEC) = ET<T>::pack(best);
#pragma unroll
    for (int e = 0; e < VEC; ++e) idx[op * C + cv * VEC + e] = bi[e];
  }
}
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dout, int ldo, const unsigned char* __restrict__ idx, T* __restrict__ din, int ldi,
                                   int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const long long total = (long long)B * H * W * CV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long ip = i / CV;
    const int w = (int)(ip % W);
    const long long t = ip / W;
    const int h = (int)(t % H), b = (int)(t / H);
    float g[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = 0.f;
    // windows that contain (h, w): oh*stride - pad <= h <= oh*stride - pad + k - 1
    int oh0 = h + pad - k + 1; oh0 = oh0 > 0 ? (oh0 + stride - 1) / stride : 0;
    int ow0 = w + pad - k + 1; ow0 = ow0 > 0 ? (ow0 + stride - 1) / stride : 0;
    const int oh1 = min(Ho - 1, (h + pad) / stride), ow1 = min(Wo - 1, (w + pad) / stride);
    for (int oh = oh0; oh <= oh1; ++oh)
      for (int ow = ow0; ow <= ow1; ++ow) {
        const int pos = (h + pad - oh * stride) * k + (w + pad - ow * stride);
        const long long op = ((long long)b * Ho + oh) * Wo + ow;
        float d[VEC];
        ET<T>::unpack(*reinterpret_cast<const uint4*>(dout + op * ldo + cv * VEC), d);
#pragma unroll
        for (int e = 0; e < VEC; ++e) if (idx[op * C + cv * VEC + e] == pos) g[e] += d[e];
      }
    *reinterpret_cast<uint4*>(din + ip * ldi + cv * VEC) = ET<T>::pack(g);
  }
}
template <typename T>
__global__ void upsample_fwd_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo, int B, int H, int W, int C, int sc) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const int Ws = W * sc, Hs = H * sc;
  const long long total = (long long)B * Hs * Ws * CV;         // one thread per OUTPUT vector
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long op = i / CV;
    const int ow = (int)(op % Ws);
    const long long t = op / Ws;
    const int oh = (int)(t % Hs), b = (int)(t / Hs);
    const long long ip = ((long long)b * H + oh / sc) * W + ow / sc;
    *reinterpret_cast<uint4*>(out + op * ldo + cv * VEC) = *reinterpret_cast<const uint4*>(in + ip * ldi + cv * VEC);
  }
}
template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dout, int ldo, T* __restrict__ din, int ldi, int B, int H, int W, int C, int sc) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = C / VEC;
  const long long total = (long long)B * H * W * CV;           // one thread per INPUT vector: sum of its sc x sc outputs (fixed order)
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const long long ip = i / CV;
    const int w = (int)(ip % W);
    const long long t = ip / W;
    const int h = (int)(t % H), b = (int)(t / H);
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    for (int dy = 0; dy < sc; ++dy)
      for (int dx = 0; dx < sc; ++dx) {
        const long long op = ((long long)b * sc * H + sc * h + dy) * sc * W + sc * w + dx;
        float v[VEC];
        ET<T>::unpack(*reinterpret_cast<const uint4*>(dout + op * ldo + cv * VEC), v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += v[e];
      }
    *reinterpret_cast<uint4*>(din + ip * ldi + cv * VEC) = ET<T>::pack(s);
  }
}

static unsigned ew_grid(long long total) {
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" {

int mdcv_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int ldc, int Cpad, void* stream) {
  if (!src || !dst || (Cpad & 7) || (ldc & 7) || Cpad < C) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_grid((long long)B * H * W * Cpad / 8)), dim3(256), 0, st, src, (bf16_t*)dst, B, C, H, W, ldc, Cpad);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(nchw_to_nhwc_kernel<float>, dim3(ew_grid((long long)B * H * W * Cpad / 4)), dim3(256), 0, st, src, (float*)dst, B, C, H, W, ldc, Cpad);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_nhwc_to_nchw(int dtype, const void* src, int ldc, float* dst, int B, int C, int H, int W, void* stream) {
  if (!src || !dst) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = ew_grid((long long)B * C * H * W);
  if (dtype == MDCV_BF16) MDCV_LAUNCH(nhwc_to_nchw_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)src, dst, B, C, H, W, ldc);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(nhwc_to_nchw_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)src, dst, B, C, H, W, ldc);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_partial_reduce(const float* partial, int rows, int nsums, int C, double* accum, void* stream) {
  if (!partial || !accum || rows < 1) return MDCV_EARG;
  const int cols = nsums * C;
  MDCV_LAUNCH(partial_reduce_kernel, dim3((unsigned)cdiv(cols, 64), (unsigned)cdiv(rows, 128)), dim3(256), 0, (hipStream_t)stream,
                     partial, rows, cols, accum);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_bn_finalize(double* accum, double count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps, float* scale, float* shift, float* mean, float* invstd, int C, void* stream) {
  if (!accum || !gamma || !beta || !scale || !shift || !mean || !invstd) return MDCV_EARG;
  MDCV_LAUNCH(bn_finalize_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, accum, count, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, mean, invstd, C);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// conv-epilogue partial rows -> batch statistics -> scale / shift (+ running stats): one launch while the partial buffer is small
int mdcv_bn_stats_finalize(const float* partial, int rows, double* accum, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                           float* invstd, int C, void* stream) {
  if (!partial || !accum || rows < 1 || !gamma || !beta || !scale || !shift || !mean || !invstd) return MDCV_EARG;
  rows = fold_rows(const_cast<float*>(partial), rows, 2 * C, (hipStream_t)stream);      // (large buffers: folded in place first)
  MDCV_CHECK_LAUNCH();
  ColFinArgs a = {};
  a.partial = partial; a.rows = rows; a.nsums = 2; a.C = C; a.count = count; a.gamma = gamma; a.beta = beta; a.rm = running_mean;
  a.rv = running_var; a.momentum = momentum; a.eps = eps; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
  MDCV_LAUNCH(bn_colfinal_kernel<0>, dim3((unsigned)cdiv(C, 16)), dim3(1024), 0, (hipStream_t)stream, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// BatchNorm(batch statistics) + activation (+ residual) with the statistics taken from the exact accumulators mdcv_conv2d_xstats /
// mdcv_pw_conv_fwd_xstats added to (`count` positions; [reps][3][2][C] words).  Writes scale / shift / mean / invstd and updates the running
// statistics as mdcv_bn_stats_finalize does.  C <= 1024.  mdcv_xstats_reps: the replica count to use for a layer with `rows` additions per word.
constexpr int kXaccBlocks = 512;      // fewer, longer strips than mdcv_bn_act_fwd: every workgroup pays the prologue
int mdcv_xstats_reps(int rows, int C) {
  int cp = 32; while (cp < C && cp < 256) cp <<= 1;
  const int cap = C <= 256 ? 1024 / cp : (C <= 512 ? 2 : 1);
  int r = 1; while (r < cap && r * 128 < rows) r <<= 1;
  return r;
}
int mdcv_bn_act_fwd_xstats(int dtype, const void* y, int ldy, const void* xacc, int reps, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                           float* invstd, const void* resid, int ldr, void* out, int ldo, int M, int C, int act, float slope, void* stream) {
  if (!y || !out || !xacc || !gamma || !beta || !scale || !shift || !mean || !invstd || reps < 1 || (reps & (reps - 1)) || (C & 7) || C > 1024 ||
      (ldy & 7) || (ldo & 7) || (dtype != MDCV_BF16 && dtype != MDCV_F32))
    return MDCV_EARG;
  BnActArgs a;
  a.y1 = y; a.y2 = nullptr; a.resid = resid; a.out = out; a.s1 = nullptr; a.b1 = nullptr; a.s2 = nullptr; a.b2 = nullptr;
  a.ld1 = ldy; a.ld2 = 0; a.ldr = ldr; a.ldo = ldo; a.M = M; a.C = C; a.act = act; a.slope = act == 2 ? 0.f : slope;
  BnXAccArgs f{reinterpret_cast<const long long*>(xacc), reps, count, 1.0 / count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd, 256, reps};
  const int nch = C <= 256 ? 1 : (C <= 512 ? 2 : 4);
  if (nch == 1) {
    int cp = 32; while (cp < C) cp <<= 1;
    f.Cp = cp; f.rp = reps / (256 / cp) > 0 ? reps / (256 / cp) : 1;
  }
  if (nch * f.rp > 4) return MDCV_EARG;                    // (the prologue keeps a thread's words in registers: mdcv_xstats_reps stays inside)
  const Strip s = dtype == MDCV_BF16 ? make_strip<bf16_t>(M, C, kXaccBlocks, 4) : make_strip<float>(M, C, kXaccBlocks, 4);
  if (s.CV > 256) return MDCV_EARG;
  a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI;
  const dim3 grid((unsigned)cdiv(M, s.PB) + 1);             // + the publishing workgroup
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16) {
    if (nch == 1) MDCV_LAUNCH((bn_act_fwd_xacc_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a, f);
    else if (nch == 2) MDCV_LAUNCH((bn_act_fwd_xacc_kernel<bf16_t, 2>), grid, dim3(256), 0, st, a, f);
    else MDCV_LAUNCH((bn_act_fwd_xacc_kernel<bf16_t, 4>), grid, dim3(256), 0, st, a, f);
  } else {
    if (nch == 1) MDCV_LAUNCH((bn_act_fwd_xacc_kernel<float, 1>), grid, dim3(256), 0, st, a, f);
    else if (nch == 2) MDCV_LAUNCH((bn_act_fwd_xacc_kernel<float, 2>), grid, dim3(256), 0, st, a, f);
    else MDCV_LAUNCH((bn_act_fwd_xacc_kernel<float, 4>), grid, dim3(256), 0, st, a, f);
  }
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                        float* scale, float* shift, int C, void* stream) {
  MDCV_LAUNCH(bn_eval_coeffs_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, scale, shift, C);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_bn_eval_coeffs_bias(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                             const float* conv_bias, float* scale, float* shift, int C, void* stream) {
  MDCV_LAUNCH(bn_eval_coeffs_bias_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, conv_bias, scale, shift, C);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// out = act(y1*s1+b1 [+ y2*s2+b2]) [+ resid] ; s1 == NULL means identity on y1 (plain add / copy)
int mdcv_bn_act_fwd(int dtype, const void* y1, int ld1, const float* s1, const float* b1, const void* y2, int ld2, const float* s2,
                    const float* b2, const void* resid, int ldr, void* out, int ldo, int M, int C, int act, float slope, void* stream) {
  if (!y1 || !out || (C & 7) || (ld1 & 7) || (ldo & 7)) return MDCV_EARG;
  BnActArgs a;
  a.y1 = y1; a.y2 = y2; a.resid = resid; a.out = out; a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2;
  a.ld1 = ld1; a.ld2 = ld2; a.ldr = ldr; a.ldo = ldo; a.M = M; a.C = C; a.act = act; a.slope = act == 2 ? 0.f : slope;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16) {
    Strip s = make_strip<bf16_t>(M, C, 2048, 4); if (s.CV > 256) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI;
    MDCV_LAUNCH(bn_act_fwd_kernel<bf16_t>, dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, a);
  } else if (dtype == MDCV_F32) {
    Strip s = make_strip<float>(M, C, 2048, 4); if (s.CV > 256) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI;
    MDCV_LAUNCH(bn_act_fwd_kernel<float>, dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, a);
  } else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// floats of scratch mdcv_bn_act_bwd_reduce needs (one [nsums][C] partial row per block)
int mdcv_bn_act_bwd_reduce_ws_floats(int dtype, int M, int C, int nsums) {
  const Strip s = dtype == MDCV_BF16 ? make_strip<bf16_t>(M, C, reduce_blocks(C, nsums), 8) : make_strip<float>(M, C, reduce_blocks(C, nsums), 8);
  return cdiv(M, s.PB) * nsums * C;
}

// pass 1 of the BN(+act) backward: accum[0] += sum g ; accum[1] += sum g*xhat1 ; accum[2] += sum g*xhat2 (if y2)
int mdcv_bn_act_bwd_reduce(int dtype, const void* dout, int ldd, const void* y1, int ld1, const float* s1, const float* b1,
                           const float* mean1, const float* invstd1, const void* y2, int ld2, const float* s2, const float* b2,
                           const float* mean2, const float* invstd2, double* accum, float* partial_ws, int M, int C, int act, float slope,
                           void* stream) {
  if (!dout || !y1 || !accum || !partial_ws || (C & 7)) return MDCV_EARG;
  BnBwdArgs a = {};
  a.dout = dout; a.y1 = y1; a.y2 = y2; a.s1 = s1; a.b1 = b1; a.m1 = mean1; a.is1 = invstd1; a.s2 = s2; a.b2 = b2; a.m2 = mean2; a.is2 = invstd2;
  a.accum = accum; a.partial = partial_ws; a.ldd = ldd; a.ld1 = ld1; a.ld2 = ld2; a.M = M; a.C = C; a.act = act; a.slope = act == 2 ? 0.f : slope;
  hipStream_t st = (hipStream_t)stream;
  const int nsums = y2 ? 3 : 2;
  int rows = 0;
  if (dtype == MDCV_BF16) {
    Strip s = make_strip<bf16_t>(M, C, reduce_blocks(C, nsums), 8); if (s.CV > 256 || (256 % s.CV)) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI; rows = cdiv(M, s.PB);
    if (a.y2) MDCV_LAUNCH((bn_act_bwd_reduce_kernel<bf16_t, true>), dim3((unsigned)rows), dim3(256), 0, st, a);
    else MDCV_LAUNCH((bn_act_bwd_reduce_kernel<bf16_t, false>), dim3((unsigned)rows), dim3(256), 0, st, a);
  } else if (dtype == MDCV_F32) {
    Strip s = make_strip<float>(M, C, reduce_blocks(C, nsums), 8); if (s.CV > 256 || (256 % s.CV)) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI; rows = cdiv(M, s.PB);
    if (a.y2) MDCV_LAUNCH((bn_act_bwd_reduce_kernel<float, true>), dim3((unsigned)rows), dim3(256), 0, st, a);
    else MDCV_LAUNCH((bn_act_bwd_reduce_kernel<float, false>), dim3((unsigned)rows), dim3(256), 0, st, a);
  } else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  const int cols = nsums * C;
  MDCV_LAUNCH(partial_reduce_kernel, dim3((unsigned)cdiv(cols, 64), (unsigned)cdiv(rows, 128)), dim3(256), 0, st, partial_ws, rows, cols, accum);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_bn_bwd_finalize(double* accum, int kx, int nsums, int zero_after, double count, const float* gamma, const float* mean,
                         const float* invstd, float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int C, void* stream) {
  if (!accum || !gamma || !dgamma || !dbeta || !cA || !cB || !cC) return MDCV_EARG;
  MDCV_LAUNCH(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, accum, kx, zero_after, count,
                     gamma, mean, invstd, dgamma, dbeta, cA, cB, cC, C, nsums);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// pass 1 of the BN(+act) backward including the finalize: partial rows -> (dgamma, dbeta, cA, cB, cC) of one or two BatchNorms
// in the same launch sequence (main reduce kernel + one column-owner kernel; no atomics, nothing left in `accum`).
int mdcv_bn_act_bwd_reduce_finalize(int dtype, const void* dout, int ldd, const void* y1, int ld1, const float* s1, const float* b1,
                                    const float* mean1, const float* invstd1, const void* y2, int ld2, const float* s2, const float* b2,
                                    const float* mean2, const float* invstd2, float* partial_ws, int M, int C, int act, float slope,
                                    double count, const float* gamma1, float* dgamma1, float* dbeta1, float* cA1, float* cB1, float* cC1,
                                    const float* gamma2, float* dgamma2, float* dbeta2, float* cA2, float* cB2, float* cC2, void* stream) {
  if (!dout || !y1 || !partial_ws || (C & 7) || !gamma1 || !dgamma1 || !dbeta1 || !cA1 || !cB1 || !cC1 || !mean1 || !invstd1) return MDCV_EARG;
  if (y2 && (!gamma2 || !dgamma2 || !dbeta2 || !cA2 || !cB2 || !cC2 || !mean2 || !invstd2)) return MDCV_EARG;
  BnBwdArgs a = {};
  a.dout = dout; a.y1 = y1; a.y2 = y2; a.s1 = s1; a.b1 = b1; a.m1 = mean1; a.is1 = invstd1; a.s2 = s2; a.b2 = b2; a.m2 = mean2; a.is2 = invstd2;
  a.accum = nullptr; a.partial = partial_ws; a.ldd = ldd; a.ld1 = ld1; a.ld2 = ld2; a.M = M; a.C = C; a.act = act; a.slope = act == 2 ? 0.f : slope;
  hipStream_t st = (hipStream_t)stream;
  const int nsums = y2 ? 3 : 2;
  int rows = 0;
  if (dtype == MDCV_BF16) {
    Strip s = make_strip<bf16_t>(M, C, reduce_blocks(C, nsums), 8); if (s.CV > 256 || (256 % s.CV)) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI; rows = cdiv(M, s.PB);
    if (a.y2) MDCV_LAUNCH((bn_act_bwd_reduce_kernel<bf16_t, true>), dim3((unsigned)rows), dim3(256), 0, st, a);
    else MDCV_LAUNCH((bn_act_bwd_reduce_kernel<bf16_t, false>), dim3((unsigned)rows), dim3(256), 0, st, a);
  } else if (dtype == MDCV_F32) {
    Strip s = make_strip<float>(M, C, reduce_blocks(C, nsums), 8); if (s.CV > 256 || (256 % s.CV)) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI; rows = cdiv(M, s.PB);
    if (a.y2) MDCV_LAUNCH((bn_act_bwd_reduce_kernel<float, true>), dim3((unsigned)rows), dim3(256), 0, st, a);
    else MDCV_LAUNCH((bn_act_bwd_reduce_kernel<float, false>), dim3((unsigned)rows), dim3(256), 0, st, a);
  } else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  ColFinArgs f = {};
  f.partial = partial_ws; f.rows = rows; f.nsums = nsums; f.C = C; f.count = count;
  f.g1 = gamma1; f.mean1 = mean1; f.is1 = invstd1; f.dg1 = dgamma1; f.db1 = dbeta1; f.cA1 = cA1; f.cB1 = cB1; f.cC1 = cC1;
  f.g2 = gamma2; f.mean2 = mean2; f.is2 = invstd2; f.dg2 = dgamma2; f.db2 = dbeta2; f.cA2 = cA2; f.cB2 = cB2; f.cC2 = cC2;
  MDCV_LAUNCH(bn_colfinal_kernel<1>, dim3((unsigned)cdiv(C, 16)), dim3(1024), 0, st, f);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_bn_bwd_finalize_rows(const float* partial, int rows, int C, double count, const float* gamma, const float* mean,
                              const float* invstd, float* dgamma, float* dbeta, float* cA, float* cB, float* cC, void* stream) {
  if (!partial || rows < 1 || !gamma || !mean || !invstd || !dgamma || !dbeta || !cA || !cB || !cC) return MDCV_EARG;
  rows = fold_rows(const_cast<float*>(partial), rows, 2 * C, (hipStream_t)stream);
  MDCV_CHECK_LAUNCH();
  ColFinArgs f = {};
  f.partial = partial; f.rows = rows; f.nsums = 2; f.C = C; f.count = count;
  f.g1 = gamma; f.mean1 = mean; f.is1 = invstd; f.dg1 = dgamma; f.db1 = dbeta; f.cA1 = cA; f.cB1 = cB; f.cC1 = cC;
  MDCV_LAUNCH(bn_colfinal_kernel<3>, dim3((unsigned)cdiv(C, 16)), dim3(1024), 0, (hipStream_t)stream, f);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_bn_act_bwd_apply(int dtype, const void* dout, int ldd, const void* y1, int ld1, const float* s1, const float* b1,
                          const float* cA1, const float* cB1, const float* cC1, void* dy1, int ldy1,
                          const void* y2, int ld2, const float* s2, const float* b2, const float* cA2, const float* cB2,
                          const float* cC2, void* dy2, int ldy2, int M, int C, int act, float slope, void* stream) {
  if (!dout || !y1 || !dy1 || (C & 7)) return MDCV_EARG;
  BnBwdArgs a = {};
  a.dout = dout; a.y1 = y1; a.y2 = y2; a.dy1 = dy1; a.dy2 = dy2; a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2;
  a.cA1 = cA1; a.cB1 = cB1; a.cC1 = cC1; a.cA2 = cA2; a.cB2 = cB2; a.cC2 = cC2;
  a.ldd = ldd; a.ld1 = ld1; a.ld2 = ld2; a.ldy1 = ldy1; a.ldy2 = ldy2; a.M = M; a.C = C; a.act = act; a.slope = act == 2 ? 0.f : slope;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16) {
    Strip s = make_strip<bf16_t>(M, C, 2048, 4); if (s.CV > 256) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI;
    if (a.y2) MDCV_LAUNCH((bn_act_bwd_apply_kernel<bf16_t, true>), dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, a);
    else MDCV_LAUNCH((bn_act_bwd_apply_kernel<bf16_t, false>), dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, a);
  } else if (dtype == MDCV_F32) {
    Strip s = make_strip<float>(M, C, 2048, 4); if (s.CV > 256) return MDCV_EARG;
    a.PB = s.PB; a.CV = s.CV; a.PPI = s.PPI;
    if (a.y2) MDCV_LAUNCH((bn_act_bwd_apply_kernel<float, true>), dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, a);
    else MDCV_LAUNCH((bn_act_bwd_apply_kernel<float, false>), dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, a);
  } else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_colsum(int dtype, const void* x, int ldc, int M, int C, double* accum, void* stream) {
  if (!x || !accum || (C & 7)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16) {
    Strip s = make_strip<bf16_t>(M, C, reduce_blocks(C, 1)); if (s.CV > 256) return MDCV_EARG;
    MDCV_LAUNCH(colsum_kernel<bf16_t>, dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, (const bf16_t*)x, ldc, M, C, accum, s.PB, s.CV, s.PPI);
  } else if (dtype == MDCV_F32) {
    Strip s = make_strip<float>(M, C, reduce_blocks(C, 1)); if (s.CV > 256) return MDCV_EARG;
    MDCV_LAUNCH(colsum_kernel<float>, dim3((unsigned)cdiv(M, s.PB)), dim3(256), 0, st, (const float*)x, ldc, M, C, accum, s.PB, s.CV, s.PPI);
  } else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// column sums without atomics: per-block partial rows (partial_ws: mdcv_colsum_ws_floats floats) + one column-owner launch
int mdcv_colsum_ws_floats(int dtype, int M, int C) {
  const Strip s = dtype == MDCV_BF16 ? make_strip<bf16_t>(M, C, reduce_blocks(C, 1)) : make_strip<float>(M, C, reduce_blocks(C, 1));
  return cdiv(M, s.PB) * C;
}
int mdcv_colsum_f32(int dtype, const void* x, int ldc, int M, int C, float* partial_ws, float* out, void* stream) {
  if (!x || !partial_ws || !out || (C & 7)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  int rows = 0;
  if (dtype == MDCV_BF16) {
    Strip s = make_strip<bf16_t>(M, C, reduce_blocks(C, 1)); if (s.CV > 256) return MDCV_EARG;
    rows = cdiv(M, s.PB);
    MDCV_LAUNCH(colsum_rows_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)x, ldc, M, C, partial_ws, s.PB, s.CV, s.PPI);
  } else if (dtype == MDCV_F32) {
    Strip s = make_strip<float>(M, C, reduce_blocks(C, 1)); if (s.CV > 256) return MDCV_EARG;
    rows = cdiv(M, s.PB);
    MDCV_LAUNCH(colsum_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)x, ldc, M, C, partial_ws, s.PB, s.CV, s.PPI);
  } else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  ColFinArgs f = {};
  f.partial = partial_ws; f.rows = rows; f.nsums = 1; f.C = C; f.count = 1.0; f.scale = out;
  MDCV_LAUNCH(bn_colfinal_kernel<2>, dim3((unsigned)cdiv(C, 16)), dim3(1024), 0, st, f);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_accum_to_f32(double* accum, float* out, int n, int zero_after, void* stream) {
  MDCV_LAUNCH(accum_to_f32_kernel, dim3((unsigned)cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, accum, out, n, zero_after);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_upsample2x_fwd(int dtype, const void* in, int ldi, void* out, int ldo, int B, int H, int W, int C, void* stream) {
  if (!in || !out || (C & 7)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(upsample2x_fwd_kernel<bf16_t>, dim3(ew_grid((long long)B * H * W * 4 * C / 8)), dim3(256), 0, st, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, B, H, W, C);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(upsample2x_fwd_kernel<float>, dim3(ew_grid((long long)B * H * W * 4 * C / 4)), dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, B, H, W, C);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_upsample2x_bwd(int dtype, const void* dout, int ldo, void* din, int ldi, int B, int H, int W, int C, void* stream) {
  if (!dout || !din || (C & 7)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(upsample2x_bwd_kernel<bf16_t>, dim3(ew_grid((long long)B * H * W * C / 8)), dim3(256), 0, st, (const bf16_t*)dout, ldo, (bf16_t*)din, ldi, B, H, W, C);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(upsample2x_bwd_kernel<float>, dim3(ew_grid((long long)B * H * W * C / 4)), dim3(256), 0, st, (const float*)dout, ldo, (float*)din, ldi, B, H, W, C);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_maxpool2x2_fwd(int dtype, const void* in, int ldi, void* out, int ldo, unsigned char* idx, int B, int H, int W, int C, int stride,
                        void* stream) {
  if (!in || !out || !idx || (C & 7) || (stride != 1 && stride != 2)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * (stride == 2 ? H / 2 : H) * (stride == 2 ? W / 2 : W) * C;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(maxpool2x2_fwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, idx, B, H, W, C, stride);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(maxpool2x2_fwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, idx, B, H, W, C, stride);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_maxpool2x2_bwd(int dtype, const void* dout, int ldo, const unsigned char* idx, void* din, int ldi, int B, int H, int W, int C, int stride,
                        void* stream) {
  if (!dout || !din || !idx || (C & 7) || (stride != 1 && stride != 2)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * H * W * C;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(maxpool2x2_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)dout, ldo, idx, (bf16_t*)din, ldi, B, H, W, C, stride);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(maxpool2x2_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)dout, ldo, idx, (float*)din, ldi, B, H, W, C, stride);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// generic forms (any window <= 15 / stride / scale); Ho = (H + 2*pad - k) / stride + 1
int mdcv_maxpool_fwd(int dtype, const void* in, int ldi, void* out, int ldo, unsigned char* idx, int B, int H, int W, int C, int k, int stride,
                     int pad, void* stream) {
  if (!in || !out || !idx || (C & 7) || k < 1 || k > 15 || stride < 1 || pad < 0 || 2 * pad >= k + (k == 1) || H + 2 * pad < k || W + 2 * pad < k) return MDCV_EARG;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * Ho * Wo * C;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(maxpool_fwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, idx, B, H, W, C, k, stride, pad, Ho, Wo);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(maxpool_fwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, idx, B, H, W, C, k, stride, pad, Ho, Wo);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_maxpool_bwd(int dtype, const void* dout, int ldo, const unsigned char* idx, void* din, int ldi, int B, int H, int W, int C, int k, int stride,
                     int pad, void* stream) {
  if (!dout || !din || !idx || (C & 7) || k < 1 || k > 15 || stride < 1 || pad < 0 || 2 * pad >= k + (k == 1) || H + 2 * pad < k || W + 2 * pad < k) return MDCV_EARG;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * H * W * C;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(maxpool_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)dout, ldo, idx, (bf16_t*)din, ldi, B, H, W, C, k, stride, pad, Ho, Wo);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(maxpool_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)dout, ldo, idx, (float*)din, ldi, B, H, W, C, k, stride, pad, Ho, Wo);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_upsample_fwd(int dtype, const void* in, int ldi, void* out, int ldo, int B, int H, int W, int C, int scale, void* stream) {
  if (!in || !out || (C & 7) || scale < 1 || scale > 64) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * H * W * scale * scale * C;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(upsample_fwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, B, H, W, C, scale);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(upsample_fwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, B, H, W, C, scale);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_upsample_bwd(int dtype, const void* dout, int ldo, void* din, int ldi, int B, int H, int W, int C, int scale, void* stream) {
  if (!dout || !din || (C & 7) || scale < 1 || scale > 64) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * H * W * C;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(upsample_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)dout, ldo, (bf16_t*)din, ldi, B, H, W, C, scale);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(upsample_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)dout, ldo, (float*)din, ldi, B, H, W, C, scale);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

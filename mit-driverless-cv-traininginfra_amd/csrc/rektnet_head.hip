// RektNet keypoint head + loss as wavefront-reduction kernels for gfx950.
//
//   flat softmax over H*W + soft-argmax   <- RektNet/keypoint_net.py:46-56,68-70
//   CrossRatioLoss forward + backward     <- RektNet/cross_ratio_loss.py:20-63 (backward formulas: SURVEY appendix B)
//
// Logits come from the 1x1 head conv as NHWC (channel = keypoint, stride ldc); heat-maps leave as NCHW fp32 because
// that is what the reference API returns.  The head always computes in fp32.
#include "common.h"

namespace {

// one block per (image, keypoint): max, exp/sum, normalise, expected x/y
template <typename T>
__global__ __launch_bounds__(256) void softargmax_fwd_kernel(const T* __restrict__ logits, int ldc, int K, int H, int W,
                                                             float* __restrict__ hm, float* __restrict__ pts) {
  extern __shared__ float sh[];              // H*W exps
  __shared__ float red[8];
  const int bk = blockIdx.x, b = bk / K, k = bk - b * K;
  const int HW = H * W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* src = logits + (size_t)b * HW * ldc + k;
  float mx = -INFINITY;
  for (int j = tid; j < HW; j += 256) { const float v = ET<T>::ld(src + (size_t)j * ldc); sh[j] = v; mx = fmaxf(mx, v); }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < HW; j += 256) { const float e = expf(sh[j] - mx); sh[j] = e; sum += e; }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  const float inv = 1.f / sum;
  const float stepx = ((W - 1.f) / W) / (W > 1 ? (W - 1.f) : 1.f), stepy = ((H - 1.f) / H) / (H > 1 ? (H - 1.f) : 1.f);   // linspace(0,(n-1)/n,n)
  float ex = 0.f, ey = 0.f;
  float* dst = hm + (size_t)bk * HW;
  for (int j = tid; j < HW; j += 256) {
    const float p = sh[j] * inv;
    dst[j] = p;
    const int yy = j / W, xx = j - yy * W;
    ex += p * (xx * stepx); ey += p * (yy * stepy);
  }
  ex = wave_sum(ex); ey = wave_sum(ey);
  if (lane == 0) { red[wave] = ex; red[4 + wave] = ey; }
  __syncthreads();
  if (tid == 0) {
    pts[(size_t)bk * 2 + 0] = red[0] + red[1] + red[2] + red[3];     // (x, y) order, keypoint_net.py:56
    pts[(size_t)bk * 2 + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

// The head's 1x1 conv with an fp32 OUTPUT (round 6): logits[p][k] = bias[k] + sum_c x[p][c] * w[k][c], x = the last block's bf16 activations (NHWC),
// w = the fp32 master weights [K][C] as they sit in the parameter buffer (no packing), out = fp32 [M][KP].  Why: the flat softmax over 6400
// positions turns a logit's rounding step into a relative error of every heat-map weight -- with bf16 logits (8 mantissa bits at |logit| ~ 4..16:
// steps of 0.03..0.06) the key points of the batch-256 test moved by up to 0.065 against the fp32 oracle where the reference's own arithmetic under
// torch.autocast(bfloat16) shows 0.061; the features stay bf16, only the 8 numbers per pixel behind them are kept exact.  HBM-bound (256 B in,
// 32 B out per pixel): four lanes share a pixel, each multiplies a quarter of the channels against the weights in LDS (broadcast reads), two
// DPP-free butterfly steps add the quarters in a fixed order.
template <int KP>
__global__ __launch_bounds__(256) void head1x1_f32_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, long long M, int C, int K) {
  extern __shared__ float sw[];                                // [KP][C], rows >= K are zeros
  const int tid = threadIdx.x;
  for (int i = tid; i < KP * C; i += 256) sw[i] = (i / C) < K ? w[i] : 0.f;
  __syncthreads();
  const long long p = (long long)blockIdx.x * 64 + (tid >> 2);
  const int part = tid & 3, cq = C >> 2;                       // C % 32 == 0: whole 16-byte vectors per lane
  float acc[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) acc[k] = 0.f;
  if (p < M) {
    const bf16_t* xp = x + p * ldx + part * cq;
    const float* wp = sw + part * cq;
    for (int c = 0; c < cq; c += 8) {
      float f[8];
      ET<bf16_t>::unpack(mdcv_ld_stream(xp + c), f);          // last reader of the features on the way forward
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const float4 w0 = *reinterpret_cast<const float4*>(wp + k * C + c), w1 = *reinterpret_cast<const float4*>(wp + k * C + c + 4);
        float t = acc[k];
        t = __builtin_fmaf(f[0], w0.x, t); t = __builtin_fmaf(f[1], w0.y, t); t = __builtin_fmaf(f[2], w0.z, t); t = __builtin_fmaf(f[3], w0.w, t);
        t = __builtin_fmaf(f[4], w1.x, t); t = __builtin_fmaf(f[5], w1.y, t); t = __builtin_fmaf(f[6], w1.z, t); t = __builtin_fmaf(f[7], w1.w, t);
        acc[k] = t;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    acc[k] += __shfl_xor(acc[k], 1, 64);
    acc[k] += __shfl_xor(acc[k], 2, 64);
  }
  if (p < M && part == 0) {
    float o[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) o[k] = k < K ? acc[k] + (bias ? bias[k] : 0.f) : 0.f;
    float4* dst = reinterpret_cast<float4*>(out + p * KP);
#pragma unroll
    for (int k = 0; k < KP; k += 4) dst[k >> 2] = float4{o[k], o[k + 1], o[k + 2], o[k + 3]};
  }
}

// The same product for C = 128 (KeypointNet's head, keypoint_net.py:40) with the WEIGHTS IN REGISTERS: eight lanes share a pixel, each holds its 16 channels
// of all 8 rows (128 VGPRs) for the life of the workgroup and walks a run of pixels.  The LDS form above issues sixteen 16-byte weight reads per 8 channels
// per lane -- 85 us of LDS time for the 80^2 x 256-image tensor, 190 us in the step against 90 us of bytes (420 MB of features in, 52 MB of logits out).
template <int KP, int KR>                                  // KR: rows computed (K <= KR <= KP); the others are zeros
__global__ __launch_bounds__(256) void head1x1_f32_c128_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, long long M, int K, int ppb) {
  constexpr int C = 128, CPL = 16;
  const int tid = threadIdx.x, part = tid & 7, slot = tid >> 3;
  float wr[KR][CPL];
#pragma unroll
  for (int k = 0; k < KR; ++k)
#pragma unroll
    for (int j = 0; j < CPL; ++j) wr[k][j] = k < K ? w[k * C + part * CPL + j] : 0.f;
  float bv[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) bv[k] = (bias && k < K) ? bias[k] : 0.f;
  // Grid-stride over 32-pixel groups, PF groups requested ahead: one group in flight per wave (8 waves of 160 VGPRs per CU = 16 KiB) is the
  // latency-bound rate -- 1.5 us per group, 3.0 TB/s; four groups ahead = 64 KiB per CU.
  constexpr int PF = 4;
  const long long gs = (long long)gridDim.x * 32;
  const long long pf = (long long)blockIdx.x * 32 + slot;
  uint4 nb[PF][2];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    nb[d][0] = nb[d][1] = uint4{0, 0, 0, 0};
    const long long pp = pf + d * gs;
    if (pp < M) { const bf16_t* xp = x + pp * ldx + part * CPL; nb[d][0] = mdcv_ld_stream(xp); nb[d][1] = mdcv_ld_stream(xp + 8); }   // (last reader of the features on the way forward)
  }
  for (long long pb = pf; pb < M; pb += PF * gs) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const long long p = pb + d * gs;
      if (p < M) {
        float f[CPL];
        const uint4 q0 = nb[d][0], q1 = nb[d][1];
        const long long pn = p + PF * gs;
        if (pn < M) { const bf16_t* xn = x + pn * ldx + part * CPL; nb[d][0] = mdcv_ld_stream(xn); nb[d][1] = mdcv_ld_stream(xn + 8); }
        ET<bf16_t>::unpack(q0, f);
        ET<bf16_t>::unpack(q1, f + 8);
        float acc[KP];
#pragma unroll
        for (int k = KR; k < KP; ++k) acc[k] = 0.f;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < CPL; ++j) t = __builtin_fmaf(f[j], wr[k][j], t);
          t += __shfl_xor(t, 1, 64);
          t += __shfl_xor(t, 2, 64);
          t += __shfl_xor(t, 4, 64);
          acc[k] = t + bv[k];
        }
        if (part == 0) {
          float4* dst = reinterpret_cast<float4*>(out + p * KP);
#pragma unroll
          for (int k = 0; k < KP; k += 4) dst[k >> 2] = float4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
        }
      }
    }
  }
}

// s[b,k] = sum_j p_j * dhm_j   (softmax Jacobian term, only needed when the heat-map itself carries a gradient)
__global__ __launch_bounds__(256) void softmax_dot_kernel(const float* __restrict__ hm, const float* __restrict__ dhm, int HW, float* __restrict__ s) {
  __shared__ float red[4];
  const size_t off = (size_t)blockIdx.x * HW;
  float acc = 0.f;
  for (int j = threadIdx.x; j < HW; j += 256) acc += hm[off + j] * dhm[off + j];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) s[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// dz[b,j,k] = p * ( gx*(vx_j - xhat) + gy*(vy_j - yhat) + dhm_j - s )   one thread per pixel, all K keypoints -> one NHWC vector store
template <typename T, int CP>
__global__ __launch_bounds__(256) void softargmax_bwd_kernel(const float* __restrict__ hm, const float* __restrict__ pts, const float* __restrict__ dpts,
                                                             const float* __restrict__ dhm, const float* __restrict__ sdot, int B, int K, int H, int W,
                                                             T* __restrict__ dlogits, int ldd) {
  const int HW = H * W;
  const long long total = (long long)B * HW;
  const float stepx = ((W - 1.f) / W) / (W > 1 ? (W - 1.f) : 1.f), stepy = ((H - 1.f) / H) / (H > 1 ? (H - 1.f) : 1.f);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), j = (int)(i - (long long)b * HW);
    const int yy = j / W, xx = j - yy * W;
    const float vx = xx * stepx, vy = yy * stepy;
    float o[CP];
#pragma unroll
    for (int k = 0; k < CP; ++k) {
      float g = 0.f;
      if (k < K) {
        const size_t bk = (size_t)b * K + k;
        const float p = hm[bk * HW + j];
        float t = 0.f;
        if (dpts) t += dpts[bk * 2] * (vx - pts[bk * 2]) + dpts[bk * 2 + 1] * (vy - pts[bk * 2 + 1]);
        if (dhm) t += dhm[bk * HW + j] - sdot[bk];
        g = p * t;
      }
      o[k] = g;
    }
    T* dst = dlogits + (size_t)i * ldd;
#pragma unroll
    for (int v = 0; v < CP / ET<T>::VEC; ++v) *reinterpret_cast<uint4*>(dst + v * ET<T>::VEC) = ET<T>::pack(o + v * ET<T>::VEC);
  }
}

// sum (hm - thm)^2 -> acc (fp64) ; dhm = 2 (hm - thm) / B * up
__global__ __launch_bounds__(256) void hm_l2_kernel(const float* __restrict__ hm, const float* __restrict__ thm, long long n, float scale,
                                                    const float* __restrict__ gscale, float* __restrict__ dhm, double* __restrict__ acc) {
  __shared__ double red[4];
  const float up = gscale ? gscale[0] : 1.f;
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = hm[i] - thm[i];
    s += (double)(d * d);
    if (dhm) dhm[i] = 2.f * d * scale * up;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}

// difference vectors d_q = P[a]-P[b] used by the six geometric terms (cross_ratio_loss.py:36-55)
__constant__ int DQ_A[9] = {5, 3, 1, 6, 4, 2, 2, 4, 6};
__constant__ int DQ_B[9] = {3, 1, 0, 4, 2, 0, 1, 3, 5};
//                 q:        0    1    2    3    4    5    6    7    8
//                          d53  d31  d10  d64  d42  d20  d21  d43  d65
__constant__ int TERM_U[6] = {1, 2, 3, 4, 7, 8};     // vA: d31.d53  vB: d10.d31  vC: d64.d42  vD: d42.d20  hA: d43.d21  hB: d65.d43
__constant__ int TERM_V[6] = {0, 1, 4, 5, 6, 7};

// single block: location loss on points + geometric loss, forward value and d/dpts
//   loss_type: 0 l2_softargmax, 1 l2_heatmap (location part supplied in hm_acc), 2 l1_softargmax
__global__ __launch_bounds__(256) void cross_ratio_kernel(const float* __restrict__ pts, const float* __restrict__ tpts, int B, int loss_type,
                                                          int include_geo, float gamma_h, float gamma_v, double* __restrict__ hm_acc,
                                                          const float* __restrict__ gscale, float* __restrict__ out3, float* __restrict__ dpts) {
  __shared__ double red[4][19];
  __shared__ double tot[19];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float up = gscale ? gscale[0] : 1.f, up_geo = gscale ? gscale[1] : 1.f;   // upstream grads of (location, geo) parts
  double part[19];                                   // 0: location sum ; 1..18: sums of the 9 unit vectors (x,y)
#pragma unroll
  for (int k = 0; k < 19; ++k) part[k] = 0.0;
  for (int i = tid; i < B; i += 256) {
    const float* P = pts + (size_t)i * 14;
    const float* Tg = tpts + (size_t)i * 14;
    if (loss_type != 1) {
      float s = 0.f;
      for (int c = 0; c < 14; ++c) { const float d = P[c] - Tg[c]; s += loss_type == 0 ? d * d : fabsf(d); }
      part[0] += (double)s;
    }
    if (include_geo) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float dx = P[2 * DQ_A[q]] - P[2 * DQ_B[q]], dy = P[2 * DQ_A[q] + 1] - P[2 * DQ_B[q] + 1];
        const float nrm = fmaxf(sqrtf(dx * dx + dy * dy), 1e-12f);       // F.normalize eps
        part[1 + 2 * q] += (double)(dx / nrm); part[2 + 2 * q] += (double)(dy / nrm);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 19; ++k) { const double s = wave_sum_d(part[k]); if (lane == 0) red[wave][k] = s; }
  __syncthreads();
  if (tid < 19) tot[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  __syncthreads();
  const double invB = 1.0 / (double)B, invB2 = invB * invB;
  const double wt[6] = {gamma_v / 4.0, gamma_v / 4.0, gamma_v / 4.0, gamma_v / 4.0, gamma_h / 2.0, gamma_h / 2.0};
  if (tid == 0) {
    double loc = loss_type == 1 ? hm_acc[0] * invB : tot[0] * invB;
    if (loss_type == 1) hm_acc[0] = 0.0;
    double geo = 0.0;
    if (include_geo)
      for (int t = 0; t < 6; ++t) {
        const int u = TERM_U[t], v = TERM_V[t];
        const double dot = tot[1 + 2 * u] * tot[1 + 2 * v] + tot[2 + 2 * u] * tot[2 + 2 * v];
        geo += wt[t] * (1.0 - dot * invB2);            // mean over the [B,B] all-pairs matrix of 1 - u_i.v_j
      }
    out3[0] = (float)loc; out3[1] = (float)geo; out3[2] = (float)(loc + geo);
  }
  if (!dpts) return;
  for (int i = tid; i < B; i += 256) {
    const float* P = pts + (size_t)i * 14;
    const float* Tg = tpts + (size_t)i * 14;
    float g[14];
    for (int c = 0; c < 14; ++c) {
      const float d = P[c] - Tg[c];
      g[c] = up * (loss_type == 0 ? 2.f * d * (float)invB : (loss_type == 2 ? ((d > 0.f) - (d < 0.f)) * (float)invB : 0.f));
    }
    if (include_geo) {
      float G[9][2];
      for (int q = 0; q < 9; ++q) { G[q][0] = 0.f; G[q][1] = 0.f; }
      for (int t = 0; t < 6; ++t) {
        const int u = TERM_U[t], v = TERM_V[t];
        const float w = (float)(wt[t] * invB2);
        G[u][0] -= w * (float)tot[1 + 2 * v]; G[u][1] -= w * (float)tot[2 + 2 * v];
        G[v][0] -= w * (float)tot[1 + 2 * u]; G[v][1] -= w * (float)tot[2 + 2 * u];
      }
      for (int q = 0; q < 9; ++q) {
        const int pa = DQ_A[q], pb = DQ_B[q];
        const float dx = P[2 * pa] - P[2 * pb], dy = P[2 * pa + 1] - P[2 * pb + 1];
        const float nrm = fmaxf(sqrtf(dx * dx + dy * dy), 1e-12f);
        const float ux = dx / nrm, uy = dy / nrm;
        const float ug = ux * G[q][0] + uy * G[q][1];
        const float ddx = (G[q][0] - ux * ug) / nrm, ddy = (G[q][1] - uy * ug) / nrm;
        g[2 * pa] += up_geo * ddx; g[2 * pa + 1] += up_geo * ddy; g[2 * pb] -= up_geo * ddx; g[2 * pb + 1] -= up_geo * ddy;
      }
    }
    for (int c = 0; c < 14; ++c) dpts[(size_t)i * 14 + c] = g[c];
  }
}

}  // namespace

extern "C" {

int mdcv_softargmax_fwd(int dtype, const void* logits, int ldc, int B, int K, int H, int W, float* hm, float* pts, void* stream) {
  if (!logits || !hm || !pts || H * W * 4 > 64 * 1024) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const unsigned lds = (unsigned)(H * W * 4);
  if (dtype == MDCV_BF16) MDCV_LAUNCH(softargmax_fwd_kernel<bf16_t>, dim3((unsigned)(B * K)), dim3(256), lds, st, (const bf16_t*)logits, ldc, K, H, W, hm, pts);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(softargmax_fwd_kernel<float>, dim3((unsigned)(B * K)), dim3(256), lds, st, (const float*)logits, ldc, K, H, W, hm, pts);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_head1x1_f32(const void* x_bf16, int ldx, const float* w, const float* bias, float* out, long long M, int C, int K, void* stream) {
  if (!x_bf16 || !w || !out || M < 0 || K < 1 || K > 8 || C < 32 || (C & 31) || (ldx & 7) || ldx < C || C > 1024) return MDCV_EARG;
  if (M == 0) return MDCV_OK;
  if (C == 128) {                                            // weights in registers, 2048 pixels per workgroup (>= 4 workgroups per CU on KeypointNet's tensors)
    const int ppb = M >= 2048LL * 1024 ? 2048 : (M >= 256LL * 1024 ? 1024 : 256);
    if (K <= 7) MDCV_LAUNCH((head1x1_f32_c128_kernel<8, 7>), dim3((unsigned)((M + ppb - 1) / ppb)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, ldx, w, bias, out, M, K, ppb);
    else MDCV_LAUNCH((head1x1_f32_c128_kernel<8, 8>), dim3((unsigned)((M + ppb - 1) / ppb)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, ldx, w, bias, out, M, K, ppb);
    MDCV_CHECK_LAUNCH();
    return MDCV_OK;
  }
  MDCV_LAUNCH(head1x1_f32_kernel<8>, dim3((unsigned)((M + 63) / 64)), dim3(256), (unsigned)(8 * C * 4), (hipStream_t)stream, (const bf16_t*)x_bf16, ldx, w, bias,
              out, M, C, K);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// dlogits (NHWC, 8 channels: K real + zero pad) from d pts and/or d heat-map ; sdot_ws: B*K floats of scratch
int mdcv_softargmax_bwd(int dtype, const float* hm, const float* pts, const float* dpts, const float* dhm, float* sdot_ws, int B, int K,
                        int H, int W, void* dlogits, int ldd, void* stream) {
  if (!hm || !pts || !dlogits || K > 8 || (ldd & 7)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (dhm) {
    if (!sdot_ws) return MDCV_EARG;
    MDCV_LAUNCH(softmax_dot_kernel, dim3((unsigned)(B * K)), dim3(256), 0, st, hm, dhm, H * W, sdot_ws);
    MDCV_CHECK_LAUNCH();
  }
  const unsigned grid = (unsigned)cdiv((long long)B * H * W, 256);
  if (dtype == MDCV_BF16) MDCV_LAUNCH((softargmax_bwd_kernel<bf16_t, 8>), dim3(grid), dim3(256), 0, st, hm, pts, dpts, dhm, sdot_ws, B, K, H, W, (bf16_t*)dlogits, ldd);
  else if (dtype == MDCV_F32) MDCV_LAUNCH((softargmax_bwd_kernel<float, 8>), dim3(grid), dim3(256), 0, st, hm, pts, dpts, dhm, sdot_ws, B, K, H, W, (float*)dlogits, ldd);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// CrossRatioLoss forward + gradients in one call.  loss_type: 0 l2_softargmax, 1 l2_heatmap, 2 l1_softargmax.
// out3 = (location, geo, total) ; dpts [B,7,2] ; dhm [B,7,H,W] (l2_heatmap only) ; acc_ws: one zero-initialised double (l2_heatmap only,
// may be NULL otherwise).
// gscale (device, 2 floats, may be NULL = ones): upstream gradients of the location part and of the geo part.
int mdcv_cross_ratio_loss(const float* hm, const float* pts, const float* thm, const float* tpts, int B, int H, int W, int loss_type,
                          int include_geo, float gamma_horz, float gamma_vert, double* acc_ws, const float* gscale, float* out3,
                          float* dpts, float* dhm, void* stream) {
  if (!pts || !tpts || !out3 || loss_type < 0 || loss_type > 2 || (loss_type == 1 && !acc_ws)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (loss_type == 1) {
    if (!hm || !thm) return MDCV_EARG;
    const long long n = (long long)B * 7 * H * W;
    MDCV_LAUNCH(hm_l2_kernel, dim3((unsigned)min(cdiv(n, 1024), 2048)), dim3(256), 0, st, hm, thm, n, 1.f / (float)B, gscale, dhm, acc_ws);
    MDCV_CHECK_LAUNCH();
  }
  MDCV_LAUNCH(cross_ratio_kernel, dim3(1), dim3(256), 0, st, pts, tpts, B, loss_type, include_geo, gamma_horz, gamma_vert, acc_ws, gscale, out3, dpts);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

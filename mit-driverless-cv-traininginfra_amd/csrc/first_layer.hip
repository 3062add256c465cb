// The first layer's weight gradient without its BatchNorm-apply pass (gfx950).
//
// Reference path: conv -> BatchNorm2d -> LeakyReLU of the network's first block (CVC-YOLOv3/models.py:57-71, layer 0 of yolov3_80class.cfg);
// autograd forms dy = cA*g + cB*y + cC (g = dz * act'(BN(y)), the per-channel coefficients of the BatchNorm backward) over the whole 416^2 x 32
// tensor, then correlates it with the input patches.  The layer has no data gradient, so dy exists for the weight gradient alone, and the
// correlation is linear in dy:
//
//     dW[co][ci][t] = cA[co] * G[co][ci][t] + cB[co] * Y[co][ci][t] + cC[co] * X1[t][ci]
//     G = sum_p g[p][co] * xcol[p][t][ci]      (weight-gradient kernel on g: the stride-2 data gradient above stores g instead of dz,
//                                               mdcv_conv2d_dgrad_bnsums_masked)
//     Y = sum_p y[p][co] * xcol[p][t][ci]      (weight-gradient kernel on the forward output: depends on forward data only, runs on the side
//                                               stream under the MFMA-bound middle of the FORWARD pass)
//     X1 = sum_p xcol[p][t][ci]                (tap sums of the input, below)
//
// so the 1.06 GB apply pass (read dz, y; write dy) at the HBM-bound tail of the backward disappears and the last weight gradient no longer
// waits for it.  Everything here is deterministic (no atomics).
#include "common.h"

namespace {

constexpr int TAP_K_MAX = 7;      // kernel rows / columns
constexpr int TAP_BAND = 32;      // input rows per workgroup (85 us at 32 x 416^2 inside the step; 8-row bands: 102 us)

__device__ __forceinline__ bool tap_hits(int i, int k, int stride, int pad, int dil, int nout) {   // input index i is read through tap k by some output
  const int n = i + pad - k * dil;
  return n >= 0 && n % stride == 0 && n / stride < nout;
}

// Workgroup (image b, band of TAP_BAND input rows, chunk of 256 input columns): a thread owns ONE column and walks down the band with one
// accumulator vector per kernel row (which rows count for kh is uniform per input row); the kernel columns its column counts for are a
// per-thread constant applied once, in the workgroup reduction at the end.  Coalesced 16-byte loads, one reduction per band.
__global__ __launch_bounds__(256) void tap_bandsums_kernel(const bf16_t* __restrict__ x, int ldc, int H, int W, int KH, int KW, int stride,
                                                           int pad, int dil, int Hout, int Wout, int nbands, float* __restrict__ part) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / nbands, r0 = (blockIdx.x - b * nbands) * TAP_BAND, iw = blockIdx.y * 256 + tid;
  const int r1 = min(r0 + TAP_BAND, H);
  float acc[TAP_K_MAX][8];
#pragma unroll
  for (int k = 0; k < TAP_K_MAX; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
  if (iw < W) {
    const bf16_t* __restrict__ xc = x + ((long long)b * H * W + iw) * ldc;
#pragma unroll 4
    for (int ih = r0; ih < r1; ++ih) {
      float v[8];
      ET<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xc + (long long)ih * W * ldc), v);
#pragma unroll
      for (int k = 0; k < TAP_K_MAX; ++k) {
        const bool ok = k < KH && tap_hits(ih, k, stride, pad, dil, Hout);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] += ok ? v[e] : 0.f;
      }
    }
  }
  __shared__ float red[4][TAP_K_MAX * TAP_K_MAX * 8];
#pragma unroll
  for (int kh = 0; kh < TAP_K_MAX; ++kh) {
    if (kh >= KH) break;
    for (int kw = 0; kw < KW; ++kw) {
      const bool ok = iw < W && tap_hits(iw, kw, stride, pad, dil, Wout);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = ok ? acc[kh][e] : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave][(kh * KW + kw) * 8 + e] = s;
      }
    }
  }
  __syncthreads();
  const int n = KH * KW * 8;
  float* __restrict__ dst = part + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * n;
  for (int t = tid; t < n; t += 256) dst[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// one workgroup per tap: its 8 channel sums over all workgroup rows of `part`, in a fixed order (32 slices, then the slices)
__global__ __launch_bounds__(256) void tap_sums_final_kernel(const float* __restrict__ part, int nparts, int ntaps, float* __restrict__ out) {
  const int t = blockIdx.x, c = threadIdx.x & 7, slice = threadIdx.x >> 3;
  double s = 0.0;
#pragma unroll 4
  for (int p = slice; p < nparts; p += 32) s += (double)part[((long long)p * ntaps + t) * 8 + c];
  __shared__ double red[32][8];
  red[slice][c] = s;
  __syncthreads();
  if (slice == 0) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) tot += red[k][c];
    out[t * 8 + c] = (float)tot;
  }
}

__global__ __launch_bounds__(256) void first_layer_combine_kernel(const float* __restrict__ G, const float* __restrict__ Y,
                                                                  const float* __restrict__ X1, const float* __restrict__ cA,
                                                                  const float* __restrict__ cB, const float* __restrict__ cC,
                                                                  float* __restrict__ dw, int Cout, int Cin, int KK) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cout * Cin * KK) return;
  const int co = idx / (Cin * KK), r = idx - co * (Cin * KK), ci = r / KK, t = r - ci * KK;
  dw[idx] = cA[co] * G[idx] + (cB[co] * Y[idx] + cC[co] * X1[t * 8 + ci]);
}

}  // namespace

extern "C" {

// floats of the scratch mdcv_conv_tap_sums needs: one vector of KH*KW*8 sums per workgroup of the first pass
static int tap_bands(int H) { return (H + TAP_BAND - 1) / TAP_BAND; }
long long mdcv_conv_tap_sums_ws_floats(int B, int H, int W, int KH, int KW) {
  return (long long)B * tap_bands(H) * ((W + 255) / 256) * KH * KW * 8;
}

// X1[kh*KW + kw][c] (c < 8) = sum over images and output positions (oh, ow) of x[b][oh*stride - pad + kh*dil][ow*stride - pad + kw*dil][c]
// (zero outside the image): the column sums of the layer's im2col matrix.  x: bf16 NHWC with 8 (padded) channels.
int mdcv_conv_tap_sums(int dtype, const void* x, int ldc, int B, int H, int W, int Hout, int Wout, int KH, int KW, int stride, int pad, int dil,
                       float* ws, float* out, void* stream) {
  if (dtype != MDCV_BF16 || !x || !ws || !out || (ldc & 7) || ldc < 8 || KW < 1 || KW > TAP_K_MAX || KH < 1 || KH > TAP_K_MAX || stride < 1 ||
      dil < 1 || B < 1 || H < 1 || W < 1 || (long long)B * H * W * ldc >= (1LL << 40))
    return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int nb = tap_bands(H), gy = (W + 255) / 256;
  MDCV_LAUNCH(tap_bandsums_kernel, dim3((unsigned)(B * nb), (unsigned)gy), dim3(256), 0, st, (const bf16_t*)x, ldc, H, W, KH, KW, stride, pad, dil,
              Hout, Wout, nb, ws);
  MDCV_CHECK_LAUNCH();
  MDCV_LAUNCH(tap_sums_final_kernel, dim3((unsigned)(KH * KW)), dim3(256), 0, st, (const float*)ws, B * nb * gy, KH * KW, out);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// dw[co][ci][t] = cA[co] * G[co][ci][t] + cB[co] * Y[co][ci][t] + cC[co] * X1[t][ci]     (G, Y, dw: OIHW fp32 of the real channel counts)
int mdcv_first_layer_wgrad_combine(const float* G, const float* Y, const float* X1, const float* cA, const float* cB, const float* cC,
                                   float* dw, int Cout, int Cin, int KK, void* stream) {
  if (!G || !Y || !X1 || !cA || !cB || !cC || !dw || Cout < 1 || Cin < 1 || Cin > 8 || KK < 1) return MDCV_EARG;
  MDCV_LAUNCH(first_layer_combine_kernel, dim3((unsigned)((Cout * Cin * KK + 255) / 256)), dim3(256), 0, (hipStream_t)stream, G, Y, X1, cA, cB,
              cC, dw, Cout, Cin, KK);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

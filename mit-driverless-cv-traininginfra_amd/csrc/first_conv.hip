// The network's FIRST conv -> BatchNorm(batch statistics) -> activation (CVC-YOLOv3/models.py:57-71 at index 0: 3 -> 32 channels, 3x3 / stride 1 / pad 1,
// 416^2 x 32 images) as two streaming launches that never re-read the layer's output.
//
// Why: the layer is 25 GFLOP over a 354 MB output -- pure HBM traffic.  The generic path writes y (conv launch, 173-189 us = 2.5 TB/s: 21 632 tiles of
// 256 x 32 with K = 72), sums the statistics, then reads y again and writes z (apply pass, 135-150 us): 1 150 MB.  But the layer's INPUT is 88 MB and
// its arithmetic is free, so the conv is simply computed twice:
//   pass 1 (MODE 0): x -> per-channel sum / sum of squares of the fp32 accumulators (one partial row per workgroup), nothing else written;
//   pass 2 (MODE 1): x -> y (bf16, kept for the backward) AND z = act(scale * y + shift), formed from the y it stores, in the same store loop.
// 884 MB instead of 1 150.  A workgroup owns TR = 4 output rows of one image over the full width (one row per wave); the six input rows (16 bytes per
// pixel: 8 padded channels) sit in LDS once, zero padding included, and every tap reads them at a displaced address.  MFMA: v_mfma_f32_16x16x32_bf16 with
// A = weights [16 co x 32 k] (registers for the life of the workgroup), B = pixels [32 k x 16 px], k = (four taps, 8 channels): a lane's B fragment is ONE
// whole input pixel (ds_read_b128), 3 K steps (taps 9 .. 11 have zero weights), 6 MFMAs per 16 pixels.  D[co][px] leaves four consecutive channels of one
// pixel per lane; a wave stages its 16 px x 32 co tile (1 KiB per output) in LDS of its own and stores whole 64-byte pixel rows, 16 bytes per lane.
// Alone at 416^2 x 32 (scripts/first_conv_ab.py): pass 1 57 us, pass 2 180 us (4.4 TB/s) against 173 + 135..145 us.  What was measured on the way: the same
// tile with v_mfma_f32_16x16x16_bf16 (5 K steps of two taps) 70 / 183 us; TR = 2 / 8 rows 67 / 69 us (pass 1) and 193 us (pass 2, TR = 8); no LDS tile
// at all -- B fragments loaded straight from global memory, four groups in flight per wave -- 57 / 212 us (every pixel then crosses L2 -> L1 nine times).
#include "common.h"

namespace {

constexpr int NW = 4;                      // waves per workgroup
constexpr int TR = 4;                      // output rows per workgroup (wave w: rows w, w + NW, ...)
constexpr int COUT = 32, CIN = 8, PXB = CIN * 2;   // 16 bytes per input pixel

struct FirstConvArgs {
  const bf16_t* x; const bf16_t* wf;       // x: [B][H][W][ldx] ; wf: [32][3][3][8] (the forward operand layout of mdcv_pack_weights)
  bf16_t* y; bf16_t* z; const float* scale; const float* shift;
  float* partial;                          // MODE 0: [B * strips][2][32]
  int ldx, ldy, ldz, B, H, W, strips, act;
  float slope;
};

template <int MODE>
__global__ __launch_bounds__(NW * 64) void first_conv_kernel(FirstConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kg = lane >> 4;
  const int img = blockIdx.x / a.strips, strip = blockIdx.x - img * a.strips;
  const int r0 = strip * TR;
  const int WP = a.W + 2;                                  // LDS row: input columns -1 .. W
  unsigned char* const stage = smem + (size_t)(TR + 2) * (WP + 16) * PXB;   // per wave: 2 x 1 KiB (y, z); (+16 columns: the last group's masked pixels)
  // ---- the TR + 2 input rows, zero padding included; eight 16-byte loads per thread in flight before the first store
  {
    const bf16_t* xi = a.x + (size_t)img * a.H * a.W * a.ldx;
    const int total = (TR + 2) * WP;
    for (int base = tid; base < total; base += 8 * NW * 64) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NW * 64;
        int rr = 0;                                          // i / WP without a division: TR + 2 rows
#pragma unroll
        for (int k = 1; k < TR + 2; ++k) rr += i >= k * WP;
        const int cc = i - rr * WP;
        const int gy = r0 - 1 + rr, gx = cc - 1;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (i < total && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v[u] = *reinterpret_cast<const uint4*>(xi + ((size_t)gy * a.W + gx) * a.ldx);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NW * 64;
        if (i < total) *reinterpret_cast<uint4*>(smem + (size_t)i * PXB) = v[u];
      }
    }
  }
  // ---- weights: A fragments of v_mfma_f32_16x16x32_bf16, lane (co = h * 16 + lane % 16, kg): tap 4 s + kg, all 8 channels; taps 9 .. 11 are zero
  bf16x8_t wa[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = 4 * s + kg;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (tap < 9) v = *reinterpret_cast<const uint4*>(a.wf + ((size_t)(h * 16 + l16) * 9 + tap) * CIN);
      wa[h][s] = __builtin_bit_cast(bf16x8_t, v);
    }
  // B fragment of step s: the whole 16-byte pixel (row + dy, col + dx) of tap 4 s + kg; taps past 8 re-read tap 8 (zero weights)
  int boff[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    int tap = 4 * s + kg;
    tap = tap < 9 ? tap : 8;
    const int dy = tap / 3, dx = tap - dy * 3;
    boff[s] = (dy * WP + l16 + dx) * PXB;
  }
  float sc[2][4], sh[2][4];
  if constexpr (MODE == 1) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc[h][r] = a.scale[h * 16 + kg * 4 + r]; sh[h][r] = a.shift[h * 16 + kg * 4 + r]; }
  }
  float ssum[2][4], ssq[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[h][r] = 0.f; ssq[h][r] = 0.f; }
  __syncthreads();

  unsigned char* const st = stage + wave * 2048;
  for (int lr = wave; lr < TR; lr += NW) {
    const int row = r0 + lr;                                // (wave-uniform)
    if (row >= a.H) break;
    const size_t rowpix = ((size_t)img * a.H + row) * a.W;
    const unsigned char* const rb = smem + (size_t)lr * WP * PXB;
    // the fragments of a group are requested one group AHEAD of the MFMAs that use them
    // (columns past W read the right padding column and beyond -- inside the tile, the values belong to pixels that are masked below)
    uint4 bn[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) bn[s] = *reinterpret_cast<const uint4*>(rb + boff[s]);
    for (int c0 = 0; c0 < a.W; c0 += 16) {
      uint4 b[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) b[s] = bn[s];
      if (c0 + 16 < a.W) {
#pragma unroll
        for (int s = 0; s < 3; ++s) bn[s] = *reinterpret_cast<const uint4*>(rb + boff[s] + (c0 + 16) * PXB);
      }
      f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, b[s]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][s], bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][s], bb, acc[1], 0, 0, 0);
      }
      // D[co = h * 16 + 4 kg + r][pixel = c0 + l16]
      if constexpr (MODE == 0) {
        if (c0 + l16 < a.W) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float v = acc[h][r]; ssum[h][r] += v; ssq[h][r] += v * v; }
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint2 yq, zq;
          yq.x = pack_bf16x2(acc[h][0], acc[h][1]); yq.y = pack_bf16x2(acc[h][2], acc[h][3]);
          float v[4];                                       // the apply sees y as stored
          v[0] = __uint_as_float(yq.x << 16); v[1] = __uint_as_float(yq.x & 0xffff0000u);
          v[2] = __uint_as_float(yq.y << 16); v[3] = __uint_as_float(yq.y & 0xffff0000u);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pre = __builtin_fmaf(v[r], sc[h][r], sh[h][r]);
            v[r] = a.act == 0 ? pre : (pre > 0.f ? pre : pre * a.slope);
          }
          zq.x = pack_bf16x2(v[0], v[1]); zq.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(st + l16 * 64 + (h * 16 + kg * 4) * 2) = yq;
          *reinterpret_cast<uint2*>(st + 1024 + l16 * 64 + (h * 16 + kg * 4) * 2) = zq;
        }
        // one wave, LDS of its own: the hardware serves its LDS operations in order, so no s_barrier -- but the lanes exchange data, which the
        // compiler's per-thread view cannot see: the waits below are also its fences (memory clobber)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint4 yo = *reinterpret_cast<const uint4*>(st + lane * 16);   // -> whole 64-byte pixel rows, 16 bytes per lane
        const uint4 zo = *reinterpret_cast<const uint4*>(st + 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (the next group's writes stay behind these reads)
        const int px = c0 + (lane >> 2);
        if (px < a.W) {
          *reinterpret_cast<uint4*>(a.y + (rowpix + px) * a.ldy + (lane & 3) * 8) = yo;
          *reinterpret_cast<uint4*>(a.z + (rowpix + px) * a.ldz + (lane & 3) * 8) = zo;
        }
      }
    }
  }
  if constexpr (MODE == 0) {
    // pixels of one channel sit one lane apart inside a 16-lane row: fold the row, then the four waves meet in LDS (fixed order)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { ssum[h][r] += __shfl_xor(ssum[h][r], o, 64); ssq[h][r] += __shfl_xor(ssq[h][r], o, 64); }
    float* red = reinterpret_cast<float*>(stage);           // [NW][2][32] (nobody reads the input tile through it)
    if (l16 == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[(wave * 2 + 0) * COUT + h * 16 + kg * 4 + r] = ssum[h][r];
          red[(wave * 2 + 1) * COUT + h * 16 + kg * 4 + r] = ssq[h][r];
        }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * 2 * COUT + tid];
      a.partial[(size_t)blockIdx.x * 2 * COUT + tid] = t;
    }
  }
}

static size_t lds_bytes(int W) { return (size_t)(TR + 2) * (W + 2 + 16) * PXB + NW * 2048; }
static int nstrips(int H) { return (H + TR - 1) / TR; }

}  // namespace

extern "C" {

// 1 where the geometry takes the two-pass streaming form: bf16, 8 padded input channels at stride 8, 32 output channels, 3x3 / stride 1 / pad 1
int mdcv_first_conv_ok(int dtype, int B, int H, int W, int Cin_pad, int Cout_pad, int KH, int KW, int stride, int pad, int dil, int ldx) {
  return (dtype & 0xff) == MDCV_BF16 && Cin_pad == CIN && ldx == CIN && Cout_pad == COUT && KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 &&
         B >= 1 && H >= 1 && W >= 1 && W <= 1024 && (long long)B * H * W * COUT * 2 < (1LL << 40);
}
// partial rows pass 1 writes ([rows][2][32]: sum, sum of squares of the fp32 accumulators; finish with mdcv_bn_stats_finalize)
int mdcv_first_conv_rows(int B, int H) { return B * nstrips(H); }

int mdcv_first_conv_stats(int dtype, const void* x, int ldx, const void* w_packed, float* partial, int B, int H, int W, void* stream) {
  if (!x || !w_packed || !partial || !mdcv_first_conv_ok(dtype, B, H, W, CIN, COUT, 3, 3, 1, 1, 1, ldx)) return MDCV_EARG;
  FirstConvArgs a{};
  a.x = reinterpret_cast<const bf16_t*>(x); a.wf = reinterpret_cast<const bf16_t*>(w_packed); a.partial = partial;
  a.ldx = ldx; a.B = B; a.H = H; a.W = W; a.strips = nstrips(H);
  const int lds = (int)lds_bytes(W);
  static DynLds dyn;
  if (hipError_t e = mdcv_dyn_lds(dyn, reinterpret_cast<const void*>(first_conv_kernel<0>), lds); e != hipSuccess) return (int)e;
  MDCV_LAUNCH(first_conv_kernel<0>, dim3((unsigned)(B * a.strips)), dim3(NW * 64), lds, (hipStream_t)stream, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_first_conv_bn_act(int dtype, const void* x, int ldx, const void* w_packed, const float* scale, const float* shift, int act, float slope,
                           void* y, int ldy, void* z, int ldz, int B, int H, int W, void* stream) {
  if (!x || !w_packed || !scale || !shift || !y || !z || (ldy & 7) || (ldz & 7) || ldy < COUT || ldz < COUT ||
      !mdcv_first_conv_ok(dtype, B, H, W, CIN, COUT, 3, 3, 1, 1, 1, ldx))
    return MDCV_EARG;
  FirstConvArgs a{};
  a.x = reinterpret_cast<const bf16_t*>(x); a.wf = reinterpret_cast<const bf16_t*>(w_packed);
  a.y = reinterpret_cast<bf16_t*>(y); a.z = reinterpret_cast<bf16_t*>(z); a.scale = scale; a.shift = shift;
  a.ldx = ldx; a.ldy = ldy; a.ldz = ldz; a.B = B; a.H = H; a.W = W; a.strips = nstrips(H); a.act = act; a.slope = act == 2 ? 0.f : slope;
  const int lds = (int)lds_bytes(W);
  static DynLds dyn;
  if (hipError_t e = mdcv_dyn_lds(dyn, reinterpret_cast<const void*>(first_conv_kernel<1>), lds); e != hipSuccess) return (int)e;
  MDCV_LAUNCH(first_conv_kernel<1>, dim3((unsigned)(B * a.strips)), dim3(NW * 64), lds, (hipStream_t)stream, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

// On-device synthetic cone data with the output contracts of the reference's two datasets:
//   ImageLabelDataset.__getitem__ (CVC-YOLOv3/utils/datasets.py:124-315): image [3,H,W] in [0,1], labels [T,5] zero-padded
//   ConeDataset.__getitem__      (RektNet/dataset.py:34-56, RektNet/utils.py:83-111): image [3,S,S], 7 heat-maps, 7 key points
// Every value is a pure function of (seed, step, index) through a counter-based integer hash, so the numpy oracle
// (oracle/synth_oracle.py) reproduces the tensors bit for bit; built with -ffp-contract=off for that reason.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float urand(unsigned seed, unsigned stream, unsigned long long idx) {
  unsigned h = hash32(seed * 0x9E3779B1u + stream * 0x85EBCA77u + (unsigned)idx * 0xC2B2AE3Du);
  h = hash32(h + 0x27D4EB2Fu);
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
__constant__ float CONE_RGB[3][3] = {{1.0f, 0.55f, 0.10f}, {0.15f, 0.35f, 0.95f}, {0.95f, 0.85f, 0.15f}};
__constant__ float KPX[7] = {0.5f, 0.36f, 0.64f, 0.25f, 0.75f, 0.12f, 0.88f};
__constant__ float KPY[7] = {0.04f, 0.36f, 0.36f, 0.66f, 0.66f, 0.96f, 0.96f};

// ---- detector data: one thread per (image, target slot)
__global__ void synth_targets_kernel(unsigned seed, int step, int B, int T, int num_classes, float* __restrict__ tg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, t = i - b * T;
  const unsigned long long key = (unsigned long long)step * 4099ull + (unsigned long long)b;
  int n = 1 + (int)(urand(seed, 11, key) * (float)T);
  n = n < T ? n : T;
  float* o = tg + (size_t)i * 5;
  if (t >= n) { o[0] = o[1] = o[2] = o[3] = o[4] = 0.f; return; }
  const unsigned long long k = key * 64ull + (unsigned long long)t;
  const float w = 0.03f + urand(seed, 12, k) * 0.12f;
  const float h = w * (1.3f + urand(seed, 13, k) * 0.9f);
  const float cx = w * 0.5f + urand(seed, 14, k) * (1.0f - w);
  const float cy = h * 0.5f + urand(seed, 15, k) * (1.0f - h);
  o[0] = (float)(int)(urand(seed, 16, k) * (float)num_classes);
  o[1] = cx; o[2] = cy; o[3] = w; o[4] = h;
}

// one thread per pixel; the targets of the image are walked in order (later cones paint over earlier ones)
__global__ __launch_bounds__(256) void synth_images_kernel(unsigned seed, int step, int B, int T, int H, int W, const float* __restrict__ tg,
                                                           float* __restrict__ img) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const float noise = urand(seed, 21, ((unsigned long long)step * 4099ull + (unsigned long long)b) * (unsigned long long)(H * W) + (unsigned long long)p);
  const float base = 0.25f + 0.35f * ((float)y / (float)H) + 0.10f * noise;
  float c0 = base * 1.0f, c1 = base * (float)(1.0 - 0.08 * 1), c2 = base * (float)(1.0 - 0.08 * 2);
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  for (int t = 0; t < T; ++t) {
    const float* l = tg + ((size_t)b * T + t) * 5;
    const float cx = l[1], cy = l[2], w = l[3], h = l[4];
    if (!(w > 0.f && h > 0.f)) continue;
    const float x0 = (cx - w * 0.5f) * (float)W, x1 = (cx + w * 0.5f) * (float)W;
    const float y0 = (cy - h * 0.5f) * (float)H, y1 = (cy + h * 0.5f) * (float)H;
    const float v = (py - y0) / (y1 - y0);
    const float half = v * 0.5f * (x1 - x0);
    const float mid = (x0 + x1) * 0.5f;
    if (py >= y0 && py < y1 && fabsf(px - mid) <= half) {
      const bool stripe = v > 0.35f && v < 0.55f;
      const int cls = ((int)l[0]) % 3;
      c0 = stripe ? 0.95f : CONE_RGB[cls][0];
      c1 = stripe ? 0.95f : CONE_RGB[cls][1];
      c2 = stripe ? 0.95f : CONE_RGB[cls][2];
    }
  }
  const size_t plane = (size_t)H * W;
  float* o = img + (size_t)b * 3 * plane + p;
  o[0] = c0; o[plane] = c1; o[2 * plane] = c2;
}

// ---- key-point data: one workgroup per crop
__device__ __forceinline__ double resize_onehot(int d, int hot, int src, int dst) {   // cv2 INTER_LINEAR sample d of a one-hot at `hot`
  const double sc = (double)src / (double)dst;
  float fx = (float)(((double)d + 0.5) * sc - 0.5);
  int sx = (int)floorf(fx);
  fx = fx - (float)sx;
  if (sx < 0) { sx = 0; fx = 0.f; }
  if (sx >= src - 1) { sx = src - 1; fx = 0.f; }
  const int s1 = sx + 1 < src ? sx + 1 : src - 1;
  const double f = (double)fx;
  return (1.0 - f) * (sx == hot ? 1.0 : 0.0) + f * (s1 == hot ? 1.0 : 0.0);
}

template <int S>
__global__ __launch_bounds__(256) void synth_crops_kernel(unsigned seed, int step, int B, float* __restrict__ img, float* __restrict__ hm,
                                                          float* __restrict__ pts) {
  __shared__ double rz[2][7][S], bl[2][7][S], tot[7];
  __shared__ int hot[7][2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned long long key = (unsigned long long)step * 4099ull + (unsigned long long)b;
  const int oh = 24 + (int)(urand(seed, 31, key) * 57.0f);
  const int ow = 24 + (int)(urand(seed, 32, key) * 57.0f);
  const int cls = (int)(urand(seed, 33, key) * 3.0f);
  // image
  for (int p = tid; p < S * S; p += 256) {
    const int y = p / S, x = p - y * S;
    const float noise = urand(seed, 34, key * (unsigned long long)(S * S) + (unsigned long long)p);
    const float base = 0.30f + 0.25f * ((float)y / (float)S) + 0.10f * noise;
    const float px = ((float)x + 0.5f) / (float)S, py = ((float)y + 0.5f) / (float)S;
    const float half = py * 0.44f;
    const bool inside = fabsf(px - 0.5f) <= half && py >= 0.02f && py < 0.98f;
    const bool stripe = py > 0.38f && py < 0.58f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float bg = base * (float)(1.0 - 0.08 * c);
      img[((size_t)b * 3 + c) * S * S + p] = inside ? (stripe ? 0.95f : CONE_RGB[cls][c]) : bg;
    }
  }
  // key points (RektNet/utils.py:104-111 scale_labels, RektNet/dataset.py:43-44)
  if (tid < 7) {
    const unsigned long long kk = key * 8ull + (unsigned long long)tid;
    const float jx = (urand(seed, 35, kk) - 0.5f) * 0.04f, jy = (urand(seed, 36, kk) - 0.5f) * 0.04f;
    const float lx = fminf(fmaxf(KPX[tid] + jx, 0.f), 0.999f) * (float)ow;
    const float ly = fminf(fmaxf(KPY[tid] + jy, 0.f), 0.999f) * (float)oh;
    const int ix = (int)lx, iy = (int)ly;
    hot[tid][0] = ix; hot[tid][1] = iy;
    const double ws = (double)S / (double)ow, hs = (double)S / (double)oh;
    pts[((size_t)b * 7 + tid) * 2 + 0] = (float)(ceil((double)ix * ws) / (double)S);
    pts[((size_t)b * 7 + tid) * 2 + 1] = (float)(ceil((double)iy * hs) / (double)S);
  }
  __syncthreads();
  // heat-maps (RektNet/utils.py:83-97): resized one-hot and 5-tap blur are separable -> two vectors per key point
  for (int i = tid; i < 2 * 7 * S; i += 256) {
    const int a = i / (7 * S), r = i - a * 7 * S, k = r / S, d = r - k * S;     // a = 0: x axis, 1: y axis
    rz[a][k][d] = resize_onehot(d, hot[k][a], a == 0 ? ow : oh, S);
  }
  __syncthreads();
  for (int i = tid; i < 2 * 7 * S; i += 256) {
    const int a = i / (7 * S), r = i - a * 7 * S, k = r / S, d = r - k * S;
    const double g[5] = {1.0 / 16.0, 4.0 / 16.0, 6.0 / 16.0, 4.0 / 16.0, 1.0 / 16.0};
    double acc = 0.0;
#pragma unroll
    for (int t = -2; t <= 2; ++t) {
      int j = d + t;
      if (j < 0) j = -j;
      if (j >= S) j = 2 * (S - 1) - j;                                          // BORDER_REFLECT_101
      acc += g[t + 2] * rz[a][k][j];
    }
    bl[a][k][d] = acc;
  }
  __syncthreads();
  if (tid < 7) {
    double sy = 0.0, sx = 0.0;
    for (int d = 0; d < S; ++d) sy += bl[1][tid][d];
    for (int d = 0; d < S; ++d) sx += bl[0][tid][d];
    tot[tid] = sy * sx;
  }
  __syncthreads();
  for (int i = tid; i < 7 * S * S; i += 256) {
    const int k = i / (S * S), r = i - k * S * S, y = r / S, x = r - y * S;
    hm[(size_t)b * 7 * S * S + i] = (float)((bl[1][k][y] * bl[0][k][x]) / tot[k]);
  }
}

}  // namespace

extern "C" {

int mdcv_synth_cone_batch(unsigned seed, int step, int B, int T, int H, int W, int num_classes, float* images, float* targets, void* stream) {
  if (!images || !targets || B <= 0 || T <= 0 || H <= 0 || W <= 0 || num_classes <= 0 || B > 65535) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  MDCV_LAUNCH(synth_targets_kernel, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, st, seed, step, B, T, num_classes, targets);
  MDCV_CHECK_LAUNCH();
  MDCV_LAUNCH(synth_images_kernel, dim3((unsigned)((H * W + 255) / 256), (unsigned)B), dim3(256), 0, st, seed, step, B, T, H, W, targets, images);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_synth_crop_batch(unsigned seed, int step, int B, int size, float* images, float* heatmaps, float* points, void* stream) {
  if (!images || !heatmaps || !points || B <= 0 || size != 80) return MDCV_EARG;       // ConeDataset's target size (train_eval.py default)
  MDCV_LAUNCH(synth_crops_kernel<80>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, seed, step, B, images, heatmaps, points);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

// Detect -> crop -> keypoint glue: one launch cuts every kept detection box out of its frame and
// resamples it to the KeypointNet input size.  Semantics of the resample: RektNet/utils.py:73-76 `prep_image`
// (cv2.resize, default INTER_LINEAR) applied to the box crop, in OpenCV's float32 formulation: half-pixel centres,
// source coordinate evaluated in double and cast to float, taps clamped to the crop edge, horizontal blend then
// vertical blend.  Box coordinates go back to frame pixels like CVC-YOLOv3/detect.py:98-101 (x * scale + offset).
// Built with -ffp-contract=off so that the blends round like the oracle (two products, one sum).
#include "common.h"

#define MDCV_CROP_MAX_SIDE 256

namespace {

struct CropArgs {
  const float* frames; int B, C, H, W;
  const float* boxes; const int* count; int K;
  float sx, sy, ox, oy;
  int oh, ow;
  float* out; int* owner; int* total;
};

__device__ __forceinline__ void make_tap(int d, int dst, int src, int& i0, int& i1, float& w1) {
  const double sc = (double)src / (double)dst;
  float f = (float)(((double)d + 0.5) * sc - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= src - 1) { s = src - 1; f = 0.f; }
  i0 = s; i1 = s + 1 < src ? s + 1 : src - 1; w1 = f;
}

// grid (K, B): workgroup (k, b) resamples box k of frame b when k < count[b]; crops are packed image-major.
__global__ __launch_bounds__(256) void crop_resize_kernel(CropArgs A) {
  __shared__ int x0[MDCV_CROP_MAX_SIDE], x1[MDCV_CROP_MAX_SIDE], y0[MDCV_CROP_MAX_SIDE], y1[MDCV_CROP_MAX_SIDE];
  __shared__ float ax[MDCV_CROP_MAX_SIDE], ay[MDCV_CROP_MAX_SIDE];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int before = 0;
  for (int i = 0; i < b; ++i) { const int c = A.count[i]; before += c < A.K ? c : A.K; }
  int mine = A.count[b]; mine = mine < A.K ? mine : A.K;
  if (b == A.B - 1 && k == 0 && tid == 0) *A.total = before + mine;
  if (k >= mine) return;
  const int m = before + k;
  const float* bx = A.boxes + ((size_t)b * A.K + k) * 4;
  const float fx1 = bx[0] * A.sx + A.ox, fy1 = bx[1] * A.sy + A.oy, fx2 = bx[2] * A.sx + A.ox, fy2 = bx[3] * A.sy + A.oy;
  const int cx1 = (int)fminf(fmaxf(floorf(fx1), 0.f), (float)(A.W - 1)), cy1 = (int)fminf(fmaxf(floorf(fy1), 0.f), (float)(A.H - 1));
  const int cx2 = (int)fminf(fmaxf(ceilf(fx2), (float)(cx1 + 1)), (float)A.W), cy2 = (int)fminf(fmaxf(ceilf(fy2), (float)(cy1 + 1)), (float)A.H);
  const int cw = cx2 - cx1, ch = cy2 - cy1;
  if (tid < A.ow) { int i0, i1; float w; make_tap(tid, A.ow, cw, i0, i1, w); x0[tid] = cx1 + i0; x1[tid] = cx1 + i1; ax[tid] = w; }
  if (tid < A.oh) { int i0, i1; float w; make_tap(tid, A.oh, ch, i0, i1, w); y0[tid] = cy1 + i0; y1[tid] = cy1 + i1; ay[tid] = w; }
  if (tid == 0) A.owner[m] = b;
  __syncthreads();
  const int plane = A.oh * A.ow;
  float* o = A.out + (size_t)m * A.C * plane;
  for (int c = 0; c < A.C; ++c) {
    const float* src = A.frames + ((size_t)b * A.C + c) * A.H * A.W;
    for (int i = tid; i < plane; i += 256) {
      const int y = i / A.ow, x = i - y * A.ow;
      const float a1 = ax[x], a0 = 1.f - a1, b1 = ay[y], b0 = 1.f - b1;
      const float* r0 = src + (size_t)y0[y] * A.W;
      const float* r1 = src + (size_t)y1[y] * A.W;
      const float h0 = r0[x0[x]] * a0 + r0[x1[x]] * a1;
      const float h1 = r1[x0[x]] * a0 + r1[x1[x]] * a1;
      o[(size_t)c * plane + i] = h0 * b0 + h1 * b1;
    }
  }
}

// ---- the rule the reference actually runs: RektNet/dataset.py:35-38,52 and RektNet/detect.py:29-35 read a uint8 BGR image
// (cv2.imread), resize THAT with cv2.resize (utils.py:73-76), and only then divide by 255.  For 8-bit images OpenCV's INTER_LINEAR
// is fixed point (resize.cpp: HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>):
//   fx = (float)((dx + 0.5) * scale - 0.5), scale = 1. / ((double)dst / src);  sx = floor(fx);  fx -= sx;
//   x axis: sx < 0 -> (0, fx = 0);  sx >= src - 1 -> (src - 1, fx = 0);  y axis: weights kept, ROWS clamped (sy, sy + 1 -> [0, src - 1])
//   coefficients (short) = round-half-even((1 - f) * 2048), round-half-even(f * 2048)
//   horizontal: D = S[sx] * a0 + S[sx + 1] * a1 (int);   vertical: u8(((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2)
//   output = (float)(u8 / 255.0)           (numpy divides in float64, `.type('torch.FloatTensor')` rounds once)
struct CropU8Args {
  const unsigned char* frames; int B, C, H, W;
  const float* boxes; const int* count; int K;
  float sx, sy, ox, oy;
  int oh, ow;
  float* out; int* owner; int* total;
};

__device__ __forceinline__ void make_tap_u8(int d, int dst, int src, bool clamp_weights, int& i0, int& i1, int& c0, int& c1) {
  const double sc = 1.0 / ((double)dst / (double)src);
  float f = (float)(((double)d + 0.5) * sc - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (clamp_weights) {
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src - 1) { s = src - 1; f = 0.f; }
  }
  c0 = (int)rintf((1.f - f) * 2048.f);
  c1 = (int)rintf(f * 2048.f);
  i0 = s < 0 ? 0 : (s < src ? s : src - 1);
  i1 = s + 1 < 0 ? 0 : (s + 1 < src ? s + 1 : src - 1);
}

__global__ __launch_bounds__(256) void crop_resize_u8_kernel(CropU8Args A) {
  __shared__ int x0[MDCV_CROP_MAX_SIDE], x1[MDCV_CROP_MAX_SIDE], y0[MDCV_CROP_MAX_SIDE], y1[MDCV_CROP_MAX_SIDE];
  __shared__ int a0s[MDCV_CROP_MAX_SIDE], a1s[MDCV_CROP_MAX_SIDE], b0s[MDCV_CROP_MAX_SIDE], b1s[MDCV_CROP_MAX_SIDE];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int before = 0;
  for (int i = 0; i < b; ++i) { const int c = A.count[i]; before += c < A.K ? c : A.K; }
  int mine = A.count[b]; mine = mine < A.K ? mine : A.K;
  if (b == A.B - 1 && k == 0 && tid == 0) *A.total = before + mine;
  if (k >= mine) return;
  const int m = before + k;
  const float* bx = A.boxes + ((size_t)b * A.K + k) * 4;
  const float fx1 = bx[0] * A.sx + A.ox, fy1 = bx[1] * A.sy + A.oy, fx2 = bx[2] * A.sx + A.ox, fy2 = bx[3] * A.sy + A.oy;
  const int cx1 = (int)fminf(fmaxf(floorf(fx1), 0.f), (float)(A.W - 1)), cy1 = (int)fminf(fmaxf(floorf(fy1), 0.f), (float)(A.H - 1));
  const int cx2 = (int)fminf(fmaxf(ceilf(fx2), (float)(cx1 + 1)), (float)A.W), cy2 = (int)fminf(fmaxf(ceilf(fy2), (float)(cy1 + 1)), (float)A.H);
  const int cw = cx2 - cx1, ch = cy2 - cy1;
  if (tid < A.ow) { int i0, i1, c0, c1; make_tap_u8(tid, A.ow, cw, true, i0, i1, c0, c1); x0[tid] = cx1 + i0; x1[tid] = cx1 + i1; a0s[tid] = c0; a1s[tid] = c1; }
  if (tid < A.oh) { int i0, i1, c0, c1; make_tap_u8(tid, A.oh, ch, false, i0, i1, c0, c1); y0[tid] = cy1 + i0; y1[tid] = cy1 + i1; b0s[tid] = c0; b1s[tid] = c1; }
  if (tid == 0) A.owner[m] = b;
  __syncthreads();
  const int plane = A.oh * A.ow;
  float* o = A.out + (size_t)m * A.C * plane;
  for (int c = 0; c < A.C; ++c) {
    const unsigned char* src = A.frames + ((size_t)b * A.C + c) * A.H * A.W;
    for (int i = tid; i < plane; i += 256) {
      const int y = i / A.ow, x = i - y * A.ow;
      const unsigned char* r0 = src + (size_t)y0[y] * A.W;
      const unsigned char* r1 = src + (size_t)y1[y] * A.W;
      const int d0 = (int)r0[x0[x]] * a0s[x] + (int)r0[x1[x]] * a1s[x];
      const int d1 = (int)r1[x0[x]] * a0s[x] + (int)r1[x1[x]] * a1s[x];
      const int v = (((b0s[y] * (d0 >> 4)) >> 16) + ((b1s[y] * (d1 >> 4)) >> 16) + 2) >> 2;
      o[(size_t)c * plane + i] = (float)((double)(unsigned char)v / 255.0);
    }
  }
}

}  // namespace

extern "C" {

int mdcv_crop_resize(const float* frames, int B, int C, int H, int W, const float* boxes, const int* count, int K, float scale_x,
                     float scale_y, float off_x, float off_y, int out_h, int out_w, float* out, int* owner, int* total, void* stream) {
  if (!frames || !boxes || !count || !out || !owner || !total) return MDCV_EARG;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || out_h <= 0 || out_w <= 0 || out_h > MDCV_CROP_MAX_SIDE || out_w > MDCV_CROP_MAX_SIDE ||
      B > 65535)
    return MDCV_EARG;
  CropArgs a{frames, B, C, H, W, boxes, count, K, scale_x, scale_y, off_x, off_y, out_h, out_w, out, owner, total};
  MDCV_LAUNCH(crop_resize_kernel, dim3(K, B), dim3(256), 0, (hipStream_t)stream, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_crop_resize_u8(const unsigned char* frames, int B, int C, int H, int W, const float* boxes, const int* count, int K, float scale_x,
                        float scale_y, float off_x, float off_y, int out_h, int out_w, float* out, int* owner, int* total, void* stream) {
  if (!frames || !boxes || !count || !out || !owner || !total) return MDCV_EARG;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || out_h <= 0 || out_w <= 0 || out_h > MDCV_CROP_MAX_SIDE || out_w > MDCV_CROP_MAX_SIDE ||
      B > 65535)
    return MDCV_EARG;
  CropU8Args a{frames, B, C, H, W, boxes, count, K, scale_x, scale_y, off_x, off_y, out_h, out_w, out, owner, total};
  MDCV_LAUNCH(crop_resize_u8_kernel, dim3(K, B), dim3(256), 0, (hipStream_t)stream, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

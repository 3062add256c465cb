// Fused YOLO detection head for gfx950: grid/anchor target assignment, 6-part loss, d(loss)/d(logits), eval decode.
//
// Replaces YOLOLayer.forward (CVC-YOLOv3/models.py:140-220) and build_targets / bbox_iou
// (CVC-YOLOv3/utils/utils.py:163-275), ~150 tiny ATen launches per step in the reference, with 4 launches per head.
// Integer assignment (gi, gj, best anchor, masks, collision winner) is BIT-EXACT with the reference: every fp32
// operation below is written in the reference's order and this file is compiled with -ffp-contract=off so no
// mul+add pair is fused into an fma.
//
// logits are NHWC: channel = a*(5+C) + attr, channel stride ldc (pad channels ignored on read, zero on write).
#include "common.h"

namespace {

struct HeadGeom {
  int B, T, A, C, Gh, Gw, ldc;
  float thresh;
};

// row (b,t) -> source row (padding rows take row 0, utils.py:223-228), grid coords, best anchor, "any IoU > thresh"
__device__ __forceinline__ void assign_row(const float* __restrict__ tg, const float* __restrict__ anchors, const HeadGeom& g, int b, int t,
                                           int& src, int& gi, int& gj, int& best, bool& over, float& gx, float& gy, float& gw, float& gh) {
  const float* r = tg + ((size_t)b * g.T + t) * 5;
  float s = r[0]; s = s + r[1]; s = s + r[2]; s = s + r[3]; s = s + r[4];       // utils.py:210
  src = s > 0.f ? t : 0;
  const float* q = tg + ((size_t)b * g.T + src) * 5;
  gx = q[1] * (float)g.Gw; gy = q[2] * (float)g.Gh; gw = q[3] * (float)g.Gw; gh = q[4] * (float)g.Gh;   // :213-216
  gi = (int)gx; gj = (int)gy;                                                    // :219-220 truncation
  best = 0; over = false;
  float best_iou = -1.f;
  for (int a = 0; a < g.A; ++a) {                                               // :236-240 via bbox_iou :178-191
    const float aw = anchors[2 * a], ah = anchors[2 * a + 1];
    const float ix = fminf(gw, aw) - 0.f + 1.f, iy = fminf(gh, ah) - 0.f + 1.f;
    const float inter = fmaxf(ix, 0.f) * fmaxf(iy, 0.f);
    const float a1 = (gw - 0.f + 1.f) * (gh - 0.f + 1.f);
    const float a2 = (aw - 0.f + 1.f) * (ah - 0.f + 1.f);
    const float iou = inter / (a1 + a2 - inter + 1e-12f);
    if (iou > g.thresh) over = true;
    if (iou > best_iou) { best_iou = iou; best = a; }                            // :257 first max wins
  }
}

// owner[b,a,j,i] = highest t assigned to the cell (sequential index_put => last writer wins), ignore[j,i] batch-wide
__global__ void yolo_assign_kernel(const float* __restrict__ tg, const float* __restrict__ anchors, HeadGeom g, int* __restrict__ owner,
                                   int* __restrict__ ignore, int* __restrict__ rowinfo, int* __restrict__ err) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < g.B * g.T; idx += gridDim.x * blockDim.x) {
    const int b = idx / g.T, t = idx - b * g.T;
    int src, gi, gj, best; bool over; float gx, gy, gw, gh;
    assign_row(tg, anchors, g, b, t, src, gi, gj, best, over, gx, gy, gw, gh);
    if (gi < 0 || gi >= g.Gw || gj < 0 || gj >= g.Gh) {        // the reference raises IndexError here (Q5)
      atomicExch(err, 1);
      if (rowinfo) rowinfo[idx] = -1;
      continue;
    }
    if (over) ignore[gj * g.Gw + gi] = 1;                       // :244-255, all images / all anchors
    atomicMax(&owner[(((size_t)b * g.A + best) * g.Gh + gj) * g.Gw + gi], t);
    if (rowinfo) rowinfo[idx] = (best << 24) | (gj << 12) | gi;
  }
}

static unsigned grid_for(long long n) { long long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// out[0] += head loss ; out[1..6] += (x,y,w,h,obj,noobj) parts (models.py:199-211,332,338).  A launch of its own: folding it into the
// last block of the loss kernel needs an agent-scope fence per block, and on eight L2s that tripled the loss kernel (11 -> 33 us).
__global__ void yolo_finalize_kernel(const double* __restrict__ acc, float xy_loss, float wh_loss, float obj_loss, float noobj_loss,
                                     float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double nM = acc[6], nN = acc[7];
  const float lx = xy_loss * (float)(acc[0] / nM), ly = xy_loss * (float)(acc[1] / nM);      // mean over an empty selection = NaN,
  const float lw = wh_loss * (float)(acc[2] / nM), lh = wh_loss * (float)(acc[3] / nM);      // exactly like the reference
  const float lob = obj_loss * (float)(acc[4] / nM), lno = noobj_loss * (float)(acc[5] / nN);
  const float loss = lx + ly + lw + lh + lno + lob;
  out[0] += loss;
  out[1] += lx; out[2] += ly; out[3] += lw; out[4] += lh; out[5] += lob; out[6] += lno;
}

// workspace of one head before the assignment: owner = -1 (no target), everything behind it (ignore, err, acc) = 0
__global__ void yolo_init_kernel(int* __restrict__ ws, long long cells, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) ws[i] = i < cells ? -1 : 0;
}

struct LossArgs {
  const void* logits; const float* tg; const float* anchors; const int* owner; const int* ignore; double* acc;
  HeadGeom g;
};

// per cell: masked MSE / BCE terms -> acc[0..5] = sums (x,y,w,h,obj,noobj), acc[6] = nM, acc[7] = nN
template <typename T>
__global__ __launch_bounds__(256) void yolo_loss_kernel(LossArgs a) {
  const HeadGeom& g = a.g;
  const int attrs = 5 + g.C;
  const long long cells = (long long)g.B * g.A * g.Gh * g.Gw;
  const T* lg = reinterpret_cast<const T*>(a.logits);
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.0;
  for (long long cidx = blockIdx.x * (long long)blockDim.x + threadIdx.x; cidx < cells; cidx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(cidx % g.Gw);
    long long r = cidx / g.Gw;
    const int j = (int)(r % g.Gh); r /= g.Gh;
    const int an = (int)(r % g.A), b = (int)(r / g.A);
    const int t = a.owner[cidx];
    const bool pos = t >= 0;
    const bool neg = !pos && a.ignore[j * g.Gw + i] == 0;        // conf_mask - mask (models.py:196)
    if (!pos && !neg) continue;
    const T* px = lg + (((size_t)b * g.Gh + j) * g.Gw + i) * g.ldc + an * attrs;
    const float pc = sigmoidf_(ET<T>::ld(px + 4));
    if (pos) {
      int src, gi, gj, best; bool over; float gx, gy, gw, gh;
      assign_row(a.tg, a.anchors, g, b, t, src, gi, gj, best, over, gx, gy, gw, gh);
      const float tx = gx - (float)gi, ty = gy - (float)gj;                         // utils.py:265-266
      const float tw = logf(gw / a.anchors[2 * an] + 1e-16f), th = logf(gh / a.anchors[2 * an + 1] + 1e-16f);   // :268-269
      const float dx = sigmoidf_(ET<T>::ld(px + 0)) - tx, dy = sigmoidf_(ET<T>::ld(px + 1)) - ty;
      const float dw = ET<T>::ld(px + 2) - tw, dh = ET<T>::ld(px + 3) - th;
      v[0] += (double)(dx * dx); v[1] += (double)(dy * dy); v[2] += (double)(dw * dw); v[3] += (double)(dh * dh);
      v[4] += (double)(-fmaxf(logf(pc), -100.f));                                   // BCE(p,1), log clamped like torch
      v[6] += 1.0;
    } else {
      v[5] += (double)(-fmaxf(logf(1.f - pc), -100.f));                             // BCE(p,0)
      v[7] += 1.0;
    }
  }
  __shared__ double red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double s = wave_sum_d(v[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (s != 0.0) atomicAdd(&a.acc[threadIdx.x], s);
  }
}

struct GradArgs {
  const void* logits; void* dlogits; const float* tg; const float* anchors; const int* owner; const int* ignore; const double* acc;
  const float* gscale;
  HeadGeom g; int ldd, Cpad;
  float xy_loss, wh_loss, obj_loss, noobj_loss;
};

// d(loss)/d(logits) of one (pixel, channel) element; class channels and pad channels are exact zeros
template <typename T>
__device__ __forceinline__ float yolo_grad_elem(const GradArgs& a, const T* __restrict__ lg, long long pix, int ch, float nM, float nN) {
  const HeadGeom& g = a.g;
  const int attrs = 5 + g.C;
  float gr = 0.f;
  const int an = ch / attrs, at = ch - an * attrs;
  if (an < g.A && at < 5) {
    const int i = (int)(pix % g.Gw);
    long long r = pix / g.Gw;
    const int j = (int)(r % g.Gh), b = (int)(r / g.Gh);
    const int t = a.owner[(((size_t)b * g.A + an) * g.Gh + j) * g.Gw + i];
    const float s = ET<T>::ld(lg + pix * g.ldc + ch);
    if (t >= 0) {
      int src, gi, gj, best; bool over; float gx, gy, gw, gh;
      assign_row(a.tg, a.anchors, g, b, t, src, gi, gj, best, over, gx, gy, gw, gh);
      if (at == 0) { const float p = sigmoidf_(s); gr = a.xy_loss * 2.f * (p - (gx - (float)gi)) / nM * p * (1.f - p); }
      else if (at == 1) { const float p = sigmoidf_(s); gr = a.xy_loss * 2.f * (p - (gy - (float)gj)) / nM * p * (1.f - p); }
      else if (at == 2) gr = a.wh_loss * 2.f * (s - logf(gw / a.anchors[2 * an] + 1e-16f)) / nM;
      else if (at == 3) gr = a.wh_loss * 2.f * (s - logf(gh / a.anchors[2 * an + 1] + 1e-16f)) / nM;
      else { const float p = sigmoidf_(s), pq = p * (1.f - p); gr = a.obj_loss * (p - 1.f) * pq / fmaxf(pq, 1e-12f) / nM; }
    } else if (at == 4 && a.ignore[j * g.Gw + i] == 0) {
      const float p = sigmoidf_(s), pq = p * (1.f - p);
      gr = a.noobj_loss * p * pq / fmaxf(pq, 1e-12f) / nN;
    }
  }
  return gr;
}

template <typename T>
__global__ __launch_bounds__(256) void yolo_grad_kernel(GradArgs a) {
  const HeadGeom& g = a.g;
  const long long total = (long long)g.B * g.Gh * g.Gw * a.Cpad;
  const T* lg = reinterpret_cast<const T*>(a.logits);
  T* dl = reinterpret_cast<T*>(a.dlogits);
  const float nM = (float)a.acc[6], nN = (float)a.acc[7];
  const float up = a.gscale ? a.gscale[0] : 1.f;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(e % a.Cpad);
    const long long pix = e / a.Cpad;
    ET<T>::st(dl + pix * a.ldd + ch, yolo_grad_elem<T>(a, lg, pix, ch, nM, nN) * up);
  }
}

// The same values through an LDS tile of kGradPix pixels x Cpad channels: one thread per (pixel, anchor, x/y/w/h/conf) element computes
// its gradient (the owner / logit loads of a block are all in flight together instead of chained inside one thread), then the tile
// leaves as 16-byte stores.  With 80 classes 15 of the 255 channels carry a gradient, so the kernel is close to a fill of the
// dlogits plane.  Needs Cpad % V == 0, ldd % V == 0, a 16-byte aligned base and a tile that fits LDS (checked by the launcher).
constexpr int kGradPix = 16;
template <typename T>
__global__ __launch_bounds__(256) void yolo_grad_tile_kernel(GradArgs a) {
  constexpr int V = 16 / (int)sizeof(T);
  extern __shared__ uint4 tile_raw[];
  T* tile = reinterpret_cast<T*>(tile_raw);
  const HeadGeom& g = a.g;
  const int attrs = 5 + g.C;
  const int nvec = a.Cpad / V;
  const long long pixels = (long long)g.B * g.Gh * g.Gw;
  const long long pix0 = (long long)blockIdx.x * kGradPix;
  const int np = (int)(pixels - pix0 < kGradPix ? pixels - pix0 : kGradPix);
  const T* lg = reinterpret_cast<const T*>(a.logits);
  T* dl = reinterpret_cast<T*>(a.dlogits);
  const float nM = (float)a.acc[6], nN = (float)a.acc[7];
  const float up = a.gscale ? a.gscale[0] : 1.f;
  alignas(16) T z[V];
#pragma unroll
  for (int k = 0; k < V; ++k) ET<T>::st(z + k, 0.f * up);
  for (int v = threadIdx.x; v < np * nvec; v += blockDim.x) tile_raw[v] = *reinterpret_cast<const uint4*>(z);
  __syncthreads();
  const int live = g.A * 5;
  for (int idx = threadIdx.x; idx < np * live; idx += blockDim.x) {
    const int p = idx / live, r = idx - p * live;
    const int an = r / 5, ch = an * attrs + (r - an * 5);
    ET<T>::st(tile + p * a.Cpad + ch, yolo_grad_elem<T>(a, lg, pix0 + p, ch, nM, nN) * up);
  }
  __syncthreads();
  for (int v = threadIdx.x; v < np * nvec; v += blockDim.x) {
    const int p = v / nvec, c = v - p * nvec;
    *reinterpret_cast<uint4*>(dl + (size_t)(pix0 + p) * a.ldd + c * V) = tile_raw[v];
  }
}

template <typename T>
static void launch_yolo_grad(const GradArgs& ga, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const HeadGeom& g = ga.g;
  const long long pixels = (long long)g.B * g.Gh * g.Gw;
  const size_t lds = (size_t)kGradPix * ga.Cpad * sizeof(T);
  const long long blocks = (pixels + kGradPix - 1) / kGradPix;
  const bool tiled = ga.Cpad % V == 0 && ga.ldd % V == 0 && ((uintptr_t)ga.dlogits & 15) == 0 && lds <= 32768 && blocks < (1LL << 31);
  if (tiled) MDCV_LAUNCH(yolo_grad_tile_kernel<T>, dim3((unsigned)blocks), dim3(256), lds, st, ga);
  else MDCV_LAUNCH(yolo_grad_kernel<T>, dim3(grid_for(pixels * ga.Cpad)), dim3(256), 0, st, ga);
}

// eval: [B, rows_total, 5+C] fp32, this head's rows start at row_off ; row = (a*Gh + j)*Gw + i (models.py:215-220)
template <typename T>
__global__ void yolo_decode_kernel(const T* __restrict__ lg, int ldc, const float* __restrict__ anchors, float stride, int B, int A, int C,
                                   int Gh, int Gw, float* __restrict__ out, int rows_total, int row_off) {
  const int attrs = 5 + C;
  const long long total = (long long)B * A * Gh * Gw * attrs;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int at = (int)(e % attrs);
    long long r = e / attrs;
    const int i = (int)(r % Gw); r /= Gw;
    const int j = (int)(r % Gh); r /= Gh;
    const int an = (int)(r % A), b = (int)(r / A);
    const float s = ET<T>::ld(lg + (((size_t)b * Gh + j) * Gw + i) * ldc + an * attrs + at);
    float v;
    if (at == 0) v = (sigmoidf_(s) + (float)i) * stride;
    else if (at == 1) v = (sigmoidf_(s) + (float)j) * stride;
    else if (at == 2) v = expf(s) * anchors[2 * an] * stride;
    else if (at == 3) v = expf(s) * anchors[2 * an + 1] * stride;
    else v = sigmoidf_(s);
    out[((size_t)b * rows_total + row_off + ((size_t)an * Gh + j) * Gw + i) * attrs + at] = v;
  }
}

// ---- standalone build_targets API (utils.py:195-275): the 8 dense tensors
__global__ void bt_dense_kernel(const float* __restrict__ tg, const float* __restrict__ anchors, HeadGeom g, const int* __restrict__ owner,
                                const int* __restrict__ ignore, unsigned char* __restrict__ mask, unsigned char* __restrict__ conf_mask,
                                float* __restrict__ tx, float* __restrict__ ty, float* __restrict__ tw, float* __restrict__ th,
                                float* __restrict__ tconf) {
  const long long cells = (long long)g.B * g.A * g.Gh * g.Gw;
  for (long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x; c < cells; c += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(c % g.Gw);
    long long r = c / g.Gw;
    const int j = (int)(r % g.Gh); r /= g.Gh;
    const int an = (int)(r % g.A), b = (int)(r / g.A);
    const int t = owner[c];
    float vx = 0.f, vy = 0.f, vw = 0.f, vh = 0.f;
    if (t >= 0) {
      int src, gi, gj, best; bool over; float gx, gy, gw, gh;
      assign_row(tg, anchors, g, b, t, src, gi, gj, best, over, gx, gy, gw, gh);
      vx = gx - (float)gi; vy = gy - (float)gj;
      vw = logf(gw / anchors[2 * an] + 1e-16f); vh = logf(gh / anchors[2 * an + 1] + 1e-16f);
    }
    mask[c] = t >= 0;
    conf_mask[c] = (t >= 0) || ignore[j * g.Gw + i] == 0;
    tx[c] = vx; ty[c] = vy; tw[c] = vw; th[c] = vh; tconf[c] = t >= 0 ? 1.f : 0.f;
  }
}
__global__ void bt_tcls_kernel(const float* __restrict__ tg, HeadGeom g, const int* __restrict__ rowinfo, unsigned char* __restrict__ tcls) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < g.B * g.T; idx += gridDim.x * blockDim.x) {
    const int info = rowinfo[idx];
    if (info < 0) continue;
    const int b = idx / g.T;
    const int best = info >> 24, gj = (info >> 12) & 0xfff, gi = info & 0xfff;
    const int label = (int)tg[(size_t)idx * 5];                 // label of the ROW itself, padding rows included (:271)
    if (label >= 0 && label < g.C) tcls[((((size_t)b * g.A + best) * g.Gh + gj) * g.Gw + gi) * g.C + label] = 1;
  }
}

// utils.bbox_iou (CVC-YOLOv3/utils/utils.py:163-193) as the reference writes it, one thread per pair: boxes as rows of `stride` floats (the
// first four are read), either side broadcast when it has one row.  Every fp32 operation rounds where the reference's torch op rounds (this
// file is compiled with -ffp-contract=off): bit-identical to the reference on the same inputs.
__global__ __launch_bounds__(256) void bbox_iou_kernel(const float* __restrict__ b1, long long n1, int s1, const float* __restrict__ b2, long long n2,
                                                       int s2, int corners, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = b1 + (n1 == 1 ? 0 : i) * s1;
  const float* q = b2 + (n2 == 1 ? 0 : i) * s2;
  float ax1, ay1, ax2, ay2, bx1, by1, bx2, by2;
  if (corners) {
    ax1 = p[0]; ay1 = p[1]; ax2 = p[2]; ay2 = p[3];
    bx1 = q[0]; by1 = q[1]; bx2 = q[2]; by2 = q[3];
  } else {                                                                       // :168-173 centre / size -> corners
    ax1 = p[0] - p[2] / 2; ax2 = p[0] + p[2] / 2; ay1 = p[1] - p[3] / 2; ay2 = p[1] + p[3] / 2;
    bx1 = q[0] - q[2] / 2; bx2 = q[0] + q[2] / 2; by1 = q[1] - q[3] / 2; by2 = q[1] + q[3] / 2;
  }
  // torch.max / torch.min propagate NaN; fmaxf / fminf would drop it
  const float ix1 = (ax1 != ax1 || bx1 != bx1) ? __builtin_nanf("") : (ax1 > bx1 ? ax1 : bx1);
  const float iy1 = (ay1 != ay1 || by1 != by1) ? __builtin_nanf("") : (ay1 > by1 ? ay1 : by1);
  const float ix2 = (ax2 != ax2 || bx2 != bx2) ? __builtin_nanf("") : (ax2 < bx2 ? ax2 : bx2);
  const float iy2 = (ay2 != ay2 || by2 != by2) ? __builtin_nanf("") : (ay2 < by2 ? ay2 : by2);
  float w = ix2 - ix1 + 1.f, h = iy2 - iy1 + 1.f;                                // :184-186 clamp(min=0) keeps NaN
  w = w < 0.f ? 0.f : w; h = h < 0.f ? 0.f : h;
  const float inter = w * h;
  const float a1 = (ax2 - ax1 + 1.f) * (ay2 - ay1 + 1.f);                        // :188-189
  const float a2 = (bx2 - bx1 + 1.f) * (by2 - by1 + 1.f);
  out[i] = inter / (a1 + a2 - inter + 1e-12f);                                   // :191
}

}  // namespace

extern "C" {
// IoU with the reference's "+1 pixel" convention (utils/utils.py:163-193): n = max(n1, n2) pairs, a side with ONE row is broadcast; rows are
// `stride` floats apart and their first four floats are the box (corners = 1: x1 y1 x2 y2, 0: cx cy w h).  fp32, bit-identical to the reference.
int mdcv_bbox_iou(const float* box1, long long n1, int stride1, const float* box2, long long n2, int stride2, int corners, float* out, void* stream) {
  if (!box1 || !box2 || !out || n1 < 1 || n2 < 1 || stride1 < 4 || stride2 < 4 || (n1 != n2 && n1 != 1 && n2 != 1)) return MDCV_EARG;
  const long long n = n1 > n2 ? n1 : n2;
  if (n > (1LL << 31) * 256 - 256) return MDCV_EARG;
  MDCV_LAUNCH(bbox_iou_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, box1, n1, stride1, box2, n2, stride2, corners, out, n);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}


// workspace (ints): owner[B*A*Gh*Gw] | ignore[Gh*Gw] | err[1] (+pad) ; then acc[8] doubles (8-byte aligned by construction)
long long mdcv_yolo_head_workspace_bytes(int B, int A, int Gh, int Gw) {
  long long ints = (long long)B * A * Gh * Gw + (long long)Gh * Gw + 2;
  ints = (ints + 1) & ~1LL;
  return ints * 4 + 8 * 8;
}

// Training step of one head: accumulates out7[0] += loss, out7[1..6] += parts and writes dlogits (scaled by *gscale if given).
int mdcv_yolo_head_train(int dtype, const void* logits, int ldc, void* dlogits, int ldd, int Cpad, const float* targets,
                         const float* anchors_scaled, int B, int T, int A, int C, int Gh, int Gw, float thresh, float xy_loss,
                         float wh_loss, float obj_loss, float noobj_loss, void* workspace, float* out7, const float* gscale,
                         void* stream) {
  if (!logits || !targets || !anchors_scaled || !workspace || !out7) return MDCV_EARG;
  if (Gh > 4095 || Gw > 4095 || A > 127 || A * (5 + C) > Cpad) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  HeadGeom g{B, T, A, C, Gh, Gw, ldc, thresh};
  const long long cells = (long long)B * A * Gh * Gw;
  long long ints = cells + (long long)Gh * Gw + 2; ints = (ints + 1) & ~1LL;
  int* owner = (int*)workspace; int* ignore = owner + cells; int* err = ignore + (long long)Gh * Gw;
  double* acc = (double*)((int*)workspace + ints);
  MDCV_LAUNCH(yolo_init_kernel, dim3(grid_for(ints + 16)), dim3(256), 0, st, owner, cells, ints + 16);
  MDCV_CHECK_LAUNCH();
  MDCV_LAUNCH(yolo_assign_kernel, dim3(grid_for((long long)B * T)), dim3(256), 0, st, targets, anchors_scaled, g, owner, ignore, (int*)nullptr, err);
  MDCV_CHECK_LAUNCH();
  LossArgs la{logits, targets, anchors_scaled, owner, ignore, acc, g};
  if (dtype == MDCV_BF16) MDCV_LAUNCH(yolo_loss_kernel<bf16_t>, dim3(grid_for(cells)), dim3(256), 0, st, la);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(yolo_loss_kernel<float>, dim3(grid_for(cells)), dim3(256), 0, st, la);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  MDCV_LAUNCH(yolo_finalize_kernel, dim3(1), dim3(64), 0, st, acc, xy_loss, wh_loss, obj_loss, noobj_loss, out7);
  MDCV_CHECK_LAUNCH();
  if (dlogits) {
    GradArgs ga{logits, dlogits, targets, anchors_scaled, owner, ignore, acc, gscale, g, ldd, Cpad, xy_loss, wh_loss, obj_loss, noobj_loss};
    if (dtype == MDCV_BF16) launch_yolo_grad<bf16_t>(ga, st);
    else launch_yolo_grad<float>(ga, st);
    MDCV_CHECK_LAUNCH();
  }
  return MDCV_OK;
}

// backward of one head: d loss / d logits from the assignment + counts the forward call left in `workspace`
int mdcv_yolo_head_grad(int dtype, const void* logits, int ldc, void* dlogits, int ldd, int Cpad, const float* targets,
                        const float* anchors_scaled, int B, int T, int A, int C, int Gh, int Gw, float thresh, float xy_loss,
                        float wh_loss, float obj_loss, float noobj_loss, void* workspace, const float* gscale, void* stream) {
  if (!logits || !dlogits || !targets || !anchors_scaled || !workspace) return MDCV_EARG;
  if (A * (5 + C) > Cpad) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  HeadGeom g{B, T, A, C, Gh, Gw, ldc, thresh};
  const long long cells = (long long)B * A * Gh * Gw;
  long long ints = cells + (long long)Gh * Gw + 2; ints = (ints + 1) & ~1LL;
  int* owner = (int*)workspace; int* ignore = owner + cells;
  double* acc = (double*)((int*)workspace + ints);
  GradArgs ga{logits, dlogits, targets, anchors_scaled, owner, ignore, acc, gscale, g, ldd, Cpad, xy_loss, wh_loss, obj_loss, noobj_loss};
  if (dtype == MDCV_BF16) launch_yolo_grad<bf16_t>(ga, st);
  else if (dtype == MDCV_F32) launch_yolo_grad<float>(ga, st);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_yolo_head_decode(int dtype, const void* logits, int ldc, const float* anchors_scaled, float stride, int B, int A, int C, int Gh,
                          int Gw, float* out, int rows_total, int row_off, void* stream) {
  if (!logits || !anchors_scaled || !out) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)B * A * Gh * Gw * (5 + C);
  if (dtype == MDCV_BF16) MDCV_LAUNCH(yolo_decode_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)logits, ldc, anchors_scaled, stride, B, A, C, Gh, Gw, out, rows_total, row_off);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(yolo_decode_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)logits, ldc, anchors_scaled, stride, B, A, C, Gh, Gw, out, rows_total, row_off);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// utils.utils.build_targets drop-in: dense outputs; workspace ints: owner[cells] | ignore[Gh*Gw] | err[2] | rowinfo[B*T]
long long mdcv_build_targets_workspace_bytes(int B, int T, int A, int Gh, int Gw) {
  return ((long long)B * A * Gh * Gw + (long long)Gh * Gw + 2 + (long long)B * T) * 4;
}
int mdcv_build_targets(const float* targets, const float* anchors, int B, int T, int A, int C, int Gh, int Gw, float thresh,
                       unsigned char* mask, unsigned char* conf_mask, float* tx, float* ty, float* tw, float* th, float* tconf,
                       unsigned char* tcls, void* workspace, int* err_out, void* stream) {
  if (!targets || !anchors || !workspace || !mask || !conf_mask || !tcls) return MDCV_EARG;
  if (Gh > 4095 || Gw > 4095 || A > 127) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  HeadGeom g{B, T, A, C, Gh, Gw, 0, thresh};
  const long long cells = (long long)B * A * Gh * Gw;
  int* owner = (int*)workspace; int* ignore = owner + cells; int* err = ignore + (long long)Gh * Gw; int* rowinfo = err + 2;
  hipError_t e = hipMemsetAsync(owner, 0xff, cells * 4, st); if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(ignore, 0, ((long long)Gh * Gw + 2) * 4, st); if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(tcls, 0, cells * C, st); if (e != hipSuccess) return (int)e;
  MDCV_LAUNCH(yolo_assign_kernel, dim3(grid_for((long long)B * T)), dim3(256), 0, st, targets, anchors, g, owner, ignore, rowinfo, err);
  MDCV_CHECK_LAUNCH();
  MDCV_LAUNCH(bt_dense_kernel, dim3(grid_for(cells)), dim3(256), 0, st, targets, anchors, g, owner, ignore, mask, conf_mask, tx, ty, tw, th, tconf);
  MDCV_CHECK_LAUNCH();
  MDCV_LAUNCH(bt_tcls_kernel, dim3(grid_for((long long)B * T)), dim3(256), 0, st, targets, g, rowinfo, tcls);
  MDCV_CHECK_LAUNCH();
  if (err_out) { e = hipMemcpyAsync(err_out, err, 4, hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return (int)e; }
  return MDCV_OK;
}

}  // extern "C"

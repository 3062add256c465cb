// Detection post-processing on device:
//   conf filter -> xywh->corner -> greedy NMS (top_k) -> IoU matching against labels -> AP / R / P per image.
// Replaces the per-image Python loop validate.py:80-141 + utils/nms.py:4-61 + utils/utils.py:58-119.
//
// Built with -ffp-contract=off: every fp32 op rounds exactly like the reference's torch ops, the keep list,
// best-target indices and correct flags are bit-exact against the oracle.
//
// Layout.  A candidate is a 64-bit key  (ordered(score) << 32) | index .  Descending key order is the
// reference's visiting order: descending score, equal scores by descending index (its ascending stable sort
// walked from the back, nms.py:25-32).  Stage 1 (grid-wide, HBM-bound on the confidence column) compacts the
// candidates above the threshold into a per-image key list; stage 2 is one 256-thread workgroup per image:
// radix-select of the top_k keys -> LDS bitonic sort -> 64-bit suppression-mask matrix built with wave ballots
// -> a single-wave greedy scan over that matrix -> label matching and the AP integral.
#include "common.h"

#define MDCV_NMS_MAX_TOPK 512

namespace {

constexpr int KMAX = MDCV_NMS_MAX_TOPK;
constexpr int KW = KMAX / 64;
constexpr int PB = 256;  // threads per image workgroup

__device__ __forceinline__ unsigned ord32(float f) {  // monotone float -> uint (NaN with sign 0 sorts highest)
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx) { return ((unsigned long long)ord32(score) << 32) | idx; }

// ---------------------------------------------------------------- stage 1
// One thread per prediction row; wave-aggregated slot allocation.  keys[b*N + slot]; cnt[b].
__global__ __launch_bounds__(256) void post_filter_kernel(const float* __restrict__ pred, int B, int N, int row_len, float conf_thres,
                                                          unsigned long long* __restrict__ keys, int* __restrict__ cnt) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  bool pass = false;
  if (n < N) {
    s = pred[((size_t)b * N + n) * row_len + 4];
    pass = s > conf_thres;  // validate.py:81 (NaN does not pass)
  }
  const unsigned long long m = __ballot(pass);
  if (m == 0) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&cnt[b], __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1, 64);
  if (pass) keys[(size_t)b * N + base + __popcll(m & ((1ull << lane) - 1))] = make_key(s, (unsigned)n);
}

// keys for the single-image nms() entry: every box is a candidate.
__global__ __launch_bounds__(256) void nms_keys_kernel(const float* __restrict__ scores, int n, unsigned long long* __restrict__ keys,
                                                       int* __restrict__ cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = make_key(scores[i], (unsigned)i);
  if (i == 0) *cnt = n;
}

// ---------------------------------------------------------------- stage 2
struct PostArgs {
  const unsigned long long* keys;  // [B][N]
  const int* cnt;                  // [B]
  int N, top_k;
  float nms_thres;
  // batched source: prediction rows (cx, cy, w, h, conf, cls...)
  const float* pred; int row_len, C;
  // single-image source: corner boxes [n,4]
  const float* boxes;
  // labels
  const float* targets; int T; float iou_thres, width, height;
  // outputs
  float* out_boxes; float* out_prob; int* out_cls; long long* out_index; unsigned char* out_correct; int* out_count; float* out_stats;
};

struct PostSmem {
  unsigned long long key[KMAX];
  float4 box[KMAX];
  float area[KMAX];
  unsigned long long mask[KMAX][KW];
  unsigned hist[256];
  int keep[KMAX];
  unsigned long long thresh;
  int need, sel, count, ngt, done;
};

// top_k-th largest key of keys[0..cnt) (all keys distinct): MSB-first radix select, 8 bits a pass.
// Returns a threshold t such that exactly `top_k` keys are >= t.
__device__ unsigned long long radix_select(const unsigned long long* __restrict__ keys, int cnt, int top_k, PostSmem& sm) {
  const int tid = threadIdx.x;
  if (tid == 0) { sm.thresh = 0; sm.need = top_k; sm.done = 0; }
  for (int p = 7; p >= 0; --p) {
    sm.hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = sm.thresh;
    const int sh = 8 * p;
    for (int i = tid; i < cnt; i += PB) {
      const unsigned long long k = keys[i];
      const bool match = (p == 7) || ((k >> (sh + 8)) == (prefix >> (sh + 8)));
      if (match) atomicAdd(&sm.hist[(unsigned)(k >> sh) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: suffix scan over 256 bins, 4 per lane, highest bins in the highest lane
      const unsigned h0 = sm.hist[4 * tid], h1 = sm.hist[4 * tid + 1], h2 = sm.hist[4 * tid + 2], h3 = sm.hist[4 * tid + 3];
      const unsigned s = h0 + h1 + h2 + h3;
      unsigned incl = s;  // inclusive suffix sum over lanes >= tid
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_down(incl, o, 64);
        if (tid + o < 64) incl += v;
      }
      const unsigned above = incl - s;
      const unsigned need = (unsigned)sm.need;
      if (above < need && need <= incl) {
        unsigned a = above; int d; unsigned hd;
        if (a + h3 >= need) { d = 3; hd = h3; }
        else { a += h3; if (a + h2 >= need) { d = 2; hd = h2; } else { a += h2; if (a + h1 >= need) { d = 1; hd = h1; } else { a += h1; d = 0; hd = h0; } } }
        sm.thresh = prefix | ((unsigned long long)(4 * tid + d) << sh);
        sm.need = (int)(need - a);
        sm.done = (hd == need - a);  // the whole bin is taken: lower bits of the threshold stay 0
      }
    }
    __syncthreads();
    if (sm.done) break;
  }
  return sm.thresh;
}

// descending bitonic sort of sm.key[0..P), P a power of two <= KMAX
__device__ void bitonic_desc(PostSmem& sm, int P) {
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += PB) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = sm.key[i], b = sm.key[l];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { sm.key[i] = b; sm.key[l] = a; }
        }
      }
      __syncthreads();
    }
}

template <bool BATCHED>
__global__ __launch_bounds__(PB) void post_image_kernel(PostArgs A) {
  __shared__ PostSmem sm;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* keys = A.keys + (size_t)b * A.N;
  int cnt = A.cnt[b];
  cnt = cnt < A.N ? cnt : A.N;
  const int K = cnt < A.top_k ? cnt : A.top_k;

  // ---- candidates -> sm.key[0..K) sorted by descending key
  int P = 64;
  while (P < K) P <<= 1;
  unsigned long long thresh = 0;
  if (cnt > A.top_k) thresh = radix_select(keys, cnt, A.top_k, sm);
  if (tid == 0) sm.sel = 0;
  for (int i = tid; i < P; i += PB) sm.key[i] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < cnt; i0 += PB) {
    const int i = i0 + tid;
    unsigned long long k = 0;
    const bool take = i < cnt && (k = keys[i]) >= thresh;
    const unsigned long long m = __ballot(take);
    int base = 0;
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      if (lane == leader) base = atomicAdd(&sm.sel, __popcll(m));
      base = __shfl(base, leader, 64);
      if (take) sm.key[base + __popcll(m & ((1ull << lane) - 1))] = k;
    }
  }
  __syncthreads();
  bitonic_desc(sm, P);

  // ---- boxes
  for (int t = tid; t < K; t += PB) {
    const unsigned idx = (unsigned)sm.key[t];
    float4 bx;
    if (BATCHED) {
      const float* r = A.pred + ((size_t)b * A.N + idx) * A.row_len;
      const float cx = r[0], cy = r[1], hw = r[2] / 2.f, hh = r[3] / 2.f;  // validate.py:88-91
      bx = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
    } else {
      const float* r = A.boxes + (size_t)idx * 4;
      bx = make_float4(r[0], r[1], r[2], r[3]);
    }
    sm.box[t] = bx;
    sm.area[t] = (bx.z - bx.x) * (bx.w - bx.y);  // nms.py:24
  }
  const int W = (K + 63) >> 6;
  __syncthreads();

  // ---- suppression matrix: bit j of mask[i] set when candidate j (> i) is removed once i is kept
  for (int i = wave; i < K; i += PB / 64) {
    const float4 bi = sm.box[i];
    const float ai = sm.area[i];
    for (int w = i >> 6; w < W; ++w) {
      const int j = w * 64 + lane;
      bool rem = false;
      if (j > i && j < K) {
        const float4 bj = sm.box[j];
        const float xx1 = fmaxf(bj.x, bi.x), yy1 = fmaxf(bj.y, bi.y), xx2 = fminf(bj.z, bi.z), yy2 = fminf(bj.w, bi.w);
        const float ww = fmaxf(xx2 - xx1, 0.f), hh = fmaxf(yy2 - yy1, 0.f);
        const float inter = ww * hh;
        const float uni = (sm.area[j] - inter) + ai;  // nms.py:56
        const float iou = inter / uni;
        rem = !(iou <= A.nms_thres);                  // survivors are IoU.le(overlap); NaN is removed
      }
      const unsigned long long m = __ballot(rem);
      if (lane == 0) sm.mask[i][w] = m;
    }
  }
  __syncthreads();

  // ---- greedy scan, wave 0: lane w owns word w of the removed set
  if (wave == 0) {
    unsigned long long removed = 0;
    int count = 0;
    for (int i = 0; i < K; ++i) {
      const unsigned long long r = __shfl(removed, i >> 6, 64);
      if (!((r >> (i & 63)) & 1ull)) {
        if (lane == 0) sm.keep[count] = i;
        ++count;
        if (lane < W) removed |= sm.mask[i][lane];
      }
    }
    if (lane == 0) sm.count = count;
  }
  __syncthreads();
  const int count = sm.count;

  if (!BATCHED) {
    for (int c = tid; c < count; c += PB) A.out_index[c] = (long long)(unsigned)sm.key[sm.keep[c]];
    if (tid == 0) A.out_count[0] = count;
    return;
  }

  // ---- kept detections out (already in descending-confidence order, validate.py:100-104)
  const size_t ob = (size_t)b * A.top_k;
  for (int c = tid; c < count; c += PB) {
    const int t = sm.keep[c];
    const unsigned idx = (unsigned)sm.key[t];
    const float* r = A.pred + ((size_t)b * A.N + idx) * A.row_len;
    int best = 0;
    if (A.C > 0) {
      float bv = r[5];
      for (int k = 1; k < A.C; ++k) { const float v = r[5 + k]; if (v > bv) { bv = v; best = k; } }  // first maximum
    }
    const float4 bx = sm.box[t];
    *(float4*)(A.out_boxes + (ob + c) * 4) = bx;
    A.out_prob[ob + c] = r[4];
    A.out_cls[ob + c] = best;
    A.out_index[ob + c] = idx;
  }
  if (tid == 0) A.out_count[b] = count;
  if (!A.targets) {
    for (int c = tid; c < count; c += PB) A.out_correct[ob + c] = 0;
    if (tid == 0) { float4 z = make_float4(0.f, 0.f, 0.f, 0.f); *(float4*)(A.out_stats + (size_t)b * 4) = z; }
    return;
  }

  // ---- labels: rows with all of (cx,cy,w,h) > 0 are real (validate.py:105)
  float* best_iou = sm.area;             // reuse: [KMAX]
  int* best_t = (int*)&sm.mask[0][0];    // reuse: [KMAX] ints, then [KMAX] ok flags, recall, precision
  int* okf = best_t + KMAX;
  float* rec = (float*)(okf + KMAX);
  float* pre = rec + KMAX + 2;
  const float* tg = A.targets + (size_t)b * A.T * 5;
  if (tid == 0) sm.ngt = 0;
  __syncthreads();
  {
    int local = 0;
    for (int t = tid; t < A.T; t += PB) {
      const float* l = tg + t * 5;
      local += (l[1] > 0.f && l[2] > 0.f && l[3] > 0.f && l[4] > 0.f) ? 1 : 0;
    }
    if (local) atomicAdd(&sm.ngt, local);
  }
  __syncthreads();
  const int ngt = sm.ngt;
  if (count == 0 || ngt == 0) {  // validate.py:97 / :120 `continue`
    for (int c = tid; c < count; c += PB) A.out_correct[ob + c] = 0;
    if (tid == 0) { float4 z = make_float4(0.f, 0.f, 0.f, 0.f); *(float4*)(A.out_stats + (size_t)b * 4) = z; }
    return;
  }
  for (int c = tid; c < count; c += PB) {
    const float4 d = sm.box[sm.keep[c]];
    const float da = (d.z - d.x + 1.f) * (d.w - d.y + 1.f);
    float bv = -1.f; int bt = -1;
    for (int t = 0; t < A.T; ++t) {
      const float* l = tg + t * 5;
      const float lx = l[1], ly = l[2], lw = l[3], lh = l[4];
      if (!(lx > 0.f && ly > 0.f && lw > 0.f && lh > 0.f)) continue;
      const float tx1 = (lx - lw / 2.f) * A.width, ty1 = (ly - lh / 2.f) * A.height;   // utils.py:121-127, validate.py:107-109
      const float tx2 = (lx + lw / 2.f) * A.width, ty2 = (ly + lh / 2.f) * A.height;
      const float ix1 = fmaxf(d.x, tx1), iy1 = fmaxf(d.y, ty1), ix2 = fminf(d.z, tx2), iy2 = fminf(d.w, ty2);
      const float inter = fmaxf(ix2 - ix1 + 1.f, 0.f) * fmaxf(iy2 - iy1 + 1.f, 0.f);       // utils.py:184-186
      const float ta = (tx2 - tx1 + 1.f) * (ty2 - ty1 + 1.f);
      const float iou = inter / (da + ta - inter + 1e-12f);
      if (bt < 0 || iou > bv) { bv = iou; bt = t; }                                        // first maximum
    }
    best_iou[c] = bv; best_t[c] = bt; okf[c] = bv > A.iou_thres ? 1 : 0;
  }
  __syncthreads();
  // correct[c]: above the IoU threshold and no earlier such detection claimed the same label (validate.py:127-131)
  for (int c = tid; c < count; c += PB) {
    int ok = okf[c];
    const int bt = best_t[c];
    for (int e = 0; ok && e < c; ++e) ok = !(okf[e] && best_t[e] == bt);
    A.out_correct[ob + c] = (unsigned char)ok;
    best_iou[c] = ok ? 1.f : 0.f;   // tp as float for the integral below
  }
  __syncthreads();
  if (tid == 0) {  // utils.py:58-119, float32, in order
    const float fn = (float)ngt;
    float tpc = 0.f, fpc = 0.f;
    for (int c = 0; c < count; ++c) {
      const float tp = best_iou[c];
      tpc += tp; fpc += 1.f - tp;
      rec[c + 1] = tpc / fn;
      pre[c + 1] = tpc / (tpc + fpc);
    }
    rec[0] = 0.f; pre[0] = 0.f; rec[count + 1] = 1.f; pre[count + 1] = 0.f;
    for (int i = count + 1; i > 0; --i) pre[i - 1] = fmaxf(pre[i - 1], pre[i]);
    float ap = 0.f;
    for (int j = 0; j <= count; ++j)
      if (rec[j + 1] != rec[j]) ap += (rec[j + 1] - rec[j]) * pre[j + 1];
    float4 o = make_float4(ap, tpc / fn, tpc / (tpc + fpc), 1.f);
    *(float4*)(A.out_stats + (size_t)b * 4) = o;
  }
}

// average_precision() on its own (utils.py:58-88): m <= KMAX, stable sort by descending confidence.
__global__ __launch_bounds__(PB) void average_precision_kernel(const unsigned char* __restrict__ tp, const float* __restrict__ conf, int m,
                                                               int n_gt, float* __restrict__ out3) {
  __shared__ PostSmem sm;
  const int tid = threadIdx.x;
  int P = 64;
  while (P < m) P <<= 1;
  for (int i = tid; i < P; i += PB) sm.key[i] = i < m ? (((unsigned long long)ord32(conf[i]) << 32) | (unsigned)(~i)) : 0ull;  // ties: lower index first
  __syncthreads();
  bitonic_desc(sm, P);
  float* rec = (float*)&sm.mask[0][0];
  float* pre = rec + KMAX + 2;
  if (tid == 0) {
    const float fn = (float)n_gt;
    float tpc = 0.f, fpc = 0.f;
    for (int c = 0; c < m; ++c) {
      const float t = tp[~(unsigned)sm.key[c]] ? 1.f : 0.f;
      tpc += t; fpc += 1.f - t;
      rec[c + 1] = tpc / fn;
      pre[c + 1] = tpc / (tpc + fpc);
    }
    rec[0] = 0.f; pre[0] = 0.f; rec[m + 1] = 1.f; pre[m + 1] = 0.f;
    for (int i = m + 1; i > 0; --i) pre[i - 1] = fmaxf(pre[i - 1], pre[i]);
    float ap = 0.f;
    for (int j = 0; j <= m; ++j)
      if (rec[j + 1] != rec[j]) ap += (rec[j + 1] - rec[j]) * pre[j + 1];
    out3[0] = ap; out3[1] = tpc / fn; out3[2] = tpc / (tpc + fpc);
  }
}

}  // namespace

extern "C" {

// workspace: keys u64[B*N] | cnt int[B] (+pad)
long long mdcv_detect_post_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (long long)B * N * 8 + (((long long)B * 4 + 63) & ~63LL);
}

int mdcv_detect_post(const float* pred, int B, int N, int C, const float* targets, int T, float conf_thres, float nms_thres,
                     float iou_thres, float width, float height, int top_k, float* out_boxes, float* out_prob, int* out_cls,
                     long long* out_index, unsigned char* out_correct, int* out_count, float* out_stats, void* workspace,
                     void* stream) {
  if (!pred || !out_boxes || !out_prob || !out_cls || !out_index || !out_correct || !out_count || !out_stats || !workspace)
    return MDCV_EARG;
  if (B <= 0 || N <= 0 || C < 0 || T < 0 || top_k <= 0 || top_k > KMAX || (targets == nullptr && T != 0)) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)workspace;
  int* cnt = (int*)(keys + (size_t)B * N);
  hipError_t e = hipMemsetAsync(cnt, 0, (size_t)B * 4, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(post_filter_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, pred, B, N, 5 + C, conf_thres, keys, cnt);
  MDCV_CHECK_LAUNCH();
  PostArgs a{};
  a.keys = keys; a.cnt = cnt; a.N = N; a.top_k = top_k; a.nms_thres = nms_thres;
  a.pred = pred; a.row_len = 5 + C; a.C = C; a.boxes = nullptr;
  a.targets = T > 0 ? targets : nullptr; a.T = T; a.iou_thres = iou_thres; a.width = width; a.height = height;
  a.out_boxes = out_boxes; a.out_prob = out_prob; a.out_cls = out_cls; a.out_index = out_index; a.out_correct = out_correct;
  a.out_count = out_count; a.out_stats = out_stats;
  hipLaunchKernelGGL(post_image_kernel<true>, dim3(B), dim3(PB), 0, st, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

long long mdcv_nms_workspace_bytes(int n) { return n <= 0 ? 64 : (long long)n * 8 + 64; }

int mdcv_nms(const float* boxes, const float* scores, int n, float overlap, int top_k, long long* keep, int* count, void* workspace,
             void* stream) {
  if (n < 0 || top_k <= 0 || top_k > KMAX || !keep || !count || !workspace || (n > 0 && (!boxes || !scores))) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {  // nms.py:17-18
    hipError_t e = hipMemsetAsync(count, 0, 4, st);
    return e == hipSuccess ? MDCV_OK : (int)e;
  }
  unsigned long long* keys = (unsigned long long*)workspace;
  int* cnt = (int*)(keys + n);
  hipLaunchKernelGGL(nms_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, scores, n, keys, cnt);
  MDCV_CHECK_LAUNCH();
  PostArgs a{};
  a.keys = keys; a.cnt = cnt; a.N = n; a.top_k = top_k; a.nms_thres = overlap; a.boxes = boxes;
  a.out_index = keep; a.out_count = count;
  hipLaunchKernelGGL(post_image_kernel<false>, dim3(1), dim3(PB), 0, st, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_average_precision(const unsigned char* tp, const float* conf, int m, int n_gt, float* out3, void* stream) {
  if (!tp || !conf || !out3 || m <= 0 || m > KMAX) return MDCV_EARG;
  hipLaunchKernelGGL(average_precision_kernel, dim3(1), dim3(PB), 0, (hipStream_t)stream, tp, conf, m, n_gt, out3);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

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
// candidates above the threshold into a per-image key list (one global atomic per 1024 rows, counters on separate
// cache lines); stage 2 is one 1024-thread workgroup per image (16 waves so that the LDS / divide latencies overlap):
// radix-select of the top_k keys -> LDS rank sort -> 64-bit suppression-mask matrix built with wave ballots
// -> a single-wave greedy scan that resolves each 64-candidate word in registers (v_readlane, no LDS round trip per
// candidate) -> label matching and the AP integral (parallel prefix / envelope, ordered float32 sum).
#include "common.h"

#define MDCV_NMS_MAX_TOPK 512
#ifdef MDCV_POST_TS   /* phase timestamps of workgroup 0 (scripts/post_phases.sh); never defined in the shipped build */
#include <cstdio>
__device__ long long g_post_ts[16];
#define TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_post_ts[i] = (long long)wall_clock64(); } while (0)
#else
#define TS(i)
#endif

namespace {

constexpr int KMAX = MDCV_NMS_MAX_TOPK;
constexpr int KW = KMAX / 64;
constexpr int PB = 1024;  // threads per image workgroup
static_assert(PB == 2 * KMAX, "rank_sort_desc uses two threads per key");
constexpr int CNT_STRIDE = 32;  // ints between per-image counters: one 128-byte line each
constexpr int FILTER_ROWS = 1024;  // prediction rows per stage-1 workgroup

__device__ __forceinline__ unsigned ord32(float f) {  // monotone float -> uint (NaN with sign 0 sorts highest)
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx) { return ((unsigned long long)ord32(score) << 32) | idx; }

// ---------------------------------------------------------------- stage 1
// 256 threads x 4 rows; slots are allocated wave -> workgroup (LDS) -> one global atomic per workgroup, so the
// per-image counters (each on its own 128-byte line) see N/1024 atomics instead of one per wave.
__global__ __launch_bounds__(256) void post_filter_kernel(const float* __restrict__ pred, int B, int N, int row_len, float conf_thres,
                                                          unsigned long long* __restrict__ keys, int* __restrict__ cnt) {
  __shared__ int wave_cnt[4][4];
  __shared__ int block_base;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * FILTER_ROWS + tid;
  float s[4]; bool pass[4]; unsigned long long m[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + r * 256;
    s[r] = n < N ? pred[((size_t)b * N + n) * row_len + 4] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pass[r] = (n0 + r * 256 < N) && (s[r] > conf_thres);  // validate.py:81 (NaN does not pass)
    m[r] = __ballot(pass[r]);
    if (lane == 0) wave_cnt[r][wave] = __popcll(m[r]);
  }
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int w = 0; w < 4; ++w) { const int c = wave_cnt[r][w]; wave_cnt[r][w] = tot; tot += c; }
    block_base = tot ? atomicAdd(&cnt[b * CNT_STRIDE], tot) : 0;
  }
  __syncthreads();
  const int base = block_base;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (pass[r])
      keys[(size_t)b * N + base + wave_cnt[r][wave] + __popcll(m[r] & ((1ull << lane) - 1))] = make_key(s[r], (unsigned)(n0 + r * 256));
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
  int wsum[PB / 64];
};

// top_k-th largest key of keys[0..cnt) (all keys distinct): MSB-first radix select, 8 bits a pass.
// Returns a threshold t such that exactly `top_k` keys are >= t.
__device__ unsigned long long radix_select(const unsigned long long* __restrict__ keys, int cnt, int top_k, PostSmem& sm) {
  const int tid = threadIdx.x;
  if (tid == 0) { sm.thresh = 0; sm.need = top_k; sm.done = 0; }
  for (int p = 7; p >= 0; --p) {
    if (tid < 256) sm.hist[tid] = 0;
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

// AP / recall / precision (utils.py:58-119) of `count` (1..KMAX) detections already in confidence order; tp01[c] is 1.f
// for a true positive.  Called by every thread of the workgroup; the result is valid on thread 0.
// The cumulative counts are integers (prefix popcount), the precision envelope is a suffix maximum (exact in any
// order); only the final sum is order-sensitive and is accumulated in float32 in index order like the oracle.
__device__ float4 ap_integral(PostSmem& sm, const float* tp01, int count, int ngt, float* rec, float* pre, float* term) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool t = tid < count && tp01[tid] != 0.f;
  const unsigned long long m = __ballot(t);
  if (lane == 0) sm.wsum[wave] = __popcll(m);
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < PB / 64; ++w) { const int v = sm.wsum[w]; total += v; before += w < wave ? v : 0; }
  const int tpi = before + __popcll(m & ((2ull << lane) - 1ull));
  const float fn = (float)ngt;
  if (tid < count) {
    const float tpc = (float)tpi, fpc = (float)(tid + 1 - tpi);
    rec[tid + 1] = tpc / fn;                 // utils.py:77
    pre[tid + 1] = tpc / (tpc + fpc);        // utils.py:81
  }
  if (tid == 0) { rec[0] = 0.f; pre[0] = 0.f; rec[count + 1] = 1.f; pre[count + 1] = 0.f; }
  __syncthreads();
  if (tid < KMAX + 64) {                     // term j = (mrec[j+1] - mrec[j]) * envelope[j+1], zero past the curve
    float v = 0.f;
    if (tid <= count && rec[tid + 1] != rec[tid]) {
      float env = 0.f;
      for (int i = tid + 1; i <= count + 1; ++i) env = fmaxf(env, pre[i]);
      v = (rec[tid + 1] - rec[tid]) * env;
    }
    term[tid] = v;
  }
  __syncthreads();
  float ap = 0.f;
  if (wave == 0) {
    for (int ch = 0; ch * 64 <= count; ++ch) {
      const int v = __float_as_int(term[ch * 64 + lane]);
#pragma unroll
      for (int k = 0; k < 64; ++k) ap += __int_as_float(__builtin_amdgcn_readlane(v, k));
    }
  }
  const float tpc = (float)total, fpc = (float)(count - total);
  return make_float4(ap, tpc / fn, tpc / (tpc + fpc), 1.f);
}

// descending sort of the distinct keys sm.key[0..K), K <= KMAX: rank = number of larger keys.  Two threads per key,
// each counting over half of the others (the inner read is a wave-wide LDS broadcast).
__device__ void rank_sort_desc(PostSmem& sm, int K) {
  const int i = threadIdx.x & (KMAX - 1), h = threadIdx.x / KMAX;  // PB == 2 * KMAX
  const int half = (K + 1) >> 1;
  unsigned long long mine = 0;
  int rank = 0;
  if (i < K) {
    mine = sm.key[i];
    const int j0 = h * half, j1 = (j0 + half) < K ? (j0 + half) : K;
#pragma unroll 8
    for (int j = j0; j < j1; ++j) rank += sm.key[j] > mine ? 1 : 0;
  }
  if (h == 1 && i < K) sm.keep[i] = rank;
  __syncthreads();
  if (h == 0 && i < K) rank += sm.keep[i];
  __syncthreads();
  if (h == 0 && i < K) sm.key[rank] = mine;
  __syncthreads();
}

template <bool BATCHED>
__global__ __launch_bounds__(PB) void post_image_kernel(PostArgs A) {
  __shared__ PostSmem sm;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* keys = A.keys + (size_t)b * A.N;
  int cnt = A.cnt[b * CNT_STRIDE];
  cnt = cnt < A.N ? cnt : A.N;
  const int K = cnt < A.top_k ? cnt : A.top_k;

  TS(0);
  // ---- candidates -> sm.key[0..K) sorted by descending key
  unsigned long long thresh = 0;
  if (cnt > A.top_k) thresh = radix_select(keys, cnt, A.top_k, sm);
  TS(1);
  if (tid == 0) sm.sel = 0;
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
  TS(2);
  rank_sort_desc(sm, K);

  TS(3);
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

  TS(4);
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

  TS(5);
  // ---- greedy scan, wave 0: lane w owns word w of the removed set.  Each 64-candidate word is resolved in
  // registers: lane l holds row (64w + l)'s bits for this word, the walk over the 64 bits is scalar (v_readlane).
  if (wave == 0) {
    unsigned long long removed = 0;
    int count = 0;
    const int Ku = __builtin_amdgcn_readfirstlane(K);
    for (int w = 0; w < W; ++w) {
      const int il = w * 64 + lane;
      const unsigned long long diag = il < Ku ? sm.mask[il][w] : 0ull;
      const int dlo = (int)(unsigned)diag, dhi = (int)(unsigned)(diag >> 32);
      unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(removed >> 32), w) << 32) |
                               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)removed, w);
      const int nb = Ku - w * 64;
      if (nb < 64) cur |= ~0ull << nb;  // past the last candidate
      unsigned long long kept = 0;
#pragma unroll
      for (int t = 0; t < 64; ++t) {
        const unsigned long long row = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, t) << 32) | (unsigned)__builtin_amdgcn_readlane(dlo, t);
        const unsigned long long alive = ((cur >> t) & 1ull) - 1ull;  // all ones while candidate t has not been removed
        kept |= (1ull << t) & alive;
        cur |= row & alive;
      }
      if ((kept >> lane) & 1ull) sm.keep[count + __popcll(kept & ((1ull << lane) - 1))] = il;
      count += __popcll(kept);
      if (w + 1 < W && lane > w && lane < W) {  // rows of the kept candidates -> later words; all 64 reads in flight at once
#pragma unroll
        for (int t = 0; t < 64; ++t) removed |= sm.mask[w * 64 + t][lane] & (0ull - ((kept >> t) & 1ull));
      }
    }
    if (lane == 0) sm.count = count;
  }
  __syncthreads();
  TS(6);
  const int count = sm.count;

  if (!BATCHED) {
    for (int c = tid; c < count; c += PB) A.out_index[c] = (long long)(unsigned)sm.key[sm.keep[c]];
    if (tid == 0) A.out_count[0] = count;
    return;
  }

  // ---- kept detections out (already in descending-confidence order, validate.py:100-104)
  const size_t ob = (size_t)b * A.top_k;
  for (int c = tid >> 4; c < count; c += PB / 16) {  // 16 lanes per kept row: coalesced class scores, first maximum
    const int q = tid & 15;
    const int t = sm.keep[c];
    const unsigned idx = (unsigned)sm.key[t];
    const float* r = A.pred + ((size_t)b * A.N + idx) * A.row_len;
    float bv = -__builtin_inff(); int bk = 0x7fffffff;
    for (int k = q; k < A.C; k += 16) { const float v = r[5 + k]; if (v > bv || bk == 0x7fffffff) { bv = v; bk = k; } }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int ok = __shfl_xor(bk, o, 64);
      if (ok != 0x7fffffff && (bk == 0x7fffffff || ov > bv || (ov == bv && ok < bk))) { bv = ov; bk = ok; }
    }
    if (q == 0) {
      const float4 bx = sm.box[t];
      *(float4*)(A.out_boxes + (ob + c) * 4) = bx;
      A.out_prob[ob + c] = r[4];
      A.out_cls[ob + c] = A.C > 0 ? bk : 0;
      A.out_index[ob + c] = idx;
    }
  }
  TS(7);
  for (int c = count + tid; c < A.top_k; c += PB) {  // rows past the kept detections read as zeros
    *(float4*)(A.out_boxes + (ob + c) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    A.out_prob[ob + c] = 0.f; A.out_cls[ob + c] = 0; A.out_index[ob + c] = 0; A.out_correct[ob + c] = 0;
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
  float* term = pre + KMAX + 2;
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
  TS(8);
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
  TS(9);
  // correct[c]: above the IoU threshold and no earlier such detection claimed the same label (validate.py:127-131)
  for (int c = tid; c < count; c += PB) {
    int ok = okf[c];
    const int bt = best_t[c];
    for (int e = 0; ok && e < c; ++e) ok = !(okf[e] && best_t[e] == bt);
    A.out_correct[ob + c] = (unsigned char)ok;
    best_iou[c] = ok ? 1.f : 0.f;   // tp as float for the integral below
  }
  __syncthreads();
  TS(10);
  const float4 o = ap_integral(sm, best_iou, count, ngt, rec, pre, term);
  if (tid == 0) *(float4*)(A.out_stats + (size_t)b * 4) = o;
  TS(11);
}

// average_precision() on its own (utils.py:58-88): m <= KMAX, stable sort by descending confidence.
__global__ __launch_bounds__(PB) void average_precision_kernel(const unsigned char* __restrict__ tp, const float* __restrict__ conf, int m,
                                                               int n_gt, float* __restrict__ out3) {
  __shared__ PostSmem sm;
  const int tid = threadIdx.x;
  if (tid < m) sm.key[tid] = ((unsigned long long)ord32(conf[tid]) << 32) | (unsigned)(~tid);  // ties: lower index first
  __syncthreads();
  rank_sort_desc(sm, m);
  float* rec = (float*)&sm.mask[0][0];
  float* pre = rec + KMAX + 2;
  float* term = pre + KMAX + 2;
  if (tid < m) sm.area[tid] = tp[~(unsigned)sm.key[tid]] ? 1.f : 0.f;
  __syncthreads();
  const float4 o = ap_integral(sm, sm.area, m, n_gt, rec, pre, term);
  if (tid == 0) { out3[0] = o.x; out3[1] = o.y; out3[2] = o.z; }
}

}  // namespace

extern "C" {

// workspace: keys u64[B*N] | cnt int[B] (+pad)
long long mdcv_detect_post_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (long long)B * N * 8 + (long long)B * CNT_STRIDE * 4;
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
  hipError_t e = hipMemsetAsync(cnt, 0, (size_t)B * CNT_STRIDE * 4, st);
  if (e != hipSuccess) return (int)e;
  MDCV_LAUNCH(post_filter_kernel, dim3((N + FILTER_ROWS - 1) / FILTER_ROWS, B), dim3(256), 0, st, pred, B, N, 5 + C, conf_thres, keys, cnt);
  MDCV_CHECK_LAUNCH();
  PostArgs a{};
  a.keys = keys; a.cnt = cnt; a.N = N; a.top_k = top_k; a.nms_thres = nms_thres;
  a.pred = pred; a.row_len = 5 + C; a.C = C; a.boxes = nullptr;
  a.targets = T > 0 ? targets : nullptr; a.T = T; a.iou_thres = iou_thres; a.width = width; a.height = height;
  a.out_boxes = out_boxes; a.out_prob = out_prob; a.out_cls = out_cls; a.out_index = out_index; a.out_correct = out_correct;
  a.out_count = out_count; a.out_stats = out_stats;
  MDCV_LAUNCH(post_image_kernel<true>, dim3(B), dim3(PB), 0, st, a);
  MDCV_CHECK_LAUNCH();
#ifdef MDCV_POST_TS
  {
    static int calls = 0;
    long long h[16];
    if (++calls == 50 && hipStreamSynchronize(st) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_post_ts), sizeof(h)) == hipSuccess)
      for (int i = 1; i < 12; ++i) printf("post phase %2d: %7.2f us\n", i, (h[i] - h[i - 1]) / 100.0);
  }
#endif
  return MDCV_OK;
}

long long mdcv_nms_workspace_bytes(int n) { return n <= 0 ? 128 : (long long)n * 8 + 128; }

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
  MDCV_LAUNCH(nms_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, scores, n, keys, cnt);
  MDCV_CHECK_LAUNCH();
  PostArgs a{};
  a.keys = keys; a.cnt = cnt; a.N = n; a.top_k = top_k; a.nms_thres = overlap; a.boxes = boxes;
  a.out_index = keep; a.out_count = count;
  MDCV_LAUNCH(post_image_kernel<false>, dim3(1), dim3(PB), 0, st, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_average_precision(const unsigned char* tp, const float* conf, int m, int n_gt, float* out3, void* stream) {
  if (!tp || !conf || !out3 || m <= 0 || m > KMAX) return MDCV_EARG;
  MDCV_LAUNCH(average_precision_kernel, dim3(1), dim3(PB), 0, (hipStream_t)stream, tp, conf, m, n_gt, out3);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

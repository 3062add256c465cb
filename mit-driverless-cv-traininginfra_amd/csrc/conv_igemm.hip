// NHWC implicit-GEMM convolution for gfx950 (MFMA): forward, data-gradient and weight-gradient.
//
// Replaces the nn.Conv2d calls on the reference hot path (CVC-YOLOv3/models.py:59-65,
// RektNet/keypoint_net.py:17,25, RektNet/resnet.py:12-19) and their autograd backward.
//
// GEMM view (fwd):   Y[m, n] = sum_k  Xcol[m, k] * W[n, k]      m = (img, ho, wo)   n = cout   k = (kh, kw, ci)
//      (dgrad):      dX[m, n] = sum_k dYcol[m, k] * Wt[n, k]    m = (img, hi, wi)   n = cin    k = (kh, kw, co)
//      (wgrad):      dW[co, k] = sum_m dY[m, co] * Xcol[m, k]   reduction over pixels, split over the grid
//
// Activations are NHWC with an explicit channel stride (ldc) so route-concat is a strided write, channels padded
// to a multiple of 8 (pad lanes are exactly zero).  Element type T is bf16 (production: v_mfma_f32_16x16x32_bf16)
// or fp32 (parity mode: v_mfma_f32_16x16x4_f32, bit-exact fmaf chains).  Accumulation is always fp32.
//
// Tiling: 256 threads = 4 waves; K tile = 64 bytes per row (32 bf16 / 16 fp32); global->register->LDS staging with
// the next tile's loads issued before the current tile's MFMAs (one barrier per K tile, 2 LDS buffers);
// LDS rows padded to 80 bytes; epilogue staged through LDS for 16-byte coalesced stores; BatchNorm batch statistics
// (sum, sum of squares per output channel) are produced from the fp32 accumulators in the epilogue.
#include "common.h"
#include "conv_shift.h"
#include "bn_fuse.h"
#include "wgrad_shift.h"
#include "wgrad_stream.h"

// The file is compiled four times (Makefile: -DMDCV_CONV_PART=0..3) so that its template instantiations build in parallel:
//   0  host entry points, weight-gradient and pack kernels, tuning globals      1  bf16 forward      2  bf16 data gradients      3  fp32 (parity mode)
#ifndef MDCV_CONV_PART
#define MDCV_CONV_PART 0
#endif

struct ConvArgs {
  const void* in; const void* w; void* out; const float* bias; const void* addsrc; float* stats;
  int in_ldc, out_ldc, add_ldc;
  int Hin, Win, Cin, Hout, Wout, Nout;
  int KH, KW, stride, pad, dil;
  int M, Ktot, tiles_n, sshift, tiles_total, xcd_chunk;
  int ph, pw, Hs, Ws, kh0, kw0, nkh, nkw;     // MODE 2 (stride-2 data gradient, one output-parity class per launch)
  int cls_split;                              // ALLCLS: 1 = two workgroups per tile, the 4-tap class and the 1+2+2-tap classes (sparse grids)
  BnFuseArgs fuse;                            // BatchNorm-backward sums folded into the store loop of a data gradient (fuse.y == NULL: off)
  EpiArgs epi;                                // inference epilogue act(acc * oscale + bias) (oscale == NULL and act == 0: off)
  XAccArgs xacc;                              // forward statistics added to exact accumulators instead of written as rows (exact_acc.h; acc == NULL: off)
};

#if MDCV_CONV_PART == 0
#else
#endif
// the per-part dispatch entry points (each defined by exactly one part)
int mdcv_cd_bf16_fwd(const ConvArgs& a, hipStream_t st, int B);
int mdcv_cd_bf16_dgrad(const ConvArgs& a, hipStream_t st, int B);
int mdcv_cd_bf16_s2(const ConvArgs& a, hipStream_t st, int B);
int mdcv_cd_bf16_s2_all(const ConvArgs& a, hipStream_t st, int B);
int mdcv_cd_f32_fwd(const ConvArgs& a, hipStream_t st, int B);
int mdcv_cd_f32_dgrad(const ConvArgs& a, hipStream_t st, int B);
int mdcv_cd_f32_s2(const ConvArgs& a, hipStream_t st, int B);

namespace {


// One K tile of MFMAs for a wave: FM x FN fragments of 16x16, KT k-steps of 64 bytes per LDS row (row pitch RB bytes).
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  template <int FM, int FN, int KT, int RB>
  __device__ static __forceinline__ void mma(const unsigned char* sa, const unsigned char* sb, int lane, f32x4_t (&acc)[FM][FN]) {
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      bf16x8_t a[FM], b[FN];
      const int off = (lane & 15) * RB + ks * 64 + (lane >> 4) * 16;
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 16 * RB + off);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sb + j * 16 * RB + off);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};
template <> struct Frag<float> {
  template <int FM, int FN, int KT, int RB>
  __device__ static __forceinline__ void mma(const unsigned char* sa, const unsigned char* sb, int lane, f32x4_t (&acc)[FM][FN]) {
#pragma unroll
    for (int ks = 0; ks < 4 * KT; ++ks) {
      float a[FM], b[FN];
      const int off = (lane & 15) * RB + (ks * 4 + (lane >> 4)) * 4;
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const float*>(sa + i * 16 * RB + off);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const float*>(sb + j * 16 * RB + off);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};

// MODE 0: forward gather  hi = ho*stride - pad + kh*dil
// MODE 1: data gradient   hi = (h + pad - kh*dil) / stride when divisible   (stride is 1 or 2)
// Block = WM x WN waves, tile BM x BN, K tile = KT*64 bytes per row.
template <typename T, int MODE, int BM, int BN, int WM, int WN, int KT>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(ConvArgs a) {
  constexpr int NT = WM * WN * 64;
  constexpr int VEC = ET<T>::VEC;
  constexpr int VPR = 4 * KT;                 // 16-byte vectors per LDS row
  constexpr int BK = VPR * VEC;
  constexpr int RB = 64 * KT + 16;            // LDS row pitch (bytes): +16 keeps ds_read_b128 fragments conflict-light
  constexpr int RPP = NT / VPR;               // tile rows staged per pass
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int NPA = BM / RPP, NPB = (BN + RPP - 1) / RPP;
  constexpr int PIPE = 2 * (BM + BN) * RB;
  constexpr int SROW = BN * (int)sizeof(T) + 16;
  constexpr int STAGE = BM * SROW;
  constexpr int STAT_OFF = PIPE > STAGE ? PIPE : STAGE;
  static_assert(BM % RPP == 0, "tile rows must be a multiple of the staging pass");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware tile order: block b runs on XCD b%8 (observed dispatch); give every XCD one contiguous run of tiles so
  // that the tile_n variants of a row panel and its halo neighbours share that XCD's L2.  Pure speed, not correctness.
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.tiles_total) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile_m = logical / a.tiles_n, tile_n = logical % a.tiles_n;
  const int arow = tid / VPR, kv = tid % VPR;
  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);

  // per-row pixel decomposition (fixed for the whole K loop)
  int bh[NPA], bw[NPA], ib[NPA];
  bool rv[NPA];
  const int HWo = a.Hout * a.Wout;
#pragma unroll
  for (int p = 0; p < NPA; ++p) {
    const int m = tile_m * BM + arow + p * RPP;
    rv[p] = m < a.M;
    const int mm = rv[p] ? m : 0;
    const int img = mm / HWo, rem = mm - img * HWo;
    const int ho = rem / a.Wout, wo = rem - ho * a.Wout;
    ib[p] = img * a.Hin * a.Win;
    if (MODE == 0) { bh[p] = ho * a.stride - a.pad; bw[p] = wo * a.stride - a.pad; }
    else           { bh[p] = ho + a.pad;            bw[p] = wo + a.pad; }
  }
  // per-thread K cursor: k = kt*BK + kv*VEC  ->  (kh, kw, c)
  int kc, kh, kw;
  {
    const int k0 = kv * VEC, tap = k0 / a.Cin;
    kc = k0 - tap * a.Cin; kh = tap / a.KW; kw = tap - kh * a.KW;
  }
  const int smask = a.stride - 1;

  uint4 ra[NPA], rb[NPB];
  auto load_tile = [&](int kt) {
    const bool kvalid = kh < a.KH;
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
      int hi, wi; bool ok = rv[p] && kvalid;
      if (MODE == 0) { hi = bh[p] + kh * a.dil; wi = bw[p] + kw * a.dil; }
      else {
        const int th = bh[p] - kh * a.dil, tw = bw[p] - kw * a.dil;
        ok = ok && th >= 0 && tw >= 0 && (((th | tw) & smask) == 0);
        hi = th >> a.sshift; wi = tw >> a.sshift;
      }
      ok = ok && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = *reinterpret_cast<const uint4*>(in + ((size_t)(ib[p] + hi * a.Win + wi) * a.in_ldc + kc));
      ra[p] = v;
    }
    const int k = kt * BK + kv * VEC;
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int brow = arow + p * RPP;
      const int n = tile_n * BN + brow;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (brow < BN && n < a.Nout && k < a.Ktot) v = *reinterpret_cast<const uint4*>(w + ((size_t)n * a.Ktot + k));
      rb[p] = v;
    }
  };
  auto advance = [&]() {
    kc += BK;
    while (kc >= a.Cin) { kc -= a.Cin; if (++kw == a.KW) { kw = 0; ++kh; } }
  };
  auto store_tile = [&](int buf) {
    unsigned char* sA = smem + buf * (BM + BN) * RB;
    unsigned char* sB = sA + BM * RB;
#pragma unroll
    for (int p = 0; p < NPA; ++p) *reinterpret_cast<uint4*>(sA + (arow + p * RPP) * RB + kv * 16) = ra[p];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int brow = arow + p * RPP;
      if (brow < BN) *reinterpret_cast<uint4*>(sB + brow * RB + kv * 16) = rb[p];
    }
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (a.Ktot + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) { advance(); load_tile(kt + 1); }
    const unsigned char* sA = smem + cur * (BM + BN) * RB + wm * TM * RB;
    const unsigned char* sB = smem + cur * (BM + BN) * RB + BM * RB + wn * TN * RB;
    Frag<T>::template mma<FM, FN, KT, RB>(sA, sB, lane, acc);
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int n0 = tile_n * BN + wn * TN, m0 = tile_m * BM + wm * TM;
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + j * 16 + (lane & 15);
      const float bv = n < a.Nout ? a.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += bv;
    }
  }
  float* sstat = reinterpret_cast<float*>(smem + STAT_OFF);   // [WM][2][BN]
  const bool want_stats = a.stats || a.xacc.acc;
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
          const float v = m < a.M ? acc[i][j][r] : 0.f;
          s += v; q += v * v;
        }
      s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) {
        sstat[(wm * 2 + 0) * BN + wn * TN + j * 16 + lane] = s;
        sstat[(wm * 2 + 1) * BN + wn * TN + j * 16 + lane] = q;
      }
    }
  }
  // stage the tile as T (the K loop ended with a barrier, so the pipeline buffers are free)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * TM + i * 16 + (lane >> 4) * 4 + r, col = wn * TN + j * 16 + (lane & 15);
        ET<T>::st(reinterpret_cast<T*>(smem + row * SROW) + col, acc[i][j][r]);
      }
  __syncthreads();
  // statistics rows are per 128 pixels regardless of the tile height (one row per group of wave-rows)
  constexpr int G = BM / 128, WPG = WM / G;
  if (want_stats && tid < BN * G) {
    const int g = tid / BN, col = tid - g * BN;
    const int n = tile_n * BN + col, srow = tile_m * G + g;
    if (n < a.Nout && srow * 128 < a.M) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < WPG; ++r) { s += sstat[((g * WPG + r) * 2 + 0) * BN + col]; q += sstat[((g * WPG + r) * 2 + 1) * BN + col]; }
      if (a.xacc.acc) {                                     // fire-and-forget exact accumulation (exact_acc.h): no rows, no finalize launch
        long long* xp = a.xacc.acc + (size_t)(srow & (a.xacc.reps - 1)) * (XACC_DIGITS * 2) * a.Nout + n;
        xacc_add(xp, 2 * (size_t)a.Nout, s);
        xacc_add(xp + a.Nout, 2 * (size_t)a.Nout, q);
      } else {
        a.stats[((size_t)srow * 2 + 0) * a.Nout + n] = s;
        a.stats[((size_t)srow * 2 + 1) * a.Nout + n] = q;
      }
    }
  }
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  const T* __restrict__ addsrc = reinterpret_cast<const T*>(a.addsrc);
  constexpr int VPRO = BN / VEC;
  for (int v = tid; v < BM * VPRO; v += NT) {
    const int row = v / VPRO, cv = v - row * VPRO;
    const int m = tile_m * BM + row, n = tile_n * BN + cv * VEC;
    if (m < a.M && n < a.Nout) {
      uint4 d = *reinterpret_cast<const uint4*>(smem + row * SROW + cv * 16);
      if (addsrc) {
        float x[VEC], y[VEC];
        ET<T>::unpack(d, x);
        ET<T>::unpack(*reinterpret_cast<const uint4*>(addsrc + ((size_t)m * a.add_ldc + n)), y);
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[e] += y[e];
        d = ET<T>::pack(x);
      }
      *reinterpret_cast<uint4*>(out + ((size_t)m * a.out_ldc + n)) = d;
    }
  }
}

template <typename T, int MODE, int BM, int BN, int WM, int WN, int KT>
int launch_conv(const ConvArgs& a0, hipStream_t st) {
  if (a0.epi.oscale || a0.epi.act) return MDCV_EARG;          // the register-staged kernels carry no inference epilogue

  ConvArgs a = a0;
  constexpr int RB = 64 * KT + 16;
  constexpr int PIPE = 2 * (BM + BN) * RB;
  constexpr int STAGE = BM * (BN * (int)sizeof(T) + 16);
  constexpr int LDS = (PIPE > STAGE ? PIPE : STAGE) + WM * 2 * BN * 4;
  static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  static DynLds dyn_lds;
  auto kern = conv_igemm_kernel<T, MODE, BM, BN, WM, WN, KT>;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(kern), LDS); e != hipSuccess) return (int)e;
  a.tiles_n = cdiv(a.Nout, BN);
  a.tiles_total = cdiv(a.M, BM) * a.tiles_n;
  a.xcd_chunk = cdiv(a.tiles_total, 8);
  MDCV_LAUNCH(kern, dim3((unsigned)(a.xcd_chunk * 8)), dim3(WM * WN * 64), LDS, st, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant: tiles go HBM -> LDS directly (buffer_load_dwordx4 ... lds), no VGPR staging and no ds_write pass.
// An LDS-DMA writes lane-linear: 64 lanes x 16 B = one 1 KiB chunk = 16 tile rows of 64 B (4 lanes per row), so rows are NOT
// padded; bank conflicts of the ds_read_b128 fragment reads are removed by swizzling on the SOURCE side instead: the lane that
// fills 16-byte slot s of row r fetches logical k-vector  s ^ f(r),  f(r) = (-(r >> 2)) & 3, and the fragment read of
// k-vector q goes to slot q ^ f(r).  With this f every 16-lane service group of ds_read_b128 touches 16 distinct slots.
// Out-of-image taps / tail rows use an out-of-range buffer offset: the hardware range check returns zeros into LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

template <typename T> struct FragSwz;
template <> struct FragSwz<bf16_t> {
  // split form: fragment reads first, MFMAs later, so independent work (DMA address arithmetic) can sit under the LDS latency
  template <int FM, int FN>
  __device__ static __forceinline__ void load(const unsigned char* sa, const unsigned char* sb, int lane, bf16x8_t (&a)[FM], bf16x8_t (&b)[FN]) {
    const int r = lane & 15;
    const int off = r * 64 + (((lane >> 4) ^ swz(r)) << 4);
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 1024 + off);
#pragma unroll
    for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sb + j * 1024 + off);
  }
  template <int FM, int FN>
  __device__ static __forceinline__ void compute(const bf16x8_t (&a)[FM], const bf16x8_t (&b)[FN], f32x4_t (&acc)[FM][FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  template <int FM, int FN>
  __device__ static __forceinline__ void mma(const unsigned char* sa, const unsigned char* sb, int lane, f32x4_t (&acc)[FM][FN]) {
    bf16x8_t a[FM], b[FN];
    const int r = lane & 15;
    const int off = r * 64 + (((lane >> 4) ^ swz(r)) << 4);
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 1024 + off);
#pragma unroll
    for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sb + j * 1024 + off);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};
template <> struct FragSwz<float> {
  template <int FM, int FN>
  __device__ static __forceinline__ void mma(const unsigned char* sa, const unsigned char* sb, int lane, f32x4_t (&acc)[FM][FN]) {
    const int r = lane & 15;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[FM], b[FN];
      const int off = r * 64 + ((ks ^ swz(r)) << 4) + (lane >> 4) * 4;
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const float*>(sa + i * 1024 + off);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const float*>(sb + j * 1024 + off);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};

typedef __attribute__((address_space(3))) void lds_void_t;

// ALLCLS (MODE 2 only): the workgroup computes ALL FOUR output-parity classes of its tile of dY positions, one after the other (classes
// have 1, 2, 2 and 4 taps: every workgroup gets the same 9 taps of work, and the dY rows a tile reads come from HBM once instead of
// once per class launch -- 177 MB of dY per launch at 208^2 x 64 channels).  Needs even Hout / Wout (all classes share one Hs x Ws grid).
template <typename T, int MODE, int BM, int BN, int WM, int WN, int STAGES, bool UT, bool FUSE, bool EPI = false, bool ALLCLS = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_glds_kernel(ConvArgs a0, unsigned in_bytes, unsigned w_bytes) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int VEC = ET<T>::VEC;
  constexpr int BK = 4 * VEC;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int CA = BM / 16, CB = BN / 16;               // 1 KiB chunks per operand tile
  constexpr int NPA = (CA + NW - 1) / NW, NPB = (CB + NW - 1) / NW;
  // The deep ring waits with counted vmcnt, so every wave must issue the same number of DMAs per K tile: when an operand tile has
  // fewer 1 KiB chunks than there are waves (32- and 16-channel weight tiles), the surplus waves fill a 1 KiB sink with zeros.
  constexpr bool UNEVEN = STAGES > 2 && (CA % NW != 0 || CB % NW != 0);
  constexpr int SINK = STAGES * (BM + BN) * 64;
  constexpr int PIPE = SINK + (UNEVEN ? 1024 : 0);
  constexpr int GD = NPA + NPB;                           // LDS-DMA instructions every wave issues per K tile (deep pipeline: exact)
  constexpr int SROW = BN * (int)sizeof(T) + 16;
  constexpr int STAGE = BM * SROW;
  constexpr int STAT_OFF = PIPE > STAGE ? PIPE : STAGE;
  constexpr unsigned OOB = 0x80000000u;                   // >= num_records of any descriptor we build (sizes are < 2 GiB)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  static_assert(!ALLCLS || MODE == 2, "ALLCLS is a stride-2 data-gradient form");
  const int logical = (int)(blockIdx.x & 7) * a0.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a0.tiles_total) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // ALLCLS with cls_split: the tile's work is cut into two workgroups of 4 and 5 tap-GEMMs -- class (1,1) alone and classes (0,0), (0,1),
  // (1,0) -- that sit next to each other in the launch order (same XCD: the dY rows they both read meet in its L2).  The 26->52 and
  // 13->26 layers at batch 32 give only 338 / 172 tiles of 128 x 128: one workgroup per tile walked its nine tap-GEMMs on a
  // half-empty chip, four class launches did the same one class at a time.
  int tile_id = logical, cls_lo = 0, cls_hi = ALLCLS ? 4 : 1;
  if constexpr (ALLCLS) {
    if (a0.cls_split) { tile_id = logical >> 1; if (logical & 1) cls_hi = 3; else cls_lo = 3; }
  }
  const int tile_m = tile_id / a0.tiles_n, tile_n = tile_id % a0.tiles_n;
  const int lrow = lane >> 2;                              // row inside a chunk this lane fills
  const int kv = (lane & 3) ^ swz(lrow);                   // logical k-vector it fetches for that slot
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a0.in), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a0.w), 0, w_bytes, 0x00020000);
#pragma unroll 1
  for (int cls = cls_lo; cls < cls_hi; ++cls) {
  ConvArgs a = a0;
  if constexpr (ALLCLS) {                                  // class (ph, pw): its live taps kh = kh0 + 2i, kw = kw0 + 2j (see conv2d_impl)
    a.ph = cls >> 1; a.pw = cls & 1;
    a.kh0 = (a.ph + a.pad) & 1; a.kw0 = (a.pw + a.pad) & 1;
    a.nkh = (a.KH - a.kh0 + 1) / 2; a.nkw = (a.KW - a.kw0 + 1) / 2;
    a.Ktot = a.nkh * a.nkw * a.Cin;
    a.fuse.row_base = cls * ((a.M + 127) >> 7);
    if (cls != cls_lo) __syncthreads();                    // the previous class's epilogue is done with the LDS
  }

  int bh[NPA], bw[NPA], ib[NPA];
  bool rv[NPA];
  // MODE 2 enumerates only the output pixels (h,w) = (ph + 2a, pw + 2b) of one parity class and only the taps whose
  // parity matches (kh = kh0 + 2i, kw = kw0 + 2j): every tap it visits is a real MAC.
  const int HWo = MODE == 2 ? a.Hs * a.Ws : a.Hout * a.Wout;
  const int Wrow = MODE == 2 ? a.Ws : a.Wout;
  const int KWn = MODE == 2 ? a.nkw : a.KW, KHn = MODE == 2 ? a.nkh : a.KH;
#pragma unroll
  for (int p = 0; p < NPA; ++p) {
    const int chunk = wave + p * NW;
    const int m = tile_m * BM + chunk * 16 + lrow;
    rv[p] = chunk < CA && m < a.M;
    const int mm = rv[p] ? m : 0;
    const int img = mm / HWo, rem = mm - img * HWo;
    const int ho = rem / Wrow, wo = rem - ho * Wrow;
    ib[p] = img * a.Hin * a.Win;
    if (MODE == 0) { bh[p] = ho * a.stride - a.pad; bw[p] = wo * a.stride - a.pad; }
    else if (MODE == 1) { bh[p] = ho + a.pad;       bw[p] = wo + a.pad; }
    else { bh[p] = 2 * ho + a.ph + a.pad - a.kh0;   bw[p] = 2 * wo + a.pw + a.pad - a.kw0; }   // always even
  }
  int kc, kh, kw;
  {
    const int k0 = kv * VEC, tap = k0 / a.Cin;
    kc = k0 - tap * a.Cin; kh = tap / KWn; kw = tap - kh * KWn;
  }
  const int smask = a.stride - 1;

  // ---- UT (uniform tap): Cin % BK == 0, so a K tile never straddles a tap and the whole wave walks the taps together.
  // The tap cursor then lives in SGPRs and each DMA needs only: 2 adds + 2 unsigned compares + 1 select per pixel row.
  int rowoff[NPA], rowh[NPA], roww[NPA], nboff[NPB];
  bool nv[NPB];
  if (UT) {
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
      rowh[p] = MODE == 2 ? (bh[p] >> 1) : bh[p];
      roww[p] = MODE == 2 ? (bw[p] >> 1) : bw[p];
      rowoff[p] = ((ib[p] + rowh[p] * a.Win + roww[p]) * a.in_ldc + kv * VEC) * (int)sizeof(T);
    }
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int chunk = wave + p * NW;
      const int n = tile_n * BN + chunk * 16 + lrow;
      nv[p] = chunk < CB && n < a.Nout;
      nboff[p] = (n * (a.KH * a.KW * a.Cin) + kv * VEC) * (int)sizeof(T);
    }
  }
  int s_c0 = 0, s_kh = 0, s_kw = 0;       // wave-uniform tap cursor (UT)
  auto issue_tile_ut = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * (BM + BN) * 64;
    unsigned char* sB = sA + BM * 64;
    const bool kvalid = s_kh < KHn;
    // tap displacement of the source pixel (in pixels) and in bytes
    const int dh = MODE == 0 ? s_kh * a.dil : -(MODE == 1 ? s_kh * a.dil : s_kh);
    const int dw = MODE == 0 ? s_kw * a.dil : -(MODE == 1 ? s_kw * a.dil : s_kw);
    const int tapoff = ((dh * a.Win + dw) * a.in_ldc + s_c0) * (int)sizeof(T);
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
      const int chunk = wave + p * NW;
      if (CA % NW == 0 || chunk < CA) {
        const int hi = rowh[p] + dh, wi = roww[p] + dw;
        const bool ok = rv[p] & kvalid & ((unsigned)hi < (unsigned)a.Hin) & ((unsigned)wi < (unsigned)a.Win);
        const unsigned off = ok ? (unsigned)(rowoff[p] + tapoff) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_t*)(sA + chunk * 1024), 16, off, 0, 0, 0);
      } else if (UNEVEN) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_t*)(smem + SINK), 16, OOB, 0, 0, 0);
      }
    }
    const int kw_full = MODE == 2 ? ((a.kh0 + 2 * s_kh) * a.KW + a.kw0 + 2 * s_kw) * a.Cin + s_c0 : kt * BK;
    const int koff = kw_full * (int)sizeof(T);
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int chunk = wave + p * NW;
      if (CB % NW == 0 || chunk < CB) {
        const unsigned off = (nv[p] & kvalid) ? (unsigned)(nboff[p] + koff) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(sB + chunk * 1024), 16, off, 0, 0, 0);
      } else if (UNEVEN) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(smem + SINK), 16, OOB, 0, 0, 0);
      }
    }
  };
  auto advance_ut = [&]() {
    s_c0 += BK;
    if (s_c0 >= a.Cin) { s_c0 = 0; if (++s_kw == KWn) { s_kw = 0; ++s_kh; } }
  };
  auto issue_tile_gen = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * (BM + BN) * 64;
    unsigned char* sB = sA + BM * 64;
    const bool kvalid = kh < KHn;
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
      const int chunk = wave + p * NW;
      if (CA % NW == 0 || chunk < CA) {
        int hi, wi; bool ok = rv[p] & kvalid;
        if (MODE == 0) { hi = bh[p] + kh * a.dil; wi = bw[p] + kw * a.dil; }
        else if (MODE == 2) { hi = (bh[p] >> 1) - kh; wi = (bw[p] >> 1) - kw; }
        else {
          const int th = bh[p] - kh * a.dil, tw = bw[p] - kw * a.dil;
          ok = ok & (th >= 0) & (tw >= 0) & (((th | tw) & smask) == 0);
          hi = th >> a.sshift; wi = tw >> a.sshift;
        }
        ok = ok & ((unsigned)hi < (unsigned)a.Hin) & ((unsigned)wi < (unsigned)a.Win);
        const unsigned off = ok ? (unsigned)(((ib[p] + hi * a.Win + wi) * a.in_ldc + kc) * (int)sizeof(T)) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_t*)(sA + chunk * 1024), 16, off, 0, 0, 0);
      } else if (UNEVEN) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_t*)(smem + SINK), 16, OOB, 0, 0, 0);
      }
    }
    // weight row = [KH][KW][Cin] of the FULL kernel; MODE 2 visits the sub-lattice of taps
    const int k = MODE == 2 ? ((a.kh0 + 2 * kh) * a.KW + a.kw0 + 2 * kw) * a.Cin + kc : kt * BK + kv * VEC;
    const int wrow = a.KH * a.KW * a.Cin;
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int chunk = wave + p * NW;
      if (CB % NW == 0 || chunk < CB) {
        const int n = tile_n * BN + chunk * 16 + lrow;
        const unsigned off = ((n < a.Nout) & kvalid & (k < wrow)) ? (unsigned)((n * wrow + k) * (int)sizeof(T)) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(sB + chunk * 1024), 16, off, 0, 0, 0);
      } else if (UNEVEN) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(smem + SINK), 16, OOB, 0, 0, 0);
      }
    }
  };
  auto advance_gen = [&]() {
    kc += BK;
    while (kc >= a.Cin) { kc -= a.Cin; if (++kw == KWn) { kw = 0; ++kh; } }
  };
  auto issue_tile = [&](int kt, int buf) { if (UT) issue_tile_ut(kt, buf); else issue_tile_gen(kt, buf); };
  auto advance = [&]() { if (UT) advance_ut(); else advance_gen(); };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (a.Ktot + BK - 1) / BK;
  if (STAGES == 2) {
    issue_tile(0, 0);
    __syncthreads();                                 // (the compiler drains the LDS-DMA queue, vmcnt(0), ahead of the barrier)
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) { advance(); issue_tile(kt + 1, cur ^ 1); }
      const unsigned char* sA = smem + cur * (BM + BN) * 64 + wm * TM * 64;
      const unsigned char* sB = smem + cur * (BM + BN) * 64 + BM * 64 + wn * TN * 64;
      FragSwz<T>::template mma<FM, FN>(sA, sB, lane, acc);
      __syncthreads();
    }
  } else {
    // STAGES-deep ring: tiles kt+1 .. kt+STAGES-2 stay in flight ACROSS the barrier.  Only counted waits (never vmcnt(0) in
    // steady state) and a raw s_barrier, because __syncthreads() would drain the DMA queue.  Order per iteration:
    //   wait(tile kt landed for THIS wave) -> barrier (landed for ALL waves; everyone is done reading the slot reused next)
    //   -> issue tile kt+STAGES-1 into the slot read at iteration kt-1 -> MFMAs on tile kt.
    int issued = 0;
    for (; issued < STAGES - 1 && issued < nk; ++issued) { if (issued) advance(); issue_tile(issued, issued); }
    int slot = 0, islot = issued % STAGES;
    int kt = 0;
    // steady state: every iteration issues exactly one tile, so the wait count is a constant and the body is ONE basic
    // block (no branches): the DMA address arithmetic can be scheduled into the issue gaps between the MFMAs.
    for (const int nmain = nk - (STAGES - 1); kt < nmain; ++kt) {
      // (lgkmcnt(0): this wave's fragment reads of the previous tile are DONE before anyone may refill that slot -- the compiler sinks a
      //  tile's last MFMAs and their LDS waits below this barrier, and an LDS-DMA write is not ordered against queued ds_reads; conv_shift.hip)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GD * (STAGES - 2)) : "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned char* sA = smem + slot * (BM + BN) * 64 + wm * TM * 64;
      const unsigned char* sB = smem + slot * (BM + BN) * 64 + BM * 64 + wn * TN * 64;
      if constexpr (sizeof(T) == 2) {
        bf16x8_t fa[FM], fb[FN];
        FragSwz<T>::template load<FM, FN>(sA, sB, lane, fa, fb);
        advance();
        issue_tile(kt + STAGES - 1, islot);
        FragSwz<T>::template compute<FM, FN>(fa, fb, acc);
      } else {
        advance();
        issue_tile(kt + STAGES - 1, islot);
        FragSwz<T>::template mma<FM, FN>(sA, sB, lane, acc);
      }
      islot = islot + 1 == STAGES ? 0 : islot + 1;
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    for (; kt < nk; ++kt) {                              // drain: no more tiles to issue
      const int newer = nk - 1 - kt;                     // tiles issued after tile kt (<= STAGES - 2)
      if (newer >= 2 && STAGES > 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GD * 2) : "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned char* sA = smem + slot * (BM + BN) * 64 + wm * TM * 64;
      const unsigned char* sB = smem + slot * (BM + BN) * 64 + BM * 64 + wn * TN * 64;
      FragSwz<T>::template mma<FM, FN>(sA, sB, lane, acc);
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    __syncthreads();                                   // the epilogue reuses the ring as staging
  }

  // ---------------- epilogue (same as the register-staged kernel) ----------------
  const int n0 = tile_n * BN + wn * TN, m0 = tile_m * BM + wm * TM;
  if constexpr (EPI) {                                     // inference instantiation (MODE 0): act(acc * scale + shift), once per tile (template
    // parameter: as a runtime branch it cost the training step 2 %)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + j * 16 + (lane & 15);
      const float sc = (a.epi.oscale && n < a.Nout) ? a.epi.oscale[n] : 1.f;
      const float bv = (a.bias && n < a.Nout) ? a.bias[n] : 0.f;
      const float sl = a.epi.act == 1 ? a.epi.slope : (a.epi.act == 2 ? 0.f : 1.f);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r] * sc + bv;
          acc[i][j][r] = v > 0.f ? v : v * sl;
        }
    }
  } else if (a.bias) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + j * 16 + (lane & 15);
      const float bv = n < a.Nout ? a.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += bv;
    }
  }
  float* sstat = reinterpret_cast<float*>(smem + STAT_OFF);
  const bool want_stats = a.stats || a.xacc.acc;
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
          const float v = m < a.M ? acc[i][j][r] : 0.f;
          s += v; q += v * v;
        }
      s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) {
        sstat[(wm * 2 + 0) * BN + wn * TN + j * 16 + lane] = s;
        sstat[(wm * 2 + 1) * BN + wn * TN + j * 16 + lane] = q;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * TM + i * 16 + (lane >> 4) * 4 + r, col = wn * TN + j * 16 + (lane & 15);
        ET<T>::st(reinterpret_cast<T*>(smem + row * SROW) + col, acc[i][j][r]);
      }
  __syncthreads();
  constexpr int G = BM / 128 > 0 ? BM / 128 : 1, WPG = WM / G;
  if (want_stats && tid < BN * G) {
    const int g = tid / BN, col = tid - g * BN;
    const int n = tile_n * BN + col, srow = tile_m * G + g;
    if (n < a.Nout && srow * 128 < a.M) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < WPG; ++r) { s += sstat[((g * WPG + r) * 2 + 0) * BN + col]; q += sstat[((g * WPG + r) * 2 + 1) * BN + col]; }
      if (a.xacc.acc) {                                     // fire-and-forget exact accumulation (exact_acc.h): no rows, no finalize launch
        long long* xp = a.xacc.acc + (size_t)(srow & (a.xacc.reps - 1)) * (XACC_DIGITS * 2) * a.Nout + n;
        xacc_add(xp, 2 * (size_t)a.Nout, s);
        xacc_add(xp + a.Nout, 2 * (size_t)a.Nout, q);
      } else {
        a.stats[((size_t)srow * 2 + 0) * a.Nout + n] = s;
        a.stats[((size_t)srow * 2 + 1) * a.Nout + n] = q;
      }
    }
  }
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  const T* __restrict__ addsrc = reinterpret_cast<const T*>(a.addsrc);
  constexpr int VPRO = BN / VEC;
  if constexpr (FUSE) {
    // data gradient with the BatchNorm-backward sums of the producer layer folded in (bn_fuse.h)
    using Acc = BnFuseAcc<T, BN, NT>;
    Acc fz;
    const int cv = tid % VPRO, n = tile_n * BN + cv * VEC;
    fz.init(a.fuse, n, a.Nout);
    const T* __restrict__ fy = reinterpret_cast<const T*>(a.fuse.y);
    float* fred = sstat;
    constexpr int PPG = 128 / Acc::RPP;                    // passes per 128-pixel group
    // The global loads (addsrc, y) of ALL groups of the tile are issued before the first group is processed (like the shift kernel):
    // they are HBM misses, the accumulators are dead by now, and one batch per group exposed their latency once per group.
    constexpr int NG = BM / 128 > 0 ? BM / 128 : 1;
    long long pixv[NG][PPG]; uint4 aq[NG][PPG], yq[NG][PPG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
      for (int u = 0; u < PPG; ++u) {
        const int row = gi * 128 + u * Acc::RPP + tid / VPRO;
        const int m = tile_m * BM + row;
        pixv[gi][u] = -1;
        if (m < a.M && n < a.Nout) {
          long long pix = m;
          if (MODE == 2) {
            const int img = m / HWo, rem = m - img * HWo;
            const int ha = rem / Wrow, wb = rem - ha * Wrow;
            pix = ((long long)img * a.Hout + (a.ph + 2 * ha)) * a.Wout + (a.pw + 2 * wb);
          }
          pixv[gi][u] = pix;
          if (addsrc) aq[gi][u] = *reinterpret_cast<const uint4*>(addsrc + (pix * a.add_ldc + n));
          yq[gi][u] = *reinterpret_cast<const uint4*>(fy + (pix * a.fuse.ldy + n));
        }
      }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int g0 = gi * 128;
      uint4 dq[PPG];
#pragma unroll
      for (int u = 0; u < PPG; ++u) dq[u] = *reinterpret_cast<const uint4*>(smem + (g0 + u * Acc::RPP + tid / VPRO) * SROW + cv * 16);
#pragma unroll
      for (int u = 0; u < PPG; ++u) {
        if (pixv[gi][u] >= 0) {
          float x[VEC];
          uint4 d = dq[u];
          ET<T>::unpack(d, x);
          if (addsrc) {
            float y[VEC];
            ET<T>::unpack(aq[gi][u], y);
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[e] += y[e];
            d = ET<T>::pack(x);
            ET<T>::unpack(d, x);
          }
          *reinterpret_cast<uint4*>(out + (pixv[gi][u] * a.out_ldc + n)) = d;
          fz.add(a.fuse, x, yq[gi][u]);
        }
      }
      if (tile_m * BM + g0 < a.M) {                    // block-uniform: a 256-row tile's second group may start past the last pixel,
        if constexpr (Acc::kWaveFold)                    // and that row is not in the caller's buffer
          fz.fold_wave(reinterpret_cast<float*>(smem + (g0 + wave * Acc::WROWS) * SROW), lane);   // the rows this wave read in its first pass: dead, private
        else
          fz.flush(a.fuse, fred, tid, tile_n * BN, a.Nout, (tile_m * BM + g0) >> 7);
      }
    }
    if constexpr (Acc::kWaveFold) {                      // the waves meet once, behind the tile's last store
      static_assert(Acc::WROWS * SROW >= 2 * BN * 4 && SROW % 4 == 0, "a wave's dead staging rows hold its 2*BN sums");
      lds_only_barrier();
      for (int t = tid; t < NG * 2 * BN; t += NT) {
        const int gi = t / (2 * BN), g0 = gi * 128;
        if (tile_m * BM + g0 < a.M)
          Acc::write_row(a.fuse, reinterpret_cast<const float*>(smem + g0 * SROW), Acc::WROWS * SROW / 4, t - gi * 2 * BN, tile_n * BN, a.Nout,
                         (tile_m * BM + g0) >> 7);
      }
    }
  } else {
  for (int v = tid; v < BM * VPRO; v += NT) {
    const int row = v / VPRO, cv = v - row * VPRO;
    const int m = tile_m * BM + row, n = tile_n * BN + cv * VEC;
    if (m < a.M && n < a.Nout) {
      size_t pix = (size_t)m;
      if (MODE == 2) {                                   // class-local index -> full-resolution output pixel
        const int img = m / HWo, rem = m - img * HWo;
        const int ha = rem / Wrow, wb = rem - ha * Wrow;
        pix = ((size_t)img * a.Hout + (a.ph + 2 * ha)) * a.Wout + (a.pw + 2 * wb);
      }
      uint4 d = *reinterpret_cast<const uint4*>(smem + row * SROW + cv * 16);
      if (addsrc) {
        float x[VEC], y[VEC];
        ET<T>::unpack(d, x);
        ET<T>::unpack(*reinterpret_cast<const uint4*>(addsrc + (pix * a.add_ldc + n)), y);
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[e] += y[e];
        d = ET<T>::pack(x);
      }
      *reinterpret_cast<uint4*>(out + (pix * a.out_ldc + n)) = d;
    }
  }
  }
  }   // class loop
}

template <typename T, int MODE, int BM, int BN, int WM, int WN, int STAGES, bool UT, bool FUSE, bool EPI = false, bool ALLCLS = false>
int launch_conv_glds_f(const ConvArgs& a0, hipStream_t st, int B) {
  ConvArgs a = a0;
  constexpr int NWV = WM * WN;
  constexpr int PIPE = STAGES * (BM + BN) * 64 + ((STAGES > 2 && ((BM / 16) % NWV != 0 || (BN / 16) % NWV != 0)) ? 1024 : 0);   // + the DMA sink
  constexpr int STAGE = BM * (BN * (int)sizeof(T) + 16);
  constexpr int LDS = (PIPE > STAGE ? PIPE : STAGE) + WM * 2 * BN * 4;   // + statistics / fp32 fused-sum scratch (NW*BN floats <= WM*2*BN)
  static DynLds dyn_lds;
  auto kern = conv_glds_kernel<T, MODE, BM, BN, WM, WN, STAGES, UT, FUSE, EPI, ALLCLS>;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(kern), LDS); e != hipSuccess) return (int)e;
  a.tiles_n = cdiv(a.Nout, BN);
  a.tiles_total = cdiv(a.M, BM) * a.tiles_n * ((ALLCLS && a.cls_split) ? 2 : 1);
  a.xcd_chunk = cdiv(a.tiles_total, 8);
  const unsigned in_bytes = (unsigned)((long long)B * a.Hin * a.Win * a.in_ldc * (long long)sizeof(T));
  const unsigned w_bytes = (unsigned)((long long)a.Nout * a.KH * a.KW * a.Cin * (long long)sizeof(T));
  MDCV_LAUNCH(kern, dim3((unsigned)(a.xcd_chunk * 8)), dim3(WM * WN * 64), LDS, st, a, in_bytes, w_bytes);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

template <typename T, int MODE, int BM, int BN, int WM, int WN, int STAGES, bool UT, bool ALLCLS = false>
int launch_conv_glds_ut(const ConvArgs& a, hipStream_t st, int B) {
  if constexpr (ALLCLS) {
    if (a.epi.oscale || a.epi.act) return MDCV_EARG;
    if (a.fuse.y) return launch_conv_glds_f<T, MODE, BM, BN, WM, WN, STAGES, UT, true, false, true>(a, st, B);
    return launch_conv_glds_f<T, MODE, BM, BN, WM, WN, STAGES, UT, false, false, true>(a, st, B);
  }
  if constexpr (MODE != 0) {                    // the fused BatchNorm-backward sums exist for data gradients only
    if (a.fuse.y) return launch_conv_glds_f<T, MODE, BM, BN, WM, WN, STAGES, UT, true>(a, st, B);
  }
  if constexpr (MODE == 0) {                    // inference epilogue (forward only)
    if (a.epi.oscale || a.epi.act) return launch_conv_glds_f<T, MODE, BM, BN, WM, WN, STAGES, UT, false, true>(a, st, B);
  }
  if (a.epi.oscale || a.epi.act) return MDCV_EARG;
  return launch_conv_glds_f<T, MODE, BM, BN, WM, WN, STAGES, UT, false>(a, st, B);
}

template <typename T, int MODE, int BM, int BN, int WM, int WN, int STAGES = 2, bool ALLCLS = false>
int launch_conv_glds(const ConvArgs& a, hipStream_t st, int B) {
  constexpr int BK = 4 * ET<T>::VEC;
  // uniform-tap fast path: K tiles never straddle a tap; the generic stride-2 dgrad (MODE 1, stride 2) keeps the per-lane cursor
  const bool ut = (a.Cin % BK == 0) && !(MODE == 1 && a.stride != 1) && TUNE().conv_no_ut == 0;
  if constexpr (ALLCLS) {                       // (bf16 layers of a Darknet: Cin is a multiple of 32; others keep the four launches)
    if (!ut) return MDCV_EARG;
    return launch_conv_glds_ut<T, MODE, BM, BN, WM, WN, STAGES, true, true>(a, st, B);
  }
  if (ut) return launch_conv_glds_ut<T, MODE, BM, BN, WM, WN, STAGES, true>(a, st, B);
  return launch_conv_glds_ut<T, MODE, BM, BN, WM, WN, STAGES, false>(a, st, B);
}


template <typename T, int MODE>
int dispatch_conv(const ConvArgs& a, hipStream_t st, int B) {
  constexpr bool BF = sizeof(T) == 2;
  // the LDS-DMA kernels address operands through 32-bit buffer offsets: both operands must be < 2 GiB
  const bool small = (long long)B * a.Hin * a.Win * a.in_ldc * (long long)sizeof(T) < (1LL << 31) &&
                     (long long)a.Nout * a.Ktot * (long long)sizeof(T) < (1LL << 31);
  if (a.Nout > 64) {
    int v = TUNE().conv_variant;
    if (v < 0) {   // measured on MI355X (scripts/conv_ab.py): tall tiles once the grid is >= 4 waves of CUs, half-width tiles
      const long long t128 = (long long)cdiv(a.M, 128) * cdiv(a.Nout, 128);   // when 128x128 would leave CUs idle
      const int nk = a.Ktot / (4 * ET<T>::VEC);      // long K loops profit from the 3-stage DMA ring (scripts/conv_ab.py)
      v = t128 >= 1024 ? 11 : (nk >= 100 ? 9 : (t128 >= 300 ? 6 : 7));
      if (TUNE().conv_fuse_narrow && MODE == 1 && a.fuse.y && a.KH == 1 && a.KW == 1) v = 10;
      if (TUNE().conv_deep_small && nk >= TUNE().conv_deep_small && (v == 6 || v == 7)) v += 3;   // 3-stage ring for the mid / sparse grids too
    }
    if (!BF && small) v = v == 8 ? 6 : (v == 11 ? 9 : v);   // the 8-wave 256-row tiles exist in bf16 only: fp32 takes the 128x128 LDS-DMA tiles
                                                                           // (the register-staged fallback has no inference epilogue and is slower)
    if (v >= 6 && !small) v = (v == 8 || v == 11) ? 2 : ((v == 7 || v == 10) ? 4 : 0);
    if (v == 6) return launch_conv_glds<T, MODE, 128, 128, 2, 2>(a, st, B);
    if (v == 7) return launch_conv_glds<T, MODE, 128, 64, 2, 2>(a, st, B);
    if (v == 9) return launch_conv_glds<T, MODE, 128, 128, 2, 2, 3>(a, st, B);
    if (v == 10) return launch_conv_glds<T, MODE, 128, 64, 2, 2, 3>(a, st, B);
    if (BF) {   // 8-wave / deep-K tiles only exist in the production dtype
      if (v == 8) return launch_conv_glds<T, MODE, (BF ? 256 : 128), 128, (BF ? 4 : 2), 2>(a, st, B);
      if (v == 11) return launch_conv_glds<T, MODE, (BF ? 256 : 128), 128, (BF ? 4 : 2), 2, 3>(a, st, B);
      if (v == 1) return launch_conv<T, MODE, 128, 128, 2, 2, (BF ? 2 : 1)>(a, st);
      if (v == 2) return launch_conv<T, MODE, (BF ? 256 : 128), 128, (BF ? 4 : 2), 2, 1>(a, st);
      if (v == 3) return launch_conv<T, MODE, (BF ? 256 : 128), 128, (BF ? 4 : 2), 2, (BF ? 2 : 1)>(a, st);
      if (v == 4) return launch_conv<T, MODE, 128, 64, 2, 2, (BF ? 2 : 1)>(a, st);
      if (v == 5) return launch_conv<T, MODE, 128, 64, 2, 2, 1>(a, st);
    } else if (v == 4 || v == 5) {
      return launch_conv<T, MODE, 128, 64, 2, 2, 1>(a, st);
    }
    return launch_conv<T, MODE, 128, 128, 2, 2, 1>(a, st);
  }
  const bool dma = small && TUNE().conv_variant != 0;      // variant 0 forces the register-staged kernels everywhere (A/B)
  if constexpr (BF) {
    if (dma && TUNE().conv_tall_narrow && a.M >= TUNE().conv_tall_narrow * 1024) {   // tall tiles for the narrow layers of large images
      if (a.Nout > 32) return launch_conv_glds<T, MODE, 256, 64, 4, 2, 3>(a, st, B);
      if (a.Nout > 16) return launch_conv_glds<T, MODE, 256, 32, 4, 1>(a, st, B);
      return launch_conv_glds<T, MODE, 256, 16, 4, 1>(a, st, B);
    }
  }
  if (dma && TUNE().conv_deep_narrow && a.Ktot / (4 * ET<T>::VEC) >= TUNE().conv_deep_narrow) {
    if (a.Nout > 32) return launch_conv_glds<T, MODE, 128, 64, 2, 2, 3>(a, st, B);
  }
  if (a.Nout > 32) return dma ? launch_conv_glds<T, MODE, 128, 64, 2, 2>(a, st, B) : launch_conv<T, MODE, 128, 64, 2, 2, (BF ? 2 : 1)>(a, st);
  if (a.Nout > 16) return dma ? launch_conv_glds<T, MODE, 128, 32, 4, 1>(a, st, B) : launch_conv<T, MODE, 128, 32, 4, 1, (BF ? 2 : 1)>(a, st);
  return dma ? launch_conv_glds<T, MODE, 128, 16, 4, 1>(a, st, B) : launch_conv<T, MODE, 128, 16, 4, 1, (BF ? 2 : 1)>(a, st);
}



// all four classes in one launch: same tile choice as the per-class dispatch below (a.M = positions of ONE class)
static int dispatch_dgrad_s2_all(const ConvArgs& a, hipStream_t st, int B) {
  typedef bf16_t T;
  if (TUNE().conv_tall_s2 && TUNE().conv_tall_narrow && a.Nout <= 64 && a.M >= TUNE().conv_tall_narrow * 1024) {
    if (a.Nout > 32) return launch_conv_glds<T, 2, 256, 64, 4, 2, 3, true>(a, st, B);
    if (a.Nout > 16) return launch_conv_glds<T, 2, 256, 32, 4, 1, 2, true>(a, st, B);
    return MDCV_EARG;
  }
  // Measured per layer of yolo_baseline at batch 32 (one launch vs four): 208->416 279 -> 204 us, 104->208 139 -> 125, 52->104 98 -> 86, but
  // 26->52 (338 tiles of 128 x 128: one sparse round of long workgroups) 136 -> 183 and 13->26 140 -> 139: only grids of >= 512 tiles take it.
  const long long t128 = (long long)cdiv(a.M, 128) * cdiv(a.Nout, 128);
  if (a.Nout <= 64) return MDCV_EARG;
  if (t128 < TUNE().conv_s2_split) {                    // sparse grids: two workgroups per tile (4 + 5 tap-GEMMs), see conv_glds_kernel
    if (!TUNE().conv_s2_split_on) return MDCV_EARG;
    ConvArgs c = a;
    c.cls_split = 1;
    return launch_conv_glds<T, 2, 128, 128, 2, 2, 3, true>(c, st, B);
  }
  if (t128 >= 1024) return launch_conv_glds<T, 2, 256, 128, 4, 2, 3, true>(a, st, B);
  return launch_conv_glds<T, 2, 128, 128, 2, 2, 3, true>(a, st, B);
}

template <typename T>
int dispatch_dgrad_s2(const ConvArgs& a, hipStream_t st, int B) {
  if constexpr (sizeof(T) == 2) {
    if (TUNE().conv_tall_s2 && TUNE().conv_tall_narrow && a.Nout <= 64 && a.M >= TUNE().conv_tall_narrow * 1024) {
      if (a.Nout > 32) return launch_conv_glds<T, 2, 256, 64, 4, 2, 3>(a, st, B);
      if (a.Nout > 16) return launch_conv_glds<T, 2, 256, 32, 4, 1>(a, st, B);
      return launch_conv_glds<T, 2, 256, 16, 4, 1>(a, st, B);
    }
  }
  if (TUNE().conv_deep_s2 && a.Nout > 32) {
    const long long t128 = (long long)cdiv(a.M, 128) * cdiv(a.Nout, 128);
    if (a.Nout > 64) {
      if (sizeof(T) == 2 && t128 >= 1024) return launch_conv_glds<T, 2, (sizeof(T) == 2 ? 256 : 128), 128, (sizeof(T) == 2 ? 4 : 2), 2, 3>(a, st, B);
      if (t128 >= 300) return launch_conv_glds<T, 2, 128, 128, 2, 2, 3>(a, st, B);
    }
    return launch_conv_glds<T, 2, 128, 64, 2, 2, 3>(a, st, B);
  }
  if (a.Nout > 64) {
    const long long t128 = (long long)cdiv(a.M, 128) * cdiv(a.Nout, 128);
    if (sizeof(T) == 2 && t128 >= 1024) return launch_conv_glds<T, 2, (sizeof(T) == 2 ? 256 : 128), 128, (sizeof(T) == 2 ? 4 : 2), 2>(a, st, B);
    if (t128 >= 300) return launch_conv_glds<T, 2, 128, 128, 2, 2>(a, st, B);
    return launch_conv_glds<T, 2, 128, 64, 2, 2>(a, st, B);
  }
  if (a.Nout > 32) return launch_conv_glds<T, 2, 128, 64, 2, 2>(a, st, B);
  if (a.Nout > 16) return launch_conv_glds<T, 2, 128, 32, 4, 1>(a, st, B);
  return launch_conv_glds<T, 2, 128, 16, 4, 1>(a, st, B);
}

#if MDCV_CONV_PART == 0
// ------------------------------------------------------------------------------------------------
// weight gradient: dW[co, k] = sum_m dY[m, co] * Xcol[m, k]; pixels (the reduction) are split across the grid and each
// split writes an fp32 partial slab ws[split][Cout][Ktot]; mdcv_wgrad_reduce sums the slabs into the OIHW fp32 grad.
// Both operands are pixel-major in HBM, so tiles are transposed on the way into LDS ([channel][pixel] rows) with the
// channel<->row permutation  row = j*OQ + oct  (channel = oct*VEC + j)  which keeps the transposing ds_writes 2-way.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* dy; const void* x; float* ws;
  int dy_ldc, x_ldc;
  int Hin, Win, Cin, Hout, Wout, Cout;
  int KH, KW, stride, pad, dil;
  int M, Ktot, tiles_k, tiles_ck, pix_per_split, blocks_total, xcd_chunk;
  // BNA form of the narrow kernel (a layer whose input needs no gradient): `dy` holds dz, the operand dy = cA g + cB y + cC is formed in LDS
  const void* y; int y_ldc, act, creal; float slope;
  const float* s1; const float* b1; const float* cA; const float* cB; const float* cC;
};

// 128(co) x 128(k) output tile per block, 4 waves of 64x64.  One step = 128 pixels (bf16; 64 in fp32) = 256 bytes per LDS row:
// every thread issues the 16 global loads of the NEXT step before the 64 MFMAs of the current one (HBM latency is ~2 us under
// load, a 32-pixel step could not cover it), then the tile is transposed into the single LDS buffer between two barriers.
template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int VEC = ET<T>::VEC;
  constexpr int NP = 4;                // staging passes per step
  constexpr int BP = NP * 4 * VEC;     // pixels per step
  constexpr int OQ = 128 / VEC;        // 16-byte vectors per pixel across the 128-wide tile
  constexpr int PPP = 2 * (256 / OQ);  // pixels covered by one pass (two per thread)
  constexpr int RB = 64 * NP + 16;     // LDS row pitch in bytes
  constexpr int FM = 4, FN = 4;        // 2x2 waves, 64x64 per wave
  constexpr int OROW = 132;            // fp32 staging pitch
  static_assert(PPP * NP == BP, "pass geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);   // XCD-contiguous block order
  if (logical >= a.blocks_total) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int split = logical / a.tiles_ck;
  const int tck = logical - split * a.tiles_ck;
  const int tile_co = tck / a.tiles_k, tile_k = tck - tile_co * a.tiles_k;
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);

  const int oct = tid % OQ, pp = tid / OQ;           // this thread stages pixels 2pp, 2pp+1 of every pass
  const int co0 = tile_co * 128 + oct * VEC;
  const bool a_ok = co0 < a.Cout;
  const int kcol0 = tile_k * 128 + oct * VEC;
  const bool b_ok = kcol0 < a.Ktot;
  int dh, dw, ci;
  {
    const int kk = b_ok ? kcol0 : 0;
    const int tap = kk / a.Cin;
    ci = kk - tap * a.Cin;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    dh = kh * a.dil - a.pad; dw = kw * a.dil - a.pad;
  }
  const int p_begin = split * a.pix_per_split;
  const int p_end = min(a.M, p_begin + a.pix_per_split);
  const int HWo = a.Hout * a.Wout;

  uint4 ra[2 * NP], rb[2 * NP];
  auto load_step = [&](int m0) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int m = m0 + ps * PPP + 2 * pp;
      int img = m / HWo, rem = m - img * HWo;
      int ho = rem / a.Wout, wo = rem - ho * a.Wout;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool pv = m + e < p_end;
        uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
        if (pv && a_ok) va = *reinterpret_cast<const uint4*>(dy + ((size_t)(m + e) * a.dy_ldc + co0));
        const int hi = ho * a.stride + dh, wi = wo * a.stride + dw;
        if (pv && b_ok && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win)
          vb = *reinterpret_cast<const uint4*>(x + ((size_t)((img * a.Hin + hi) * a.Win + wi) * a.x_ldc + ci));
        ra[2 * ps + e] = va; rb[2 * ps + e] = vb;
        if (++wo == a.Wout) { wo = 0; if (++ho == a.Hout) { ho = 0; ++img; } }
      }
    }
  };
  auto store_step = [&]() {
    unsigned char* sA = smem;
    unsigned char* sB = smem + 128 * RB;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const unsigned a0[4] = {ra[2 * ps].x, ra[2 * ps].y, ra[2 * ps].z, ra[2 * ps].w}, a1[4] = {ra[2 * ps + 1].x, ra[2 * ps + 1].y, ra[2 * ps + 1].z, ra[2 * ps + 1].w};
      const unsigned b0[4] = {rb[2 * ps].x, rb[2 * ps].y, rb[2 * ps].z, rb[2 * ps].w}, b1[4] = {rb[2 * ps + 1].x, rb[2 * ps + 1].y, rb[2 * ps + 1].z, rb[2 * ps + 1].w};
      if (sizeof(T) == 2) {
        const int cb = ps * 64 + pp * 4;      // byte column of pixels (2pp, 2pp+1) of this pass
#pragma unroll
        for (int q = 0; q < 4; ++q) {         // channels 2q, 2q+1 of the octet ; word = (pixel 2pp | pixel 2pp+1 << 16)
          *reinterpret_cast<unsigned*>(sA + ((2 * q) * OQ + oct) * RB + cb) = (a0[q] & 0xffffu) | (a1[q] << 16);
          *reinterpret_cast<unsigned*>(sA + ((2 * q + 1) * OQ + oct) * RB + cb) = (a0[q] >> 16) | (a1[q] & 0xffff0000u);
          *reinterpret_cast<unsigned*>(sB + ((2 * q) * OQ + oct) * RB + cb) = (b0[q] & 0xffffu) | (b1[q] << 16);
          *reinterpret_cast<unsigned*>(sB + ((2 * q + 1) * OQ + oct) * RB + cb) = (b0[q] >> 16) | (b1[q] & 0xffff0000u);
        }
      } else {
        const int cb = ps * 64 + pp * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {         // channel q of the quad ; two fp32 pixels side by side
          *reinterpret_cast<uint2*>(sA + (q * OQ + oct) * RB + cb) = make_uint2(a0[q], a1[q]);
          *reinterpret_cast<uint2*>(sB + (q * OQ + oct) * RB + cb) = make_uint2(b0[q], b1[q]);
        }
      }
    }
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nt = (p_end - p_begin + BP - 1) / BP;
  if (nt > 0) {
    load_step(p_begin);
    store_step();
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) load_step(p_begin + (t + 1) * BP);
    Frag<T>::template mma<FM, FN, NP, RB>(smem + wm * 64 * RB, smem + 128 * RB + wn * 64 * RB, lane, acc);
    __syncthreads();                              // everyone is done reading the buffer
    if (t + 1 < nt) store_step();
    __syncthreads();
  }
  // stage fp32 tile [co_local][k_local] (undo the row permutation), then coalesced rows into the slab
  float* so = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int Ra = (wm * FM + i) * 16 + (lane >> 4) * 4 + r;
        const int Rb = (wn * FN + j) * 16 + (lane & 15);
        const int col = (Ra % OQ) * VEC + Ra / OQ;
        const int kl = (Rb % OQ) * VEC + Rb / OQ;
        so[col * OROW + kl] = acc[i][j][r];
      }
  __syncthreads();
  float* __restrict__ ws = a.ws + (size_t)split * a.Cout * a.Ktot;
  for (int v = tid; v < 128 * 32; v += 256) {
    const int row = v >> 5, c4 = (v & 31) * 4;
    const int co = tile_co * 128 + row, k = tile_k * 128 + c4;
    if (co < a.Cout && k < a.Ktot)   // Ktot is a multiple of 8, so a float4 never straddles the edge
      *reinterpret_cast<float4*>(ws + (size_t)co * a.Ktot + k) = *reinterpret_cast<const float4*>(so + row * OROW + c4);
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16 production kernel: LDS-DMA + hardware transpose reads.
// Operand tiles are DMA'ed in their natural HBM order [pixel][128 channels] (256-byte rows, 4 pixel rows per 1 KiB chunk);
// MFMA fragments need 8 consecutive PIXELS per channel, which ds_read_b64_tr_b16 delivers for free: a 16-lane group reads a
// 4(pixel) x 16(channel) block and lane i receives column i (verified on MI355X with a probe kernel in round 1).  Two such reads make one
// 16x16x32 fragment.  Bank conflicts between the 4 pixel rows of a block (256 B apart = same banks) are removed by a
// source-side XOR of the 16-byte column index with 2*(pixel & 7).  No ds_write, no VGPR staging, one barrier per 64-pixel step.
// ------------------------------------------------------------------------------------------------
// q = n / d, r = n % d for 0 <= n < 2^24 via one float multiply and a +-1 fix-up (an integer divide costs ~35 VALU ops)
__device__ __forceinline__ void fast_divmod(int n, int d, float inv, int& q, int& r) {
  q = (int)((float)n * inv);
  r = n - q * d;
  const int lt = r < 0;       q -= lt; r += lt ? d : 0;       // predicated (v_cndmask), no divergent branches
  const int ge = r >= d;      q += ge; r -= ge ? d : 0;
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// Transpose read as inline asm (see wgrad_stream.hip): with the builtin, the compiler -- which cannot tell the ring slots apart --
// puts s_waitcnt vmcnt(0) in front of every LDS read that follows an LDS-DMA, so the fill of tile k+1 never overlapped the MFMAs of
// tile k inside a block.  The asm read is invisible to that hazard pass; the kernels order DMA and reads themselves (barriers,
// counted vmcnt) and wait for the reads with wait_lds_tr<N>(), whose "+v" operands make the MFMAs depend on the wait.
template <int OFF> __device__ __forceinline__ s16x4_t lds_tr16_asm(unsigned addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
template <int N> __device__ __forceinline__ void wait_lds_tr(bf16x8_t& a0) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a0) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lds_tr(bf16x8_t& a0, bf16x8_t& a1, bf16x8_t& b0) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a0), "+v"(a1), "+v"(b0) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lds_tr(bf16x8_t& a0, bf16x8_t& a1, bf16x8_t& a2, bf16x8_t& a3, bf16x8_t& b0) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0) : "n"(N) : "memory");
}

template <int BP, int STAGES, bool SAME>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(WgradArgs a, unsigned dy_bytes, unsigned x_bytes) {
  constexpr int NJ = BP / 16;            // chunks (of 4 pixel rows) per operand per wave per step
  constexpr int GD = 2 * NJ;             // LDS-DMA instructions per wave per step
  constexpr int TILE = BP * 256;         // bytes per operand tile
  constexpr int OROW = 132;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.blocks_total) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int split = logical / a.tiles_ck;
  const int tck = logical - split * a.tiles_ck;
  const int tile_co = tck / a.tiles_k, tile_k = tck - tile_co * a.tiles_k;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, x_bytes, 0x00020000);

  // DMA role of this lane: wave w fills chunks w, w+4, w+8, w+12 (4 pixel rows each); inside a chunk the lane fills row
  // r = lane>>4, 16-byte slot q = lane&15, with the data of logical column q ^ 2*(pixel&7); pixel&7 = r + 4*(w&1) for all its chunks
  const int r = lane >> 4, q = lane & 15;
  const int lcol = q ^ (2 * (r + 4 * (wave & 1)));
  const int co0 = tile_co * 128 + lcol * 8;
  const bool a_ok = co0 < a.Cout;
  const int kcol0 = tile_k * 128 + lcol * 8;
  const bool b_ok = kcol0 < a.Ktot;
  int dh, dw, ci;
  {
    const int kk = b_ok ? kcol0 : 0;
    const int tap = kk / a.Cin;
    ci = kk - tap * a.Cin;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    dh = kh * a.dil - a.pad; dw = kw * a.dil - a.pad;
  }
  const int p_begin = split * a.pix_per_split;
  const int p_end = min(a.M, p_begin + a.pix_per_split);
  const int HWo = a.Hout * a.Wout;
  const float inv_hw = 1.0f / (float)HWo, inv_w = 1.0f / (float)a.Wout;
  // all pixel indices are < 2^24 (checked by the host), so 24-bit multiplies (full rate) address both operands
  const unsigned ldy2 = (unsigned)a.dy_ldc * 2u, lx2 = (unsigned)a.x_ldc * 2u;
  const unsigned lane_a = (unsigned)co0 * 2u;
  // SAME (stride 1, equal input/output size): the source pixel of output pixel m under tap (dh,dw) is simply m + dh*W + dw
  const int lane_b = SAME ? ((dh * a.Win + dw) * a.x_ldc + ci) * 2 : ci * 2;
  const bool taps = a.KH * a.KW > 1 || a.pad != 0;        // 1x1 / pad 0: every source pixel is inside the image

  auto issue = [&](int m0, int buf) {
    unsigned char* sA = smem + buf * 2 * TILE;
    unsigned char* sB = sA + TILE;
    int m = m0 + 4 * wave + r;                            // this lane's pixel in chunk j = 0; chunk j adds 16*j
    int img = 0, ho = 0, wo = 0;
    if (!SAME || taps) {                                  // (uniform) one reciprocal divmod per step, then +16 increments
      int rem;
      fast_divmod(m, HWo, inv_hw, img, rem);
      fast_divmod(rem, a.Wout, inv_w, ho, wo);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int chunk = wave + 4 * j;
      const bool pv = m < p_end;
      const unsigned offa = (pv & a_ok) ? __umul24((unsigned)m, ldy2) + lane_a : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(sA + chunk * 1024), 16, offa, 0, 0, 0);
      unsigned offb;
      bool ok = pv & b_ok;
      if (SAME) {
        if (taps) ok = ok & ((unsigned)(ho + dh) < (unsigned)a.Hin) & ((unsigned)(wo + dw) < (unsigned)a.Win);
        offb = __umul24((unsigned)m, lx2) + (unsigned)lane_b;
      } else {
        const int hi = ho * a.stride + dh, wi = wo * a.stride + dw;
        ok = ok & ((unsigned)hi < (unsigned)a.Hin) & ((unsigned)wi < (unsigned)a.Win);
        offb = __umul24((unsigned)((img * a.Hin + hi) * a.Win + wi), lx2) + (unsigned)lane_b;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(sB + chunk * 1024), 16, ok ? offb : OOB, 0, 0, 0);
      if (j + 1 < NJ) {
        m += 16;
        if (!SAME || taps) {                              // predicated wrap of (wo, ho, img); Wout >= 8 so two wraps cover +16
          wo += 16;
          int c = wo >= a.Wout; wo -= c ? a.Wout : 0; ho += c;
          c = wo >= a.Wout;     wo -= c ? a.Wout : 0; ho += c;
          c = ho >= a.Hout;     ho -= c ? a.Hout : 0; img += c;
        }
      }
    }
  };

  // fragment read offsets of this lane inside an operand tile (k-step ks adds ks*32 pixel rows, fragment F adds 32 bytes of columns)
  const int t = lane & 15, kq = lane >> 4;
  // K slot (kq, half, i) of the MFMA <-> pixel row kq*4 + i + 16*half of the 32-row k-step (any bijection works: both operands use
  // it).  Lanes 0-31 (one LDS service group) then read 8 consecutive rows = 8 distinct swizzle classes = all 64 banks; with
  // rows kq*8 + i every transpose read was a 2-way conflict (rocprofv3: SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE).
  const int prow = kq * 4 + (t >> 2);                       // pixel row of the first transpose read (second: +16)
  const int sub = (t & 1) * 8;                              // 8-byte half of the 16-byte column
  const int qlo = (t & 3) >> 1;                             // which 16-byte column of the fragment's pair
  const int g0 = 2 * (prow & 7);                            // swizzle of the two reads (same pixel & 7)
  auto frag = [&](const unsigned char* tile, int ks, int F) -> bf16x8_t {
    const int row0 = ks * 32 + prow;
    const int c = 2 * F + qlo;
    const unsigned ad = lds_addr(tile) + (unsigned)(row0 * 256 + ((c ^ g0) << 4) + sub);      // the row 16 further down has the same swizzle
    const s16x4_t lo = lds_tr16_asm<0>(ad);
    const s16x4_t hi = lds_tr16_asm<16 * 256>(ad);
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nt = (p_end - p_begin + BP - 1) / BP;
  auto compute = [&](int slot) {
    const unsigned char* sA = smem + slot * 2 * TILE;
    const unsigned char* sB = sA + TILE;
#pragma unroll
    for (int ks = 0; ks < BP / 32; ++ks) {
      bf16x8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag(sA, ks, wm * 4 + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = frag(sB, ks, wn * 4 + j);
      // the 16 reads return in order: column j of the 4x4 fragment grid starts as soon as fb[j] is in
      wait_lds_tr<6>(fa[0], fa[1], fa[2], fa[3], fb[0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[0], acc[i][0], 0, 0, 0);
      wait_lds_tr<4>(fb[1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[1], acc[i][1], 0, 0, 0);
      wait_lds_tr<2>(fb[2]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[2], acc[i][2], 0, 0, 0);
      wait_lds_tr<0>(fb[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[3], acc[i][3], 0, 0, 0);
    }
  };
  if (STAGES == 2) {
    if (nt > 0) issue(p_begin, 0);
    __syncthreads();
    for (int st = 0; st < nt; ++st) {
      const int cur = st & 1;
      if (st + 1 < nt) issue(p_begin + (st + 1) * BP, cur ^ 1);
      compute(cur);
      __syncthreads();
    }
  } else {       // STAGES-deep DMA ring, counted vmcnt + raw barrier (see conv_glds_kernel)
    int issued = 0;
    for (; issued < STAGES - 1 && issued < nt; ++issued) issue(p_begin + issued * BP, issued);
    int slot = 0, islot = issued % STAGES;
    for (int st = 0; st < nt; ++st) {
      const int newer = issued - 1 - st;
      if (newer >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GD * (STAGES - 2)) : "memory");
      else if (newer == 1 && STAGES > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (issued < nt) {
        issue(p_begin + issued * BP, islot);
        ++issued;
        islot = islot + 1 == STAGES ? 0 : islot + 1;
      }
      compute(slot);
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    __syncthreads();
  }
  // fp32 tile -> LDS -> coalesced rows of the split's slab
  float* so = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        so[((wm * 4 + i) * 16 + (lane >> 4) * 4 + rr) * OROW + (wn * 4 + j) * 16 + (lane & 15)] = acc[i][j][rr];
  __syncthreads();
  float* __restrict__ ws = a.ws + (size_t)split * a.Cout * a.Ktot;
  for (int v = tid; v < 128 * 32; v += 256) {
    const int row = v >> 5, c4 = (v & 31) * 4;
    const int co = tile_co * 128 + row, k = tile_k * 128 + c4;
    if (co < a.Cout && k < a.Ktot)
      *reinterpret_cast<float4*>(ws + (size_t)co * a.Ktot + k) = *reinterpret_cast<const float4*>(so + row * OROW + c4);
  }
}

// slabs -> OIHW fp32 gradient (real Cin, i.e. without channel padding).
// One block per (co, chunk of 64 input channels): slab rows [tap][ci] are read coalesced along ci and summed over the
// splits, transposed through LDS, and written as the contiguous OIHW run [ci0..ci0+63][tap].
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int Cout_pad,
                                                           int Cin_real, int Cin_pad, int KK, int Ktot, int accumulate) {
  extern __shared__ float tile[];                       // [KK][65]
  const int co = blockIdx.x, ci0 = blockIdx.y * 64;
  const int nci = min(64, Cin_real - ci0);
  const size_t slab = (size_t)Cout_pad * Ktot;
  const float* row = ws + (size_t)co * Ktot;
  for (int i = threadIdx.x; i < KK * 64; i += 256) {
    const int t = i >> 6, c = i & 63;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // 4 independent chains: the loads of 4 splits are in flight together
    if (c < nci) {
      const float* p = row + t * Cin_pad + ci0 + c;
      int sp = 0;
      for (; sp + 4 <= splits; sp += 4) {
        s0 += p[(size_t)sp * slab]; s1 += p[(size_t)(sp + 1) * slab]; s2 += p[(size_t)(sp + 2) * slab]; s3 += p[(size_t)(sp + 3) * slab];
      }
      for (; sp < splits; ++sp) s0 += p[(size_t)sp * slab];
    }
    tile[t * 65 + c] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  float* out = dw + ((size_t)co * Cin_real + ci0) * KK;
  for (int i = threadIdx.x; i < nci * KK; i += 256) {
    const int c = i / KK, t = i - c * KK;
    const float v = tile[t * 65 + c];
    out[i] = accumulate ? out[i] + v : v;
  }
}

// Same reduction with the loads spread out: thread (c = tid & 63, q = tid >> 6) sums the splits s = q, q+4, ... of all KK taps of
// input channel c (KK independent loads per split, several splits unrolled), so a block has ~4*KK*unroll loads in flight per
// thread group instead of four dependent chains; the four partial sums meet in LDS.  (The chained version was latency-bound:
// 25 us per layer, 1.9 ms per YOLOv3 step.)
template <int KK>
__global__ __launch_bounds__(256) void wgrad_reduce_kk_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int Cout_pad,
                                                              int Cin_real, int Cin_pad, int Ktot, int accumulate) {
  __shared__ float tile[4][KK][65];
  const int co = blockIdx.x, ci0 = blockIdx.y * 64;
  const int nci = min(64, Cin_real - ci0);
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t slab = (size_t)Cout_pad * Ktot;
  float acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = 0.f;
  if (c < nci) {
    const float* p = ws + (size_t)co * Ktot + ci0 + c;
#pragma unroll 4
    for (int sp = q; sp < splits; sp += 4) {
      const float* ps = p + (size_t)sp * slab;
#pragma unroll
      for (int t = 0; t < KK; ++t) acc[t] += __builtin_nontemporal_load(ps + t * Cin_pad);   // the slabs' only reader (elementwise.hip: ld_stream; 13.11 / 13.09 -> 13.05 / 13.07 ms)
    }
  }
#pragma unroll
  for (int t = 0; t < KK; ++t) tile[q][t][c] = acc[t];
  __syncthreads();
  float* out = dw + ((size_t)co * Cin_real + ci0) * KK;
  for (int i = threadIdx.x; i < nci * KK; i += 256) {
    const int cc = i / KK, t = i - cc * KK;
    const float v = (tile[0][t][cc] + tile[1][t][cc]) + (tile[2][t][cc] + tile[3][t][cc]);
    out[i] = accumulate ? out[i] + v : v;
  }
}

// Slab reduce for SMALL layers (Cout * ceil(Cin/64) < 128 blocks in the kernel above: RektNet's 16..64-channel layers took 13-25 us
// there, most of it idle lanes and serial split loops).  One thread per slab element k (coalesced), 16 split groups per block
// with 8 loads in flight each, fixed-order tree in LDS -> deterministic.  Grid (ceil(Ktot/64), Cout_real).
__global__ __launch_bounds__(1024) void wgrad_reduce_flat_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int Cout_pad,
                                                                 int Cin_real, int Cin_pad, int KK, int Ktot, int accumulate) {
  __shared__ float part[16][64];
  const int co = blockIdx.y, c = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + c;
  const size_t slab = (size_t)Cout_pad * Ktot;
  float acc = 0.f;
  if (k < Ktot) {
    const float* p = ws + (size_t)co * Ktot + k;
    int sp = sg;
    for (; sp + 112 < splits; sp += 128) {                 // 8 independent loads in flight
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)(sp + 16 * u) * slab);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sp < splits; sp += 16) acc += __builtin_nontemporal_load(p + (size_t)sp * slab);
  }
  part[sg][c] = acc;
  __syncthreads();
  if (sg == 0 && k < Ktot) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[u][c];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int u = 0; u < w; ++u) v[u] += v[u + w];
    const int t = k / Cin_pad, ci = k - t * Cin_pad;
    if (ci < Cin_real) {
      float* out = dw + ((size_t)co * Cin_real + ci) * KK + t;
      *out = accumulate ? *out + v[0] : v[0];
    }
  }
}

// sums the fp32 slabs ws[splits][Cout_pad][KK*Cin_pad] into the OIHW gradient
static int launch_wgrad_reduce(const float* ws, float* dw_oihw, int splits, int Cout_pad, int Cout_real, int Cin_pad, int Cin_real, int KK,
                               int accumulate, hipStream_t st) {
  const int Ktot = KK * Cin_pad;
  if (Cout_real * cdiv(Cin_real, 64) < 128) {
    MDCV_LAUNCH(wgrad_reduce_flat_kernel, dim3((unsigned)cdiv(Ktot, 64), (unsigned)Cout_real), dim3(1024), 0, st, ws, dw_oihw, splits,
                       Cout_pad, Cin_real, Cin_pad, KK, Ktot, accumulate);
  } else {
    const dim3 rgrid((unsigned)Cout_real, (unsigned)cdiv(Cin_real, 64));
    if (KK == 9) MDCV_LAUNCH(wgrad_reduce_kk_kernel<9>, rgrid, dim3(256), 0, st, ws, dw_oihw, splits, Cout_pad, Cin_real, Cin_pad, Ktot, accumulate);
    else if (KK == 1) MDCV_LAUNCH(wgrad_reduce_kk_kernel<1>, rgrid, dim3(256), 0, st, ws, dw_oihw, splits, Cout_pad, Cin_real, Cin_pad, Ktot, accumulate);
    else MDCV_LAUNCH(wgrad_reduce_kernel, rgrid, dim3(256), KK * 65 * 4, st, ws, dw_oihw, splits, Cout_pad, Cin_real, Cin_pad, KK, Ktot, accumulate);
  }
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// OIHW fp32 master weights -> GEMM operand layouts (T):
//   wf[n][tap][ci_pad]  (forward "B" operand, n < Cout_pad)      wd[ci][tap][co_pad]  (dgrad "B" operand, ci < Cin_pad)
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout, int Cin,
                                    int KK, int Cout_pad, int Cin_pad) {
  const int nf = Cout_pad * KK * Cin_pad;
  const int nd = wd ? Cin_pad * KK * Cout_pad : 0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nf + nd; e += gridDim.x * blockDim.x) {
    if (e < nf) {
      const int n = e / (KK * Cin_pad), rem = e - n * (KK * Cin_pad);
      const int t = rem / Cin_pad, ci = rem - t * Cin_pad;
      const float v = (n < Cout && ci < Cin) ? w[((size_t)n * Cin + ci) * KK + t] : 0.f;
      ET<T>::st(wf + e, v);
    } else {
      const int f = e - nf;
      const int ci = f / (KK * Cout_pad), rem = f - ci * (KK * Cout_pad);
      const int t = rem / Cout_pad, co = rem - t * Cout_pad;
      const float v = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * KK + t] : 0.f;
      ET<T>::st(wd + f, v);
    }
  }
}

// Narrow-output variant (Cout_pad <= 32: first layers, RektNet's 16/32-channel blocks, heads): output tile 32(co) x 128(k).
// The dY tile is [64 px][32 co] = 64-byte rows, 16 pixel rows per 1 KiB chunk (one chunk per wave), no swizzle needed (the 4 rows
// of a transpose read sit 64 B apart -> distinct banks).  Each wave owns a 32 x 32 slice: 4 MFMAs per 32-pixel k-step instead of
// 16 MFMAs on a tile that would be 75-87 % zero padding.  These layers are HBM-bound; the point is to stop wasting issue slots.
// BNA (round 5): the layer's input needs no gradient (YOLOv3's first conv), so dy = cA g + cB y + cC, g = dz act'(scale y + shift), has this
// kernel as its ONLY reader: it is formed here, in LDS, from the dz and y tiles (two DMAs instead of one; every thread transforms one 16-byte
// vector of the 64 x 32 tile per step, rounding to bf16 exactly as mdcv_bn_act_bwd_apply does -- the results are bit-identical to apply +
// this kernel), and the BatchNorm-apply pass over the largest tensor of the network (416^2 x 32 at batch 32: read 708 MB, write 354 MB, then
// read again here) never runs.  It sat at the exposed tail of the backward: apply 193 us on the main queue, then this kernel 131 us alone.
template <bool SAME, int STAGES, bool BNA = false>
__global__ __launch_bounds__(256) void conv_wgrad_dma_narrow_kernel(WgradArgs a, unsigned dy_bytes, unsigned x_bytes) {
  constexpr int BP = 64, NJ = 4, GD = BNA ? 6 : 5;
  constexpr int TA = BP * 64 * (BNA ? 2 : 1), TB = BP * 256;   // bytes per operand tile (BNA: dz tile + y tile)
  constexpr int OROW = 132;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.blocks_total) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = logical / a.tiles_ck;
  const int tile_k = logical - split * a.tiles_ck;          // tiles_co == 1
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BNA ? a.y : a.dy), 0,
                                                                      BNA ? (unsigned)a.M * (unsigned)a.y_ldc * 2u : 0u, 0x00020000);

  // A (dY) DMA role: chunk = wave; lane fills pixel row ra = lane>>2, 16-byte slot lane&3; rows with bit 2 set hold their two
  // 32-byte halves swapped, so the 8 consecutive rows one LDS service group reads (64-byte rows: 4 rows per bank period) hit all banks
  const int ra = lane >> 2;
  const int coA = ((lane & 3) ^ (2 * ((ra >> 2) & 1))) * 8;
  const bool a_ok = coA < a.Cout;
  // B (X) DMA role: as in the wide kernel
  const int r = lane >> 4, q = lane & 15;
  const int lcol = q ^ (2 * (r + 4 * (wave & 1)));
  const int kcol0 = tile_k * 128 + lcol * 8;
  const bool b_ok = kcol0 < a.Ktot;
  int dh, dw, ci;
  {
    const int kk = b_ok ? kcol0 : 0;
    const int tap = kk / a.Cin;
    ci = kk - tap * a.Cin;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    dh = kh * a.dil - a.pad; dw = kw * a.dil - a.pad;
  }
  const int p_begin = split * a.pix_per_split;
  const int p_end = min(a.M, p_begin + a.pix_per_split);
  const int HWo = a.Hout * a.Wout;
  const float inv_hw = 1.0f / (float)HWo, inv_w = 1.0f / (float)a.Wout;
  const unsigned ldy2 = (unsigned)a.dy_ldc * 2u, lx2 = (unsigned)a.x_ldc * 2u;
  const int lane_b = SAME ? ((dh * a.Win + dw) * a.x_ldc + ci) * 2 : ci * 2;
  const bool taps = a.KH * a.KW > 1 || a.pad != 0;

  auto issue = [&](int m0, int buf) {
    unsigned char* sA = smem + buf * (TA + TB);
    unsigned char* sB = sA + TA;
    {
      const int m = m0 + 16 * wave + ra;
      const unsigned offa = (m < p_end && a_ok) ? __umul24((unsigned)m, ldy2) + (unsigned)coA * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(sA + wave * 1024), 16, offa, 0, 0, 0);
      if constexpr (BNA) {
        const unsigned offy = (m < p_end && a_ok) ? __umul24((unsigned)m, (unsigned)a.y_ldc * 2u) + (unsigned)coA * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_void_t*)(sA + BP * 64 + wave * 1024), 16, offy, 0, 0, 0);
      }
    }
    int m = m0 + 4 * wave + r;
    int img = 0, ho = 0, wo = 0;
    if (!SAME || taps) {
      int rem;
      fast_divmod(m, HWo, inv_hw, img, rem);
      fast_divmod(rem, a.Wout, inv_w, ho, wo);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int chunk = wave + 4 * j;
      bool ok = (m < p_end) & b_ok;
      unsigned offb;
      if (SAME) {
        if (taps) ok = ok & ((unsigned)(ho + dh) < (unsigned)a.Hin) & ((unsigned)(wo + dw) < (unsigned)a.Win);
        offb = __umul24((unsigned)m, lx2) + (unsigned)lane_b;
      } else {
        const int hi = ho * a.stride + dh, wi = wo * a.stride + dw;
        ok = ok & ((unsigned)hi < (unsigned)a.Hin) & ((unsigned)wi < (unsigned)a.Win);
        offb = __umul24((unsigned)((img * a.Hin + hi) * a.Win + wi), lx2) + (unsigned)lane_b;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(sB + chunk * 1024), 16, ok ? offb : OOB, 0, 0, 0);
      if (j + 1 < NJ) {
        m += 16;
        if (!SAME || taps) {
          wo += 16;
          int c = wo >= a.Wout; wo -= c ? a.Wout : 0; ho += c;
          c = wo >= a.Wout;     wo -= c ? a.Wout : 0; ho += c;
          c = ho >= a.Hout;     ho -= c ? a.Hout : 0; img += c;
        }
      }
    }
  };

  const int t = lane & 15, kq = lane >> 4;
  const int prow = kq * 4 + (t >> 2);                       // conflict-free K-slot <-> pixel-row mapping (see conv_wgrad_dma_kernel)
  const int sub = (t & 1) * 8, qlo = (t & 3) >> 1;
  const int g0 = 2 * (prow & 7);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto fragB = [&](const unsigned char* tile, int ks, int F) -> bf16x8_t {
    const int row0 = ks * 32 + prow, c = 2 * F + qlo;
    const unsigned ad = lds_addr(tile) + (unsigned)(row0 * 256 + ((c ^ g0) << 4) + sub);
    const s16x4_t lo = lds_tr16_asm<0>(ad);
    const s16x4_t hi = lds_tr16_asm<16 * 256>(ad);
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto fragA = [&](const unsigned char* tile, int ks, int F) -> bf16x8_t {       // 64-byte rows: F selects the 32-byte half,
    const int row0 = ks * 32 + prow;                                             // stored swapped in rows with bit 2 set
    const int col = (F ^ (kq & 1)) * 32 + (t & 3) * 8;
    const unsigned ad = lds_addr(tile) + (unsigned)(row0 * 64 + col);
    const s16x4_t lo = lds_tr16_asm<0>(ad);
    const s16x4_t hi = lds_tr16_asm<16 * 64>(ad);
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  // BNA transform role: thread tid owns the 16-byte vector at byte tid * 16 of the 64 x 64-byte tile: row tid >> 2 (the row's pixel is m0 + row),
  // physical slot tid & 3 = channels coT .. coT + 7 (the DMA's half swap for rows with bit 2 set)
  const int rowT = tid >> 2, coT = ((tid & 3) ^ (2 * ((rowT >> 2) & 1))) * 8;
  float ts1[BNA ? 8 : 1], tb1[BNA ? 8 : 1], tA[BNA ? 8 : 1], tB[BNA ? 8 : 1], tC[BNA ? 8 : 1];
  if constexpr (BNA) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = coT + e < a.creal;
      ts1[e] = ok ? a.s1[coT + e] : 0.f; tb1[e] = ok ? a.b1[coT + e] : 0.f;
      tA[e] = ok ? a.cA[coT + e] : 0.f; tB[e] = ok ? a.cB[coT + e] : 0.f; tC[e] = ok ? a.cC[coT + e] : 0.f;
    }
  }
  auto transform = [&](int slot, int m0) {                   // dz tile -> dy tile, in place (rows past the split: zeros, not cC)
    unsigned char* sA = smem + slot * (TA + TB);
    const unsigned ad = lds_addr(sA) + (unsigned)(tid * 16);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t rd, ry4;
    asm volatile("ds_read_b128 %0, %1" : "=v"(rd) : "v"(ad) : "memory");            // (asm: a plain LDS access behind an LDS-DMA gets a compiler-inserted vmcnt(0))
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(ry4) : "v"(ad) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rd), "+v"(ry4)::"memory");
    float d[8], v[8], o[8];
    ET<bf16_t>::unpack(make_uint4(rd[0], rd[1], rd[2], rd[3]), d);
    ET<bf16_t>::unpack(make_uint4(ry4[0], ry4[1], ry4[2], ry4[3]), v);
    const bool live = m0 + rowT < p_end;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = live ? mdcv_bn_bwd_dy(d[e], v[e], ts1[e], tb1[e], tA[e], tB[e], tC[e], a.act, a.slope) : 0.f;
    const uint4 qo = ET<bf16_t>::pack(o);
    const u32x4_t wo = {qo.x, qo.y, qo.z, qo.w};
    asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(wo) : "memory");
  };

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nt = (p_end - p_begin + BP - 1) / BP;
  auto compute = [&](int slot) {
    const unsigned char* sA = smem + slot * (TA + TB);
    const unsigned char* sB = sA + TA;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = fragA(sA, ks, i);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = fragB(sB, ks, wave * 2 + j);
      wait_lds_tr<2>(fa[0], fa[1], fb[0]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[0], acc[i][0], 0, 0, 0);
      wait_lds_tr<0>(fb[1]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[1], acc[i][1], 0, 0, 0);
    }
  };
  // these layers are HBM-latency-bound (tiny per-step work): keep STAGES-1 steps of DMA in flight (counted vmcnt, raw barrier)
  int issued = 0;
  for (; issued < STAGES - 1 && issued < nt; ++issued) issue(p_begin + issued * BP, issued);
  int slot = 0, islot = issued % STAGES;
  for (int st = 0; st < nt; ++st) {
    const int newer = issued - 1 - st;
    if (newer >= 3 && STAGES >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GD * 3) : "memory");
    else if (newer >= 2 && STAGES >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GD * 2) : "memory");
    else if (newer >= 1 && STAGES >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (issued < nt) {
      issue(p_begin + issued * BP, islot);
      ++issued;
      islot = islot + 1 == STAGES ? 0 : islot + 1;
    }
    if constexpr (BNA) {
      transform(slot, p_begin + st * BP);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // the dy tile is complete before any wave's transpose reads
    }
    compute(slot);
    slot = slot + 1 == STAGES ? 0 : slot + 1;
  }
  __syncthreads();
  float* so = reinterpret_cast<float*>(smem);          // [32][OROW]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        so[(i * 16 + (lane >> 4) * 4 + rr) * OROW + (wave * 2 + j) * 16 + (lane & 15)] = acc[i][j][rr];
  __syncthreads();
  float* __restrict__ ws = a.ws + (size_t)split * a.Cout * a.Ktot;
  for (int v = tid; v < 32 * 32; v += 256) {
    const int row = v >> 5, c4 = (v & 31) * 4;
    const int k = tile_k * 128 + c4;
    if (row < a.Cout && k < a.Ktot)
      *reinterpret_cast<float4*>(ws + (size_t)row * a.Ktot + k) = *reinterpret_cast<const float4*>(so + row * OROW + c4);
  }
}


template <int BP, int STAGES, bool SAME>
static int launch_wgrad_dma_t(const WgradArgs& a, unsigned grid, hipStream_t st, unsigned dyb, unsigned xb) {
  constexpr int RING = STAGES * 2 * BP * 256, EPI = 128 * 132 * 4;
  constexpr int LDS = RING > EPI ? RING : EPI;
  static DynLds dyn_lds;
  auto kern = conv_wgrad_dma_kernel<BP, STAGES, SAME>;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(kern), LDS); e != hipSuccess) return (int)e;
  MDCV_LAUNCH(kern, dim3(grid), dim3(256), LDS, st, a, dyb, xb);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}
template <bool SAME, int STAGES, bool BNA = false>
static int launch_wgrad_narrow_t(const WgradArgs& a, unsigned grid, hipStream_t st, unsigned dyb, unsigned xb) {
  constexpr int LDS = STAGES * (64 * 64 * (BNA ? 2 : 1) + 64 * 256);   // 20 (24) KiB per stage (the 32 x 132 fp32 epilogue staging fits inside)
  static DynLds dyn_lds;
  auto kern = conv_wgrad_dma_narrow_kernel<SAME, STAGES, BNA>;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(kern), LDS); e != hipSuccess) return (int)e;
  MDCV_LAUNCH(kern, dim3(grid), dim3(256), LDS, st, a, dyb, xb);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}
static int launch_wgrad_dma(const WgradArgs& a, unsigned grid, hipStream_t st, unsigned dyb, unsigned xb) {
  const bool same = a.stride == 1 && a.Hin == a.Hout && a.Win == a.Wout;
  if (a.Cout <= 32 && TUNE().wgrad_variant != 5)                       // (variant 5: the wide tile for narrow layers too, A/B)
    return same ? launch_wgrad_narrow_t<true, 4>(a, grid, st, dyb, xb) : launch_wgrad_narrow_t<false, 4>(a, grid, st, dyb, xb);
  if (TUNE().wgrad_variant == 4) return launch_wgrad_dma_t<64, 2, false>(a, grid, st, dyb, xb);        // generic address path (A/B)
  return same ? launch_wgrad_dma_t<64, 2, true>(a, grid, st, dyb, xb) : launch_wgrad_dma_t<64, 2, false>(a, grid, st, dyb, xb);
}

// all layers in one launch: blockIdx.y selects the layer descriptor, blockIdx.x grid-strides inside it
// blocks per layer of the batched pack (grid.x; blocks past a layer's tile count exit at once).  With 64, the eight 4.7 M-parameter layers
// (60 % of YOLOv3's parameters) ran on 64 blocks x 8 tiles each while every other block had long finished: 327 us for 0.5 GB.
constexpr unsigned kPackBlocks = 256;
struct PackDesc { const float* w; void* wf; void* wd; int Cout, Cin, KK, Cout_pad, Cin_pad; int pad_[3]; const float* bias; float* bias_pad; };   // 72 bytes
// Tile = 16 output channels x up to 64 input channels x all taps, read from OIHW as contiguous runs (one run per output
// channel), transposed through LDS and written as  wf[co][tap][ci .. ci+63]  (128-byte runs) and  wd[ci][tap][co .. co+15].
// (A plain gather kernel read 17x the parameter bytes: rocprofv3 FETCH_SIZE 4.2 GB for 248 MB of weights.)
// Full tiles of the layers that hold nearly all parameters (bf16, Cin and Cout multiples of 64 / 16, 3x3 or 1x1, 16-byte aligned OIHW rows):
// the tap count is a compile-time constant, so no index of the tile needs a runtime division (the generic loops below spend ~100 of them per
// thread and tile), the OIHW runs are read as float4 with a whole tile's loads in flight, and both packed forms leave as 16-byte stores.
template <int KK>
__device__ __forceinline__ void pack_tile_fast(const PackDesc& d, float* tile, int co0, int ci0) {
  constexpr int PER = 64 * KK, CS = PER + 1, Q = PER / 4, NLD = (16 * Q + 255) / 256;
  const float* __restrict__ w = d.w;
  bf16_t* __restrict__ wf = reinterpret_cast<bf16_t*>(d.wf);
  bf16_t* __restrict__ wd = reinterpret_cast<bf16_t*>(d.wd);
  float4 v4[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < 16 * Q) {
      const int co = i / Q, q = i - co * Q;
      v4[k] = *reinterpret_cast<const float4*>(w + ((size_t)(co0 + co) * d.Cin + ci0) * KK + 4 * q);
    }
  }
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < 16 * Q) {
      const int co = i / Q, q = i - co * Q;
      float* t = tile + co * CS + 4 * q;
      t[0] = v4[k].x; t[1] = v4[k].y; t[2] = v4[k].z; t[3] = v4[k].w;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * KK * 8; i += 256) {        // wf[co][tap][ci .. ci + 7]
    const int cv = i & 7, r = i >> 3;
    const int co = r / KK, t = r - co * KK;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[co * CS + (cv * 8 + e) * KK + t];
    *reinterpret_cast<uint4*>(wf + ((size_t)(co0 + co) * KK + t) * d.Cin_pad + ci0 + cv * 8) = ET<bf16_t>::pack(v);
  }
  if (wd) {
    for (int i = threadIdx.x; i < 2 * PER; i += 256) {          // wd[ci][tap][co .. co + 7]
      const int cov = i & 1, r = i >> 1;
      const int c = r / KK, t = r - c * KK;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[(cov * 8 + e) * CS + r];
      *reinterpret_cast<uint4*>(wd + ((size_t)(ci0 + c) * KK + t) * d.Cout_pad + co0 + cov * 8) = ET<bf16_t>::pack(v);
    }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackDesc* __restrict__ table) {
  extern __shared__ float tile[];
  const PackDesc d = table[blockIdx.y];
  if (blockIdx.x == 0 && d.bias)                          // the layer's fp32 bias parameter -> its padded operand buffer
    for (int i = threadIdx.x; i < d.Cout; i += 256) d.bias_pad[i] = d.bias[i];
  if constexpr (sizeof(T) == 2) {
    if ((d.KK == 9 || d.KK == 1) && d.Cin % 64 == 0 && d.Cin_pad == d.Cin && d.Cout % 16 == 0 && d.Cout_pad == d.Cout &&
        (reinterpret_cast<uintptr_t>(d.w) & 15) == 0) {   // (uniform per layer)
      const int tiles_ci = d.Cin / 64, ntiles = tiles_ci * (d.Cout / 16);
      for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int co0 = (tl / tiles_ci) * 16, ci0 = (tl % tiles_ci) * 64;
        if (d.KK == 9) pack_tile_fast<9>(d, tile, co0, ci0); else pack_tile_fast<1>(d, tile, co0, ci0);
      }
      return;
    }
  }
  const float* __restrict__ w = d.w;
  T* __restrict__ wf = reinterpret_cast<T*>(d.wf);
  T* __restrict__ wd = reinterpret_cast<T*>(d.wd);
  const int KK = d.KK;
  const int CIT = d.Cin_pad < 64 ? d.Cin_pad : 64;
  const int tiles_ci = (d.Cin_pad + CIT - 1) / CIT, tiles_co = (d.Cout_pad + 15) / 16;
  const int cstride = CIT * KK + 1;                       // +1: conflict-free column reads in the wd pass
  const int per = CIT * KK;
  for (int tl = blockIdx.x; tl < tiles_ci * tiles_co; tl += gridDim.x) {
    const int co0 = (tl / tiles_ci) * 16, ci0 = (tl % tiles_ci) * CIT;
    for (int i = threadIdx.x; i < 16 * per; i += 256) {
      const int co = i / per, rem = i - co * per;
      const int c = rem / KK, t = rem - c * KK;
      const int gco = co0 + co, gci = ci0 + c;
      tile[co * cstride + rem] = (gco < d.Cout && gci < d.Cin) ? w[((size_t)gco * d.Cin + gci) * KK + t] : 0.f;
    }
    __syncthreads();
    constexpr int VEC = ET<T>::VEC;
    if (CIT % VEC == 0) {                                   // 16-byte stores: VEC consecutive ci (wf) / co (wd) per thread
      const int cvn = CIT / VEC;
      for (int i = threadIdx.x; i < 16 * KK * cvn; i += 256) {
        const int cv = i % cvn, r = i / cvn;
        const int t = r % KK, co = r / KK;
        const int gco = co0 + co, gci = ci0 + cv * VEC;
        if (gco < d.Cout_pad && gci < d.Cin_pad) {
          float v[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = tile[co * cstride + (cv * VEC + e) * KK + t];
          *reinterpret_cast<uint4*>(wf + ((size_t)gco * KK + t) * d.Cin_pad + gci) = ET<T>::pack(v);
        }
      }
      if (wd) {
        constexpr int COV = 16 / VEC;
        for (int i = threadIdx.x; i < COV * per; i += 256) {
          const int cov = i % COV, r = i / COV;
          const int t = r % KK, c = r / KK;
          const int gco = co0 + cov * VEC, gci = ci0 + c;
          if (gco < d.Cout_pad && gci < d.Cin_pad) {
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = tile[(cov * VEC + e) * cstride + c * KK + t];
            *reinterpret_cast<uint4*>(wd + ((size_t)gci * KK + t) * d.Cout_pad + gco) = ET<T>::pack(v);
          }
        }
      }
    } else {
    for (int i = threadIdx.x; i < 16 * per; i += 256) {   // wf: c fastest
      const int c = i % CIT, r = i / CIT;
      const int t = r % KK, co = r / KK;
      const int gco = co0 + co, gci = ci0 + c;
      if (gco < d.Cout_pad && gci < d.Cin_pad) ET<T>::st(wf + ((size_t)gco * KK + t) * d.Cin_pad + gci, tile[co * cstride + c * KK + t]);
    }
    if (wd) {
      for (int i = threadIdx.x; i < 16 * per; i += 256) { // wd: co fastest
        const int co = i & 15, r = i >> 4;
        const int t = r % KK, c = r / KK;
        const int gco = co0 + co, gci = ci0 + c;
        if (gco < d.Cout_pad && gci < d.Cin_pad) ET<T>::st(wd + ((size_t)gci * KK + t) * d.Cout_pad + gco, tile[co * cstride + c * KK + t]);
      }
    }
    }
    __syncthreads();
  }
}


#endif   // MDCV_CONV_PART == 0 (weight gradients, pack)
}  // namespace

#if MDCV_CONV_PART == 1
int mdcv_cd_bf16_fwd(const ConvArgs& a, hipStream_t st, int B) { return dispatch_conv<bf16_t, 0>(a, st, B); }
#elif MDCV_CONV_PART == 2
int mdcv_cd_bf16_dgrad(const ConvArgs& a, hipStream_t st, int B) { return dispatch_conv<bf16_t, 1>(a, st, B); }
int mdcv_cd_bf16_s2(const ConvArgs& a, hipStream_t st, int B) { return dispatch_dgrad_s2<bf16_t>(a, st, B); }
int mdcv_cd_bf16_s2_all(const ConvArgs& a, hipStream_t st, int B) { return dispatch_dgrad_s2_all(a, st, B); }
#elif MDCV_CONV_PART == 3
int mdcv_cd_f32_fwd(const ConvArgs& a, hipStream_t st, int B) { return dispatch_conv<float, 0>(a, st, B); }
int mdcv_cd_f32_dgrad(const ConvArgs& a, hipStream_t st, int B) { return dispatch_conv<float, 1>(a, st, B); }
int mdcv_cd_f32_s2(const ConvArgs& a, hipStream_t st, int B) { return dispatch_dgrad_s2<float>(a, st, B); }
#else

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

static int conv2d_impl(int dtype, int mode, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc,
                       const float* bias, const void* addsrc, int add_ldc, float* stats_partial,
                       int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout,
                       int KH, int KW, int stride, int pad, int dil, const BnFuseArgs* fuse, void* stream, const EpiArgs* epi = nullptr,
                       const XAccArgs* xacc = nullptr) {
  if (!in || !w_packed || !out) return MDCV_EARG;
  if (xacc && (mode != 0 || stats_partial || fuse || epi || !xacc->acc || xacc->reps < 1 || (xacc->reps & (xacc->reps - 1)))) return MDCV_EARG;
  if (epi && (mode != 0 || stats_partial || fuse)) return MDCV_EARG;        // the inference epilogue is a forward-only, statistics-free path
  if ((Cin & 7) || (Nout & 7) || (in_ldc & 7) || (out_ldc & 7) || (addsrc && (add_ldc & 7))) return MDCV_EARG;
  if (stride != 1 && stride != 2) return MDCV_EARG;
  if (mode != 0 && mode != 1) return MDCV_EARG;
  ConvArgs a;
  a.in = in; a.w = w_packed; a.out = out; a.bias = bias; a.addsrc = addsrc; a.stats = stats_partial;
  a.in_ldc = in_ldc; a.out_ldc = out_ldc; a.add_ldc = add_ldc;
  a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Nout = Nout;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil;
  a.M = B * Hout * Wout; a.Ktot = KH * KW * Cin; a.tiles_n = 0; a.sshift = stride == 2 ? 1 : 0; a.tiles_total = 0; a.xcd_chunk = 0;
  a.ph = a.pw = a.kh0 = a.kw0 = 0; a.Hs = Hout; a.Ws = Wout; a.nkh = KH; a.nkw = KW; a.cls_split = 0;
  a.fuse = fuse ? *fuse : BnFuseArgs{};
  a.epi = epi ? *epi : EpiArgs{nullptr, 0, 0.f};
  a.xacc = xacc ? *xacc : XAccArgs{nullptr, 1};
  if (a.M <= 0) return MDCV_OK;
  hipStream_t st = (hipStream_t)stream;
  // stride-2 data gradient: 4 launches, one per output-parity class, each visiting only its live taps (no masked MACs)
  const bool small = (long long)B * Hin * Win * in_ldc * (dtype == MDCV_BF16 ? 2 : 4) < (1LL << 31) &&
                     (long long)Nout * KH * KW * Cin * (dtype == MDCV_BF16 ? 2 : 4) < (1LL << 31);
  if (mode == 1 && stride == 2 && dil == 1 && small && (TUNE().conv_variant != 0 || fuse)) {
    // 32- / 64-channel outputs (208 -> 416, 104 -> 208: HBM-bound): the shift kernel's stride-2 form, whole output rows per store (conv_shift.hip MODE 3)
    if (!bias && KH == 3 && KW == 3 && pad == 1 && Hout == 2 * Hin && Wout == 2 * Win && TUNE().conv_variant < 0 &&
        mdcv_shift_s2_dgrad_eligible(dtype, B, Hin, Win, Cin, Nout, in_ldc))
      return mdcv_shift_conv(3, in, in_ldc, w_packed, out, out_ldc, nullptr, addsrc, add_ldc, nullptr, B, Hin, Win, Cin, Nout, fuse, st, nullptr, 1);
    if (TUNE().conv_s2_allcls && dtype == MDCV_BF16 && KH == 3 && KW == 3 && pad == 1 && !(Hout & 1) && !(Wout & 1) && (Cin % 32) == 0 && TUNE().conv_deep_s2 &&
        TUNE().conv_tall_s2) {
      ConvArgs c = a;                              // every class: Hs x Ws = Hout/2 x Wout/2 positions; taps and Ktot are set per class in the kernel
      c.Hs = Hout / 2; c.Ws = Wout / 2;
      c.M = B * c.Hs * c.Ws;
      c.fuse.row_base = 0;
      const int rc = mdcv_cd_bf16_s2_all(c, st, B);
      if (rc != MDCV_EARG) return rc;              // (geometries without an all-class instantiation fall through to the four launches)
    }
    int row_base = 0;
    for (int cls = 0; cls < 4; ++cls) {
      ConvArgs c = a;
      c.ph = cls >> 1; c.pw = cls & 1;
      c.Hs = (Hout - c.ph + 1) / 2; c.Ws = (Wout - c.pw + 1) / 2;
      c.kh0 = (c.ph + pad) & 1; c.kw0 = (c.pw + pad) & 1;
      c.nkh = (KH - c.kh0 + 1) / 2; c.nkw = (KW - c.kw0 + 1) / 2;
      c.M = B * c.Hs * c.Ws;
      c.Ktot = c.nkh * c.nkw * Cin;
      if (c.M <= 0) continue;
      c.fuse.row_base = row_base;                 // (fused BatchNorm sums: one partial row per 128 pixels of each parity class)
      row_base += cdiv(c.M, 128);
      int rc;
      if (c.nkh <= 0 || c.nkw <= 0) { c.nkh = c.nkh > 0 ? c.nkh : 0; c.nkw = c.nkw > 0 ? c.nkw : 0; c.Ktot = 0; }
      if (dtype == MDCV_BF16) rc = mdcv_cd_bf16_s2(c, st, B);
      else if (dtype == MDCV_F32) rc = mdcv_cd_f32_s2(c, st, B);
      else return MDCV_EARG;
      if (rc) return rc;
    }
    return MDCV_OK;
  }
  // 3x3 / stride 1 / pad 1 on wide layers: nine shifted GEMMs over one LDS-resident activation chunk (conv_shift.hip)
  const bool shift_ok = Hin == Hout && Win == Wout && mdcv_shift_eligible(dtype, B, Hout, Wout, Cin, Nout, KH, KW, stride, pad, dil, in_ldc);
  if (shift_ok && (TUNE().conv_variant < 0 || fuse))
    return mdcv_shift_conv(mode, in, in_ldc, w_packed, out, out_ldc, bias, addsrc, add_ldc, stats_partial, B, Hout, Wout, Cin, Nout, fuse, st, epi, dil, xacc);
  if (fuse) {                                     // the fused store loop lives in the LDS-DMA kernels: never fall back to the staged ones
    if (!small) return MDCV_EARG;
    MdcvTune t2 = TUNE();
    if (t2.conv_variant >= 0 && t2.conv_variant < 6) t2.conv_variant = -1;
    const TuneScope lds_dma_only(t2);
    return dtype == MDCV_BF16 ? mdcv_cd_bf16_dgrad(a, st, B) : (dtype == MDCV_F32 ? mdcv_cd_f32_dgrad(a, st, B) : MDCV_EARG);
  }
  if (shift_ok && stats_partial) {   // forced generic kernel on a shift-eligible geometry (A/B runs): the caller sized the partial
    const int r0 = cdiv(a.M, 128), r1 = mdcv_shift_fwd_stats_rows(B, Hout, Wout, Nout, dil);   // rows for the shift kernel; zero the unused tail
    if (r1 > r0) {
      hipError_t e = hipMemsetAsync(stats_partial + (size_t)r0 * 2 * Nout, 0, (size_t)(r1 - r0) * 2 * Nout * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (dtype == MDCV_BF16) return mode == 0 ? mdcv_cd_bf16_fwd(a, st, B) : mdcv_cd_bf16_dgrad(a, st, B);
  if (dtype == MDCV_F32) return mode == 0 ? mdcv_cd_f32_fwd(a, st, B) : mdcv_cd_f32_dgrad(a, st, B);
  return MDCV_EARG;
}

int mdcv_conv2d(int dtype, int mode, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc,
                const float* bias, const void* addsrc, int add_ldc, float* stats_partial,
                int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout,
                int KH, int KW, int stride, int pad, int dil, void* stream) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
  return conv2d_impl(dtype, mode, in, in_ldc, w_packed, out, out_ldc, bias, addsrc, add_ldc, stats_partial, B, Hin, Win, Cin, Hout, Wout, Nout,
                     KH, KW, stride, pad, dil, nullptr, stream);
}

int mdcv_conv2d_stats_rows_geom(int dtype, int B, int Hout, int Wout, int Cin, int Nout, int KH, int KW, int stride, int pad, int dil, int in_ldc);

// Forward conv whose BatchNorm statistics (per output channel: sum, sum of squares) are ADDED to exact accumulators (exact_acc.h:
// [reps][3][2][Nout] 64-bit words, zero before the launch; mdcv_xstats_words) instead of written as partial rows.  Every forward kernel takes
// it (both dtypes); the consumer (mdcv_bn_act_fwd_xstats) finishes the statistics in its prologue and no finalize launch runs in between.
int mdcv_xstats_words(int reps, int C) { return reps * XACC_DIGITS * 2 * C; }
int mdcv_conv2d_xstats(int dtype, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc, const float* bias, void* xacc,
                       int reps, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride, int pad, int dil,
                       void* stream) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
  const XAccArgs x{reinterpret_cast<long long*>(xacc), reps};
  return conv2d_impl(dtype, 0, in, in_ldc, w_packed, out, out_ldc, bias, nullptr, 0, nullptr, B, Hin, Win, Cin, Hout, Wout, Nout, KH, KW,
                     stride, pad, dil, nullptr, stream, nullptr, &x);
}

// Data gradient (mode 1 of mdcv_conv2d, same geometry arguments) that ALSO writes the BatchNorm-backward partial sums of the
// layer that produced the tensor whose gradient this is:  partial[row][0][c] = sum g, partial[row][1][c] = sum g*(y - mean),
// g = dz * act'(scale*y + shift), one row per 128 output positions.  rows() returns how many rows are written for a geometry,
// or 0 when this geometry cannot take the fused path (the caller then keeps mdcv_conv2d + mdcv_bn_act_bwd_reduce).
/* Inference forward: out = act(conv(in) * scale[n] + shift[n]) (+ addsrc).  BatchNorm with running statistics (scale/shift from
 * mdcv_bn_eval_coeffs) and the activation run in the conv's store path: the raw conv output never goes to HBM. */
int mdcv_conv2d_affine_act(int dtype, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc, const float* scale,
                           const float* shift, const void* addsrc, int add_ldc, int act, float slope, int B, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Nout, int KH, int KW, int stride, int pad, int dil, void* stream) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
  if (act < 0 || act > 2) return MDCV_EARG;
  const EpiArgs e{scale, act, slope};
  return conv2d_impl(dtype, 0, in, in_ldc, w_packed, out, out_ldc, shift, addsrc, add_ldc, nullptr, B, Hin, Win, Cin, Hout, Wout, Nout,
                     KH, KW, stride, pad, dil, nullptr, stream, &e);
}

int mdcv_conv2d_dgrad_bnsums_rows(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride,
                                  int pad, int dil, int in_ldc) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
  const int es = 2;
  if (dtype != MDCV_BF16) return 0;     // production dtype only: not every fp32 tile variant carries the fused store loop
  if ((long long)B * Hin * Win * in_ldc * es >= (1LL << 31) || (long long)Nout * KH * KW * Cin * es >= (1LL << 31)) return 0;
  if (Hin == Hout && Win == Wout && mdcv_shift_eligible(dtype, B, Hout, Wout, Cin, Nout, KH, KW, stride, pad, dil, in_ldc))
    return mdcv_shift_stats_rows(B, Hout, Wout, dil, Nout);
  if (stride == 2 && dil == 1 && KH == 3 && KW == 3 && pad == 1 && Hout == 2 * Hin && Wout == 2 * Win && TUNE().conv_variant < 0 &&
      mdcv_shift_s2_dgrad_eligible(dtype, B, Hin, Win, Cin, Nout, in_ldc))
    return mdcv_shift_s2_rows(B, Hin, Win);             // the shift kernel's stride-2 form: one row per tile of 8 x 31 dY positions
  if (stride == 2 && dil == 1) {
    int rows = 0;
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      const int m = B * ((Hout - ph + 1) / 2) * ((Wout - pw + 1) / 2);
      if (m > 0) rows += cdiv(m, 128);
    }
    return rows;
  }
  return cdiv(B * Hout * Wout, 128);
}

int mdcv_conv2d_dgrad_bnsums(int dtype, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc, const void* addsrc,
                             int add_ldc, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride,
                             int pad, int dil, const void* y, int ldy, const float* scale, const float* shift, const float* mean, int act,
                             float slope, float* partial, void* stream) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
  if (!y || !scale || !shift || !mean || !partial || (ldy & 7)) return MDCV_EARG;
  if (mdcv_conv2d_dgrad_bnsums_rows(dtype, B, Hin, Win, Cin, Hout, Wout, Nout, KH, KW, stride, pad, dil, in_ldc) <= 0) return MDCV_EARG;
  BnFuseArgs f;
  f.y = y; f.scale = scale; f.shift = shift; f.mean = mean; f.partial = partial; f.ldy = ldy; f.act = act; f.row_base = 0;
  f.slope = act == 2 ? 0.f : slope;
  return conv2d_impl(dtype, 1, in, in_ldc, w_packed, out, out_ldc, nullptr, addsrc, add_ldc, nullptr, B, Hin, Win, Cin, Hout, Wout, Nout,
                     KH, KW, stride, pad, dil, &f, stream);
}

// 1 when the data gradient of this geometry runs as the stride-2 form of the 3x3 shift kernel (conv_shift.hip MODE 3: whole output rows from LDS,
// one partial row of fused sums per 8 x 31 tile): the plan fuses the BatchNorm-backward sums into it at every size.
int mdcv_conv2d_dgrad_s2_form_ok(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride,
                                 int pad, int dil, int in_ldc) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
  return dtype == MDCV_BF16 && stride == 2 && dil == 1 && KH == 3 && KW == 3 && pad == 1 && Hout == 2 * Hin && Wout == 2 * Win &&
         TUNE().conv_variant < 0 && (long long)B * Hin * Win * in_ldc * 2 < (1LL << 31) &&
         mdcv_shift_s2_dgrad_eligible(dtype, B, Hin, Win, Cin, Nout, in_ldc);
}

// number of rows of the [rows][2][Nout] BatchNorm partial-statistics buffer mdcv_conv2d writes (one per 128 output pixels)
int mdcv_conv2d_stats_rows(int M) { return cdiv(M, 128); }
// rows for a given forward geometry: the 3x3 stride-1 shift kernel walks a padded position stream and writes more rows
int mdcv_conv2d_stats_rows_geom(int dtype, int B, int Hout, int Wout, int Cin, int Nout, int KH, int KW, int stride, int pad, int dil,
                                int in_ldc) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
  if (mdcv_shift_eligible(dtype, B, Hout, Wout, Cin, Nout, KH, KW, stride, pad, dil, in_ldc)) return mdcv_shift_fwd_stats_rows(B, Hout, Wout, Nout, dil);
  return cdiv(B * Hout * Wout, 128);
}

// choose the pixel split of the weight-gradient kernel; returns the number of fp32 slabs.
// ~2 blocks per CU in flight, but never less than 4 steps (512 bf16 pixels) per split so the fp32 epilogue stays amortised.
int mdcv_conv2d_wgrad_splits(int dtype, int M, int Cout, int Ktot) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_WGRAD);
  const int bp = dtype == MDCV_BF16 ? 128 : 64;
  const int tiles = cdiv(Cout, 128) * cdiv(Ktot, 128);
  const int slots = TUNE().wgrad_slots;   // resident blocks: 2 per CU (wide tile 66 KiB; narrow tile 4 x 20 KiB ring)
  int s = slots / tiles;                  // never spill into a second, nearly empty round
  const int max_s = cdiv(M, bp * 4);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  int pps = cdiv(cdiv(M, s), bp) * bp;
  return cdiv(M, pps);
}

// The kw-shared-tile kernel (wgrad_shift.hip) is used where it measured faster than the generic one on MI355X inside the
// training step: long pixel runs per block (>= 128 steps of 64 positions, i.e. RektNet's 80x80 layers: 740 -> 680 us).
// On YOLOv3's 52x52 / 26x26 layers at batch 32 the generic kernel's larger grid wins by 5-10%; on 13x13 512->1024 the new
// kernel is faster alone (126 -> 116 us) but the step is 0.5% slower with it (one fat block per CU leaves less room for the
// main stream's kernels that run beside the weight gradients).  Variant 8 forces it wherever eligible, 9 disables it.
static bool use_wgrad_shift(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride, int pad,
                            int dil, long long dy_ldc, long long x_ldc) {
  if (TUNE().wgrad_variant == 9 || Hin != Hout || Win != Wout) return false;
  if (!mdcv_wgrad_shift_eligible(dtype, B, Hout, Wout, Cin, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc)) return false;
  if (TUNE().wgrad_variant == 8) return true;
  const long long Mq = (long long)B * (Hout + 1) * (Wout + 1);
  const int s = mdcv_wgrad_shift_splits(B, Hout, Wout, Cin, Cout);
  if (Mq / (64LL * s) >= 128) return true;
  return TUNE().wgrad_variant == 11 && mdcv_conv2d_wgrad_splits(dtype, B * Hout * Wout, Cout, KH * KW * Cin) == 1;   // A/B: also the split-less layers
}

// 16..128-channel 3x3 stride-1 layers (dilation 1 or 2): all nine taps read one activation window kept in an LDS ring
// (wgrad_stream.hip).  Variants 9 and 10 disable it.
static bool use_wgrad_stream(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride, int pad,
                             int dil, long long dy_ldc, long long x_ldc) {
  if (TUNE().wgrad_variant == 9 || TUNE().wgrad_variant == 10 || Hin != Hout || Win != Wout) return false;
  return mdcv_wgrad_stream_eligible(dtype, B, Hout, Wout, Cin, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc);
}

// 3x3 / stride-2 down-sampling layers: the input's four parity planes as one LDS ring (wgrad_stream_s2.hip).  Variants 9 and 10 disable it.
static bool use_wgrad_s2(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride, int pad,
                         int dil, long long dy_ldc, long long x_ldc) {
  if (TUNE().wgrad_variant == 9 || TUNE().wgrad_variant == 10) return false;
  return mdcv_wgrad_s2_eligible(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc);
}

// geometry-aware variant: the kernel mdcv_conv2d_wgrad will pick for this layer decides the split (use this one to size `ws`)
int mdcv_conv2d_wgrad_splits_geom(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride,
                                  int pad, int dil, int dy_ldc, int x_ldc) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_WGRAD);
  if (TUNE().wgrad_variant != 9 && TUNE().wgrad_variant != 10 && Hin == Hout && Win == Wout &&
      mdcv_wgrad_stem_eligible(dtype, B, Hout, Wout, Cin, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc))
    return mdcv_wgrad_stem_splits(B, Hout, Wout);
  if (use_wgrad_stream(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc))
    return mdcv_wgrad_stream_splits(B, Hout, Wout, Cin, Cout, dil);
  if (use_wgrad_shift(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc))
    return mdcv_wgrad_shift_splits(B, Hout, Wout, Cin, Cout);
  if (use_wgrad_s2(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc))
    return mdcv_wgrad_s2_splits(B, Hout, Wout, Cin, Cout);
  return mdcv_conv2d_wgrad_splits(dtype, B * Hout * Wout, Cout, KH * KW * Cin);
}

int mdcv_conv2d_wgrad(int dtype, const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits,
                      float* dw_oihw, int accumulate, int B, int Hin, int Win, int Cin, int Cin_real,
                      int Hout, int Wout, int Cout, int Cout_real, int KH, int KW, int stride, int pad, int dil, void* stream) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_WGRAD);
  if (!dy || !x || !ws || !dw_oihw) return MDCV_EARG;
  if ((Cin & 7) || (Cout & 7) || (dy_ldc & 7) || (x_ldc & 7) || splits < 1) return MDCV_EARG;
  if (TUNE().wgrad_variant != 9 && TUNE().wgrad_variant != 10 && Hin == Hout && Win == Wout &&
      mdcv_wgrad_stem_eligible(dtype, B, Hout, Wout, Cin, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc) &&
      mdcv_wgrad_stem_splits_ok(splits, B, Hout, Wout)) {      // 7x7 stem with the input padded to 16 channels: LDS-ring kernel
    const int rc = mdcv_wgrad_stem(dy, dy_ldc, x, x_ldc, ws, splits, B, Hout, Wout, (hipStream_t)stream);
    if (rc) return rc;
    return launch_wgrad_reduce(ws, dw_oihw, splits, Cout, Cout_real, Cin, Cin_real, 49, accumulate, (hipStream_t)stream);
  }
  if (use_wgrad_stream(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc) &&
      mdcv_wgrad_stream_splits_ok(splits, B, Hout, Wout, Cin, Cout, dil)) {
    int wrote_dw = 0;
    const int rc = mdcv_wgrad_stream(dy, dy_ldc, x, x_ldc, ws, splits, B, Hout, Wout, Cin, Cout, dil, (hipStream_t)stream, dw_oihw, Cin_real, Cout_real,
                                     accumulate, &wrote_dw);
    if (rc || wrote_dw) return rc;                           // (the slab-free form wrote the OIHW gradient itself)
    return launch_wgrad_reduce(ws, dw_oihw, splits, Cout, Cout_real, Cin, Cin_real, 9, accumulate, (hipStream_t)stream);
  }
  // 3x3 / stride 1 / pad 1 with 128-multiple channel counts: the three kw taps of a kernel row share one activation tile
  const bool shift_w = use_wgrad_shift(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc) &&
                       mdcv_wgrad_shift_splits_ok(splits, B, Hout, Wout);
  if (shift_w) {
    const int rc = mdcv_wgrad_shift(dy, dy_ldc, x, x_ldc, ws, splits, B, Hout, Wout, Cin, Cout, (hipStream_t)stream);
    if (rc) return rc;
    return launch_wgrad_reduce(ws, dw_oihw, splits, Cout, Cout_real, Cin, Cin_real, 9, accumulate, (hipStream_t)stream);
  }
  if (use_wgrad_s2(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dy_ldc, x_ldc) &&
      mdcv_wgrad_s2_splits_ok(splits, B, Hout, Wout)) {
    const int rc = mdcv_wgrad_s2(dy, dy_ldc, x, x_ldc, ws, splits, B, Hout, Wout, Cin, Cout, (hipStream_t)stream);
    if (rc) return rc;
    return launch_wgrad_reduce(ws, dw_oihw, splits, Cout, Cout_real, Cin, Cin_real, 9, accumulate, (hipStream_t)stream);
  }
  WgradArgs a;
  a.dy = dy; a.x = x; a.ws = ws; a.dy_ldc = dy_ldc; a.x_ldc = x_ldc;
  a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil;
  a.M = B * Hout * Wout; a.Ktot = KH * KW * Cin;
  const int bp = dtype == MDCV_BF16 ? 128 : 64;
  a.pix_per_split = cdiv(cdiv(a.M, splits), bp) * bp;
  if (cdiv(a.M, a.pix_per_split) != splits) return MDCV_EARG;
  const long long dyb = (long long)a.M * dy_ldc * 2, xb = (long long)B * Hin * Win * x_ldc * 2;
  const bool use_dma = dtype == MDCV_BF16 && dyb < (1LL << 31) && xb < (1LL << 31) && TUNE().conv_variant != 0 &&
                       (long long)B * Hin * Win + 256 < (1LL << 24) && a.M + 256 < (1 << 24) && Wout >= 8 && x_ldc < (1 << 23) && dy_ldc < (1 << 23);
  a.tiles_k = cdiv(a.Ktot, 128);
  a.tiles_ck = a.tiles_k * cdiv(Cout, 128);
  a.blocks_total = a.tiles_ck * splits;
  a.xcd_chunk = cdiv(a.blocks_total, 8);
  hipStream_t st = (hipStream_t)stream;
  const int lds = 256 * (64 * 4 + 16);   // 69632 B: one transposed step; the fp32 epilogue staging (67584 B) reuses it
  static DynLds dyn_lds16, dyn_lds32;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds16, reinterpret_cast<const void*>(conv_wgrad_kernel<bf16_t>), lds); e != hipSuccess) return (int)e;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds32, reinterpret_cast<const void*>(conv_wgrad_kernel<float>), lds); e != hipSuccess) return (int)e;
  const unsigned grid = (unsigned)(a.xcd_chunk * 8);
  if (use_dma) {
    const int rc = launch_wgrad_dma(a, grid, st, (unsigned)dyb, (unsigned)xb);
    if (rc) return rc;
  } else if (dtype == MDCV_BF16) MDCV_LAUNCH(conv_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, a);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(conv_wgrad_kernel<float>, dim3(grid), dim3(256), lds, st, a);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return launch_wgrad_reduce(ws, dw_oihw, splits, Cout, Cout_real, Cin, Cin_real, KH * KW, accumulate, st);
}

// ---- weight gradient of a conv -> BatchNorm -> activation layer whose INPUT needs no gradient (the first layer), straight from (dz, y): the
// BatchNorm-backward apply pass is folded into the operand load of the narrow kernel (conv_wgrad_dma_narrow_kernel BNA).  _ok() = 1 when the
// geometry takes that kernel (bf16, Cout_pad <= 32, not one of the LDS-ring forms); splits = mdcv_conv2d_wgrad_splits_geom of the same geometry.
int mdcv_conv2d_wgrad_bnapply_ok(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride,
                                 int pad, int dil, int dz_ldc, int y_ldc, int x_ldc) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_WGRAD);
  if (dtype != MDCV_BF16 || Cout > 32 || (Cin & 7) || (Cout & 7) || (dz_ldc & 7) || (y_ldc & 7) || (x_ldc & 7) || TUNE().wgrad_variant != 0) return 0;
  if (Hin == Hout && Win == Wout && mdcv_wgrad_stem_eligible(dtype, B, Hout, Wout, Cin, Cout, KH, KW, stride, pad, dil, dz_ldc, x_ldc)) return 0;
  if (use_wgrad_stream(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dz_ldc, x_ldc)) return 0;
  if (use_wgrad_shift(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dz_ldc, x_ldc)) return 0;
  const long long M = (long long)B * Hout * Wout;
  const int ldmax = dz_ldc > y_ldc ? dz_ldc : y_ldc;
  return M * ldmax * 2 < (1LL << 31) && (long long)B * Hin * Win * x_ldc * 2 < (1LL << 31) && TUNE().conv_variant != 0 &&
         (long long)B * Hin * Win + 256 < (1LL << 24) && M + 256 < (1 << 24) && Wout >= 8 && x_ldc < (1 << 23) && ldmax < (1 << 23);
}
int mdcv_conv2d_wgrad_bnapply(int dtype, const void* dz, int dz_ldc, const void* y, int y_ldc, const float* scale, const float* shift,
                              const float* cA, const float* cB, const float* cC, int act, float slope, const void* x, int x_ldc, float* ws,
                              int splits, float* dw_oihw, int accumulate, int B, int Hin, int Win, int Cin, int Cin_real, int Hout, int Wout,
                              int Cout, int Cout_real, int KH, int KW, int stride, int pad, int dil, void* stream) {
  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_WGRAD);
  if (!dz || !y || !scale || !shift || !cA || !cB || !cC || !x || !ws || !dw_oihw || splits < 1) return MDCV_EARG;
  if (!mdcv_conv2d_wgrad_bnapply_ok(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, dil, dz_ldc, y_ldc, x_ldc)) return MDCV_EARG;
  WgradArgs a;
  a.dy = dz; a.x = x; a.ws = ws; a.dy_ldc = dz_ldc; a.x_ldc = x_ldc;
  a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dil;
  a.M = B * Hout * Wout; a.Ktot = KH * KW * Cin;
  a.pix_per_split = cdiv(cdiv(a.M, splits), 128) * 128;
  if (cdiv(a.M, a.pix_per_split) != splits) return MDCV_EARG;
  a.tiles_k = cdiv(a.Ktot, 128);
  a.tiles_ck = a.tiles_k;
  a.blocks_total = a.tiles_ck * splits;
  a.xcd_chunk = cdiv(a.blocks_total, 8);
  a.y = y; a.y_ldc = y_ldc; a.act = act; a.slope = act == 2 ? 0.f : slope; a.creal = Cout_real;
  a.s1 = scale; a.b1 = shift; a.cA = cA; a.cB = cB; a.cC = cC;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(a.xcd_chunk * 8), dyb = (unsigned)((long long)a.M * dz_ldc * 2), xb = (unsigned)((long long)B * Hin * Win * x_ldc * 2);
  const bool same = stride == 1 && Hin == Hout && Win == Wout;
  const int stages = TUNE().wgrad_bna_stages;
  int rc;
  if (stages <= 2) rc = same ? launch_wgrad_narrow_t<true, 2, true>(a, grid, st, dyb, xb) : launch_wgrad_narrow_t<false, 2, true>(a, grid, st, dyb, xb);
  else if (stages == 3) rc = same ? launch_wgrad_narrow_t<true, 3, true>(a, grid, st, dyb, xb) : launch_wgrad_narrow_t<false, 3, true>(a, grid, st, dyb, xb);
  else rc = same ? launch_wgrad_narrow_t<true, 4, true>(a, grid, st, dyb, xb) : launch_wgrad_narrow_t<false, 4, true>(a, grid, st, dyb, xb);
  if (rc) return rc;
  return launch_wgrad_reduce(ws, dw_oihw, splits, Cout, Cout_real, Cin, Cin_real, KH * KW, accumulate, st);
}

int mdcv_wgrad_reduce(const float* ws, int splits, float* dw_oihw, int accumulate, int Cout_pad, int Cout, int Cin_pad, int Cin, int KK,
                      void* stream) {
  if (!ws || !dw_oihw || splits < 1 || Cout < 1 || Cin < 1 || Cout > Cout_pad || Cin > Cin_pad || KK < 1) return MDCV_EARG;
  return launch_wgrad_reduce(ws, dw_oihw, splits, Cout_pad, Cout, Cin_pad, Cin, KK, accumulate, (hipStream_t)stream);
}

int mdcv_pack_weights(int dtype, const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, int KH, int KW,
                      int Cout_pad, int Cin_pad, void* stream) {
  if (!w_oihw || !w_fwd) return MDCV_EARG;
  const int KK = KH * KW;
  const long long n = (long long)Cout_pad * KK * Cin_pad * (w_dgrad ? 2 : 1);
  const unsigned grid = (unsigned)min(cdiv(n, 256), 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDCV_BF16)
    MDCV_LAUNCH(pack_weights_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w_oihw, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, Cout, Cin, KK, Cout_pad, Cin_pad);
  else if (dtype == MDCV_F32)
    MDCV_LAUNCH(pack_weights_kernel<float>, dim3(grid), dim3(256), 0, st, w_oihw, (float*)w_fwd, (float*)w_dgrad, Cout, Cin, KK, Cout_pad, Cin_pad);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// one launch for every conv of a network: `table` = nlayers device-resident 64-byte records
//   { const float* w_oihw; void* w_fwd; void* w_dgrad (or NULL); int Cout, Cin, KH*KW, Cout_pad, Cin_pad; int reserved[3]; }
int mdcv_pack_weights_batched(int dtype, const void* table, int nlayers, int max_taps, void* stream) {
  if (!table || nlayers < 1 || max_taps < 1) return MDCV_EARG;
  hipStream_t st = (hipStream_t)stream;
  const int lds = 16 * (64 * max_taps + 1) * 4;          // 16 x (64 ci x taps + 1) floats
  if (lds > 160 * 1024) return MDCV_EARG;
  static DynLds dyn_lds16, dyn_lds32;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds16, reinterpret_cast<const void*>(pack_weights_batched_kernel<bf16_t>), lds); e != hipSuccess) return (int)e;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds32, reinterpret_cast<const void*>(pack_weights_batched_kernel<float>), lds); e != hipSuccess) return (int)e;
  if (dtype == MDCV_BF16) MDCV_LAUNCH(pack_weights_batched_kernel<bf16_t>, dim3(kPackBlocks, (unsigned)nlayers), dim3(256), lds, st, (const PackDesc*)table);
  else if (dtype == MDCV_F32) MDCV_LAUNCH(pack_weights_batched_kernel<float>, dim3(kPackBlocks, (unsigned)nlayers), dim3(256), lds, st, (const PackDesc*)table);
  else return MDCV_EARG;
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

}  // extern "C"

#endif   // MDCV_CONV_PART == 0 (host entry points)

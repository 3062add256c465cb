// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with the three kw taps of a kernel row sharing ONE activation tile.
//
// The generic weight-gradient kernel (conv_igemm.hip, conv_wgrad_dma_kernel) is bound by the global->LDS fill path: a
// 128(co) x 128(k) block fills a 16 KiB dY tile and a 16 KiB im2col tile per 64 pixels = 65 FLOP per filled byte, which holds the
// MFMA pipe at ~20 %.  The im2col tiles of the taps (kh, 0), (kh, 1), (kh, 2) are the same activation rows shifted by one pixel,
// so here a block owns co-tile x ci-tile x {3 kw taps}: per 64 positions it fills the dY tile once and ONE activation tile of 64+2
// rows, and the three taps read it at row offsets 0 / 1 / 2  (194 FLOP per filled byte).
//
// Positions run over the same padded 1-D stream as the forward shift kernel (conv_shift.hip): one shared zero column per image
// row and one shared zero row per image, p = img*(H+1)(W+1) + y*(W+1) + x.  dY'[p] is zero at junk positions and X'[p] is zero
// on the padding, so "tap (kh,kw) of position p" is simply X'[p + (kh-1)(W+1) + (kw-1)] and no per-tap masking is needed.
//
// Operand tiles stay in their natural [position][128 channels] order (256-byte rows, 4 rows per 1 KiB LDS-DMA chunk) and the MFMA
// fragments (8 consecutive POSITIONS per channel) come from ds_read_b64_tr_b16 transpose reads exactly as in the generic kernel;
// the source-side XOR swizzle of the 16-byte column with 2*(row & 7) stays conflict-free for a fragment that starts at any row.
// 8 waves: 2 (co halves of 64) x 4 (ci quarters of 32); accumulators 3 taps x 4 x 2 fragments = 96 VGPRs.
// Output: the same fp32 slabs ws[split][Cout][9*Cin] the generic kernel writes (summed by wgrad_reduce_kk_kernel<9>).
#include "common.h"
#include "wgrad_shift.h"

namespace {

constexpr int BP = 64;                    // positions per step
constexpr int TA = BP * 256;              // dY tile bytes (64 rows x 128 channels)
constexpr int XROWS = BP + 4;             // activation tile rows: 1 before, 64, 1 after, rounded to whole 4-row chunks
constexpr int TB = XROWS * 256;
constexpr int STAGE = TA + TB;
#ifndef MDCV_WS_NSTAGE
#define MDCV_WS_NSTAGE 2
#endif
#ifndef MDCV_WS_TARGET
#define MDCV_WS_TARGET 256
#endif
constexpr int NSTAGE = MDCV_WS_NSTAGE;    // DMA ring depth; 2 keeps two workgroups resident per CU (3 measured slower: one workgroup per CU)
constexpr int GD = 5;                     // LDS-DMA instructions per wave per step (2 dY + 3 activation chunks)
constexpr int OROW = 132;
constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

template <int OFF> __device__ __forceinline__ s16x4_t lds_tr16_asm(unsigned addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_lds_tr(bf16x8_t& a0, bf16x8_t& a1, bf16x8_t& a2, bf16x8_t& a3, bf16x8_t& b0, bf16x8_t& b1) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lds_tr(bf16x8_t& b0, bf16x8_t& b1) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N) : "memory");
}

__device__ __forceinline__ void fast_divmod(int n, int d, float inv, int& q, int& r) {   // 0 <= n < 2^24
  q = (int)((float)n * inv);
  r = n - q * d;
  const int lt = r < 0;  q -= lt; r += lt ? d : 0;
  const int ge = r >= d; q += ge; r -= ge ? d : 0;
}

__global__ __launch_bounds__(512) void wgrad3x3_shift_kernel(WgradShiftArgs a, unsigned dy_bytes, unsigned x_bytes) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.blocks_total) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // block -> (split, co tile, ci tile, kh)
  const int split = logical / a.tiles;
  int rest = logical - split * a.tiles;
  const int kh = rest % 3; rest /= 3;
  const int tile_ci = rest % a.tiles_ci, tile_co = rest / a.tiles_ci;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, x_bytes, 0x00020000);

  // DMA role: chunk c = wave + 8j holds tile rows 4c .. 4c+3; the lane fills row r = lane>>4, slot q = lane&15 with the logical
  // 16-byte column q ^ 2*(row & 7); row & 7 = r + 4*(c & 1) = r + 4*(wave & 1) for all of this wave's chunks
  const int r = lane >> 4, q = lane & 15;
  const int lcol = q ^ (2 * (r + 4 * (wave & 1)));
  const int co0 = tile_co * 128 + lcol * 8, ci0 = tile_ci * 128 + lcol * 8;
  const bool a_ok = co0 < a.Cout, b_ok = ci0 < a.Cin;
  const unsigned lane_a = (unsigned)co0 * 2u, lane_b = (unsigned)ci0 * 2u;
  const unsigned ldy2 = (unsigned)a.dy_ldc * 2u, lx2 = (unsigned)a.x_ldc * 2u;
  const float inv_sq = 1.0f / (float)a.Sq, inv_wq = 1.0f / (float)a.Wq;
  const int p_begin = split * a.pos_per_split;
  const int p_end = min(a.Mq, p_begin + a.pos_per_split);
  const int xshift = (kh - 1) * a.Wq - 1;                     // activation tile row j <-> stream position p0 + xshift + j

  // stream position -> byte offset of its pixel row (or OOB: junk column / junk row / outside the stream)
  auto pix_off = [&](int p, unsigned ld2, bool ok) -> unsigned {
    ok = ok && p >= 0 && p < a.Mq;
    const int pp = ok ? p : 0;
    int img, rem, y, x;
    fast_divmod(pp, a.Sq, inv_sq, img, rem);
    fast_divmod(rem, a.Wq, inv_wq, y, x);
    ok = ok && x < a.W && y < a.H;
    return ok ? __umul24((unsigned)((img * a.H + y) * a.W + x), ld2) : OOB;
  };
  auto issue = [&](int p0, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + TA;
#pragma unroll
    for (int j = 0; j < 2; ++j) {                             // dY: 16 chunks
      const int chunk = wave + 8 * j;
      const int p = p0 + 4 * chunk + r;
      const unsigned off = pix_off(p, ldy2, a_ok && p < p_end);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(sA + chunk * 1024), 16, (int)(off == OOB ? OOB : off + lane_a), 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {                             // activations: 17 chunks (rows -1 .. 66 around the 64 positions)
      const int chunk = wave + 8 * j;                         // every wave issues 3 (counted vmcnt); chunks past the tile fill the sink
      const bool live = chunk < XROWS / 4;
      const int p = p0 + xshift + 4 * chunk + r;
      const unsigned off = live ? pix_off(p, lx2, b_ok) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(live ? sB + chunk * 1024 : smem + NSTAGE * STAGE), 16,
                                               (int)(off == OOB ? OOB : off + lane_b), 0, 0, 0);
    }
  };

  // fragment reads (see conv_wgrad_dma_kernel): 16-lane group = 4 rows x 4 quads of a 16-channel block, lane i receives column i
  const int t = lane & 15, kq = lane >> 4;
  const int prow = kq * 4 + (t >> 2);      // K slot (kq, half, i) <-> pixel row kq*4 + i + 16*half: lanes 0-31 read 8 consecutive rows = all 64 banks
  const int sub = (t & 1) * 8;
  const int qlo = (t & 3) >> 1;
  auto frag = [&](const unsigned char* tile, int row0, int F) -> bf16x8_t {   // rows row0 .. row0+3 and row0+16 .. row0+19 of this lane group
    const int c = 2 * F + qlo;
    const int g0 = 2 * (row0 & 7);
    const unsigned ad = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)tile + (unsigned)(row0 * 256 + ((c ^ g0) << 4) + sub);
    const s16x4_t lo = lds_tr16_asm<0>(ad);                    // asm reads: see conv_igemm.hip (no compiler-forced DMA drain)
    const s16x4_t hi = lds_tr16_asm<16 * 256>(ad);
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  f32x4_t acc[3][4][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[k][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = (p_end - p_begin + BP - 1) / BP;
  auto compute = [&](int slot) {
    const unsigned char* sA = smem + slot * STAGE;
    const unsigned char* sB = sA + TA;
#pragma unroll
    for (int ks = 0; ks < BP / 32; ++ks) {
      bf16x8_t fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag(sA, ks * 32 + prow, wm * 4 + i);
      bf16x8_t fb[3][2];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[kw][j] = frag(sB, ks * 32 + prow + kw, wn * 2 + j);
      // 20 reads in order (fa x4, then the three taps): tap kw starts as soon as its two fragments are in
      wait_lds_tr<8>(fa[0], fa[1], fa[2], fa[3], fb[0][0], fb[0][1]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[0][j], acc[0][i][j], 0, 0, 0);
      wait_lds_tr<4>(fb[1][0], fb[1][1]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[1][j], acc[1][i][j], 0, 0, 0);
      wait_lds_tr<0>(fb[2][0], fb[2][1]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[2][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[2][j], acc[2][i][j], 0, 0, 0);
    }
  };
  // NSTAGE-deep ring with counted waits (never vmcnt(0) in steady state) and a raw barrier, like conv_glds_kernel
  int issued = 0;
  for (; issued < NSTAGE - 1 && issued < nt; ++issued) issue(p_begin + issued * BP, issued);
  int slot = 0, islot = issued % NSTAGE;
  for (int st = 0; st < nt; ++st) {
    const int newer = issued - 1 - st;
    if (newer >= NSTAGE - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GD * (NSTAGE - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (issued < nt) {
      issue(p_begin + issued * BP, islot);
      ++issued;
      islot = islot + 1 == NSTAGE ? 0 : islot + 1;
    }
    compute(slot);
    slot = slot + 1 == NSTAGE ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // fp32 tiles -> LDS -> coalesced rows of the split's slab, one tap at a time
  float* so = reinterpret_cast<float*>(smem);                // [128][OROW]
  float* __restrict__ ws = a.ws + (size_t)split * a.Cout * a.Ktot;
#pragma unroll 1
  for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          so[((wm * 4 + i) * 16 + (lane >> 4) * 4 + rr) * OROW + (wn * 2 + j) * 16 + (lane & 15)] =
              kw == 0 ? acc[0][i][j][rr] : (kw == 1 ? acc[1][i][j][rr] : acc[2][i][j][rr]);
    __syncthreads();
    const int kbase = (kh * 3 + kw) * a.Cin + tile_ci * 128;
    for (int v = tid; v < 128 * 32; v += 512) {
      const int row = v >> 5, c4 = (v & 31) * 4;
      const int co = tile_co * 128 + row;
      if (co < a.Cout && tile_ci * 128 + c4 < a.Cin)
        *reinterpret_cast<float4*>(ws + (size_t)co * a.Ktot + kbase + c4) = *reinterpret_cast<const float4*>(so + row * OROW + c4);
    }
    __syncthreads();
  }
}

}  // namespace

bool mdcv_wgrad_shift_eligible(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                               long long dy_ldc, long long x_ldc) {
  if (dtype != MDCV_BF16 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || dil != 1) return false;
  if ((Cin & 127) || (Cout & 127)) return false;
  if (H < 4 || W < 4) return false;
  const long long Mq = (long long)B * (H + 1) * (W + 1);
  if (Mq + 1024 >= (1LL << 24) || (long long)B * H * W >= (1LL << 24)) return false;          // 24-bit multiplies / float divmod
  if ((long long)B * H * W * dy_ldc * 2 >= (1LL << 31) || (long long)B * H * W * x_ldc * 2 >= (1LL << 31)) return false;
  if (dy_ldc >= (1 << 23) || x_ldc >= (1 << 23)) return false;
  return true;
}

// one block per CU (8 waves, ~150 VGPRs): tiles x splits ~ 256, never less than 4 steps per split
int mdcv_wgrad_shift_splits(int B, int H, int W, int Cin, int Cout) {
  const int Mq = B * (H + 1) * (W + 1);
  const int tiles = (Cout / 128) * (Cin / 128) * 3;
  int s = MDCV_WS_TARGET / tiles;
  const int max_s = (Mq + BP * 4 - 1) / (BP * 4);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  const int pps = ((Mq + s - 1) / s + BP - 1) / BP * BP;
  return (Mq + pps - 1) / pps;
}

bool mdcv_wgrad_shift_splits_ok(int splits, int B, int H, int W) {
  if (splits < 1) return false;
  const int Mq = B * (H + 1) * (W + 1);
  const int pps = ((Mq + splits - 1) / splits + BP - 1) / BP * BP;
  return (Mq + pps - 1) / pps == splits;
}

int mdcv_wgrad_shift(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int H, int W, int Cin, int Cout,
                     hipStream_t st) {
  WgradShiftArgs a;
  a.dy = dy; a.x = x; a.ws = ws; a.dy_ldc = dy_ldc; a.x_ldc = x_ldc;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Ktot = 9 * Cin;
  a.Wq = W + 1; a.Sq = (H + 1) * (W + 1); a.Mq = B * a.Sq;
  a.tiles_ci = Cin / 128;
  a.tiles = (Cout / 128) * a.tiles_ci * 3;
  a.pos_per_split = ((a.Mq + splits - 1) / splits + BP - 1) / BP * BP;
  if ((a.Mq + a.pos_per_split - 1) / a.pos_per_split != splits) return MDCV_EARG;
  a.blocks_total = a.tiles * splits;
  a.xcd_chunk = (a.blocks_total + 7) / 8;
  const int ring = NSTAGE * STAGE + 1024, stage_out = 128 * OROW * 4;      // ring + sink, or the fp32 epilogue staging
  const int lds = ring > stage_out ? ring : stage_out;
  static DynLds dyn_lds;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(wgrad3x3_shift_kernel), lds); e != hipSuccess) return (int)e;
  const unsigned dyb = (unsigned)((long long)B * H * W * dy_ldc * 2), xb = (unsigned)((long long)B * H * W * x_ldc * 2);
  MDCV_LAUNCH(wgrad3x3_shift_kernel, dim3((unsigned)(a.xcd_chunk * 8)), dim3(512), lds, st, a, dyb, xb);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

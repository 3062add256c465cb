// Pointwise (1x1) convolution blocks with the neighbouring BatchNorm pass folded into the operand load.
//
// The reference writes  conv -> BatchNorm -> LeakyReLU (-> shortcut add) -> conv1x1 ...  as separate modules (CVC-YOLOv3/models.py:48-72,
// :322-327).  With batch statistics the BatchNorm-apply cannot move into the conv that PRODUCES its input (the statistics are not known
// before the whole tensor exists), but it can move into the 1x1 conv that CONSUMES its output, because a 1x1 conv reads every input
// element exactly once per output-channel tile:
//
//   forward   z = act(y * scale + shift) (+ resid)   ;   y_out = z . W^T       (bn_act_fwd + mdcv_conv2d 1x1   -> ONE launch)
//
// (Round 4 also built the backward counterpart -- bn_act_bwd_apply + 1x1 data gradient in one launch -- and measured it level at 52^2 and
// slower elsewhere; round 5 removed it: the 1x1 backward is csrc/pw_bwd.hip now.)
//
// z is still written to HBM (the shortcut, routes, the weight gradient and the BatchNorm backward read it), but it is not read
// back by the 1x1 conv: a workgroup builds its [BMP pixels x K channels] operand tile ONCE in LDS (global -> registers -> transform ->
// LDS, all loads of the tile in flight together), keeps it resident, and streams only the weight tiles (LDS-DMA ring) through the K loop
// for every 128-wide slice of output channels.  Per 1x1 layer this removes one launch, one dependent launch boundary and one full read
// of the activation tensor; the 1x1 layers of YOLOv3 are HBM- / latency-bound (10 % of the FLOPs, a third of the conv time).
//
// LDS image of the resident operand: K/32 blocks of [BMP rows][64 bytes] with the same 16-byte-slot swizzle as the other conv kernels
// (slot q of row r holds logical k-vector q ^ swz(r)), so the MFMA fragment reads are the conflict-free ds_read_b128 pattern of
// conv_shift.hip.  Weight tiles [128 n][32 k] arrive lane-linear by buffer_load ... lds with the swizzle on the source address.
#include "common.h"
#include "pw_block.h"

#include <mutex>

namespace {

constexpr int NCW = 4;                   // consumer waves (MFMA): waves 0..3, one per SIMD
constexpr int NPW = 4;                   // producer waves (operand transform): waves 4..7
constexpr int NT = (NCW + NPW) * 64, NPT = NPW * 64;
constexpr int BRING = 3, LA = BRING - 1; // per-wave weight-tile ring, tiles of lookahead
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 2; }
__device__ __forceinline__ float act_fwd(float v, int act, float slope) { return act == 0 ? v : (v > 0.f ? v : v * slope); }

__device__ __forceinline__ void ld8(const float* __restrict__ p, int c0, int C, float (&o)[8], float dflt) {
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (p && c0 + e < C) ? p[c0 + e] : dflt;
}

// Persistent, wave-specialised workgroup (one per CU, eight waves).  Waves 4..7 PRODUCE the operand tile of pixel tile i+1: the global
// loads of a whole tile are in flight (second register set) while the previous tile is transformed, written to HBM (z / dy) and, in the
// swizzled image, to LDS buffer (i+1)&1.  Waves 0..3 CONSUME tile i: each owns FNW*16 output channels of the current slice, gets ITS
// weight rows either once (WRES: the layer's whole weight matrix stays in LDS for the life of the workgroup -- K*N*2 <= 64 KiB, the 52^2
// and larger layers) or through a private LDS-DMA ring (own vmcnt, no barrier in the K loop), multiplies them with the resident operand
// tile, and runs a wave-local epilogue (statistics / 16-byte stores).  The two halves meet once per pixel
// tile.  vmcnt is per wave, so the producers' long HBM loads and the consumers' counted DMA waits do not see each other -- in one
// instruction stream a wait for a young DMA would drain the older prefetch loads; with WRES the consumers' K loop has no memory wait at
// all.
// Operand = BatchNorm-apply + activation (+ residual) of y, written out as z; epilogue: bias, BatchNorm partial statistics.
template <int BMP, int FNW, int NV, bool HASR, bool WRES>
__global__ __launch_bounds__(NT) void pw_block_kernel(PwArgs a) {
  constexpr int FM = BMP / 16, FN = FNW;
  constexpr int NCH = NCW * FNW * 16;                   // output channels per slice (128 or 256)
  constexpr int WSLOT = FNW * 1024;                     // bytes of one weight tile of one wave: FNW x [16 n][32 k]
  constexpr int VPW = FNW * 2;                          // 16-byte channel vectors per staged row of one wave
  constexpr int SROWW = FNW * 32 + 16;                  // staging pitch of one wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int K = a.K, KB = K >> 5, CV = K >> 3;
  const int ABYTES = BMP * K * 2;
  const int BBASE = 2 * ABYTES;                         // [NCW][WRES ? KB : BRING][WSLOT]
  const int NSLOT = WRES ? KB : BRING;
  const int STAGE = BBASE + NCW * NSLOT * WSLOT;        // [NCW][BMP][SROWW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (a.M + BMP - 1) / BMP;
  const int nchunks = (a.N + NCH - 1) / NCH;
  const int first = blockIdx.x, stride = gridDim.x;
  if (first >= ntiles) return;

  if (wave >= NCW) {
    // =========================================================== producers
    const int ptid = tid - NCW * 64;
    const int cv = ptid % CV, prow = ptid / CV, pstep = NPT / CV;        // CV in {4 .. 128} divides 512: cv is fixed per thread
    const int c0 = cv * 8;
    const int kb = cv >> 2, kvl = cv & 3;
    const bf16_t* __restrict__ src0 = reinterpret_cast<const bf16_t*>(a.in0);
    const bf16_t* __restrict__ src1 = reinterpret_cast<const bf16_t*>(a.in1);
    bf16_t* __restrict__ tout = reinterpret_cast<bf16_t*>(a.tout);
    constexpr int UN = NV / 2;                             // 16-byte vectors per producer thread and HALF tile (NV = BMP * K / 2048 per tile)
    static_assert(NV >= 2 && NV % 2 == 0, "the producers pipeline over half tiles");
    float s1[8], b1[8];
    ld8(a.scale, c0, K, s1, 1.f); ld8(a.scale ? a.shift : nullptr, c0, K, b1, 0.f);
    const int mytiles = (ntiles - first + stride - 1) / stride, nitems = 2 * mytiles;
    // item j of this workgroup = half (j & 1) of its tile (j >> 1); loads past the last item are clamped to it (never used)
#define PW_LOAD(Q0, Q1, ITEM)                                                                                             \
  do {                                                                                                                  \
    const int j__ = (ITEM) < nitems ? (ITEM) : nitems - 1;                                                              \
    const long long p0__ = (long long)(first + (j__ >> 1) * stride) * BMP + (j__ & 1) * (UN * pstep);                   \
    _Pragma("unroll") for (int u = 0; u < UN; ++u) {       /* unconditional (clamped) loads: a branch per load would drain vmcnt */ \
      long long p = p0__ + prow + u * pstep;                                                                            \
      p = p < a.M ? p : (long long)a.M - 1;                                                                             \
      Q0[u] = *reinterpret_cast<const uint4*>(src0 + p * a.ld0 + c0);                                                   \
      if constexpr (HASR) Q1[u] = *reinterpret_cast<const uint4*>(src1 + p * a.ld1 + c0);                               \
    }                                                                                                                   \
  } while (0)
#define PW_EMIT(Q0, Q1, ITEM)                                                                                             \
  do {                                                                                                                  \
    const int t__ = (ITEM) >> 1, h__ = (ITEM) & 1;                                                                      \
    const long long p0__ = (long long)(first + t__ * stride) * BMP;                                                     \
    unsigned char* abase = smem + (t__ & 1) * ABYTES + kb * (BMP * 64);                                                 \
    _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                                                    \
      const int r = prow + (h__ * UN + u) * pstep;                                                                      \
      const long long p = p0__ + r;                                                                                     \
      uint4 o = make_uint4(0u, 0u, 0u, 0u);                  /* rows past the tensor: zeros in LDS */                    \
      if (p < a.M) {                                                                                                    \
        float v[8], w[8];                                                                                               \
        ET<bf16_t>::unpack(Q0[u], v);                                                                                   \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e] * s1[e] + b1[e], a.act, a.slope);             \
        if constexpr (HASR) {                                                                                           \
          ET<bf16_t>::unpack(Q1[u], w);                                                                                 \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) v[e] += w[e];                                                   \
        }                                                                                                               \
        o = ET<bf16_t>::pack(v);                                                                                        \
        if (tout) *reinterpret_cast<uint4*>(tout + p * a.ldt + c0) = o;                                                 \
      }                                                                                                                 \
      *reinterpret_cast<uint4*>(abase + r * 64 + ((kvl ^ swz(r)) << 4)) = o;                                            \
    }                                                                                                                   \
    if (h__) {                                                                                                          \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* this tile's LDS image is complete */                     \
      __builtin_amdgcn_s_barrier();                          /* ... and the consumers are done with the buffer written next */ \
    }                                                                                                                   \
  } while (0)
    // NS register sets rotate over the half tiles: NS - 1 half tiles of HBM loads are in flight while one is transformed (three sets of 4
    // vectors for the widest operands, five sets of 1 - 2 vectors for the narrow ones: >= 64 KiB of reads in flight per CU either way).
    // No conditional load in front of a use (at a join hipcc would wait vmcnt(0) and drain the prefetch): the loop leaves through a goto
    // behind an EMIT, and the loads past the last item are clamped repeats.  The set indices are compile-time (unrolled rotation).
    constexpr int NS = UN >= 4 ? 3 : 5;
    uint4 q0[NS][UN], q1[NS][UN];
    int it = 0;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) PW_LOAD(q0[s], q1[s], s);
    while (true) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        PW_LOAD(q0[(s + NS - 1) % NS], q1[(s + NS - 1) % NS], it + NS - 1);
        PW_EMIT(q0[s], q1[s], it);
        if (++it == nitems) goto produced;
      }
    }
  produced:
#undef PW_LOAD
#undef PW_EMIT
    __builtin_amdgcn_s_barrier();                              // the consumers' last tile
    return;
  }

  // ============================================================= consumers
  const int cw = wave;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  const int lrow = lane >> 2, kvs = (lane & 3) ^ swz(lrow);
  const unsigned bvo = (unsigned)(((cw * FNW * 16 + lrow) * K + kvs * 8) * 2);     // + j * 16 rows + slice offset + k offset in soffset
  unsigned char* const bring = smem + BBASE + cw * (NSLOT * WSLOT);
  int ikb = 0, ich = 0, islot = 0;
  // one weight tile of this wave = FNW DMAs (16 rows each); the sequence (slice, k block) repeats for every pixel tile
#define ISSUE_B()                                                                                                         \
  do {                                                                                                                  \
    const unsigned so__ = (unsigned)(ich * NCH * K * 2 + ikb * 64);                                                     \
    _Pragma("unroll") for (int j__ = 0; j__ < FNW; ++j__)                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(bring + islot * WSLOT + j__ * 1024), 16, (int)bvo,       \
                                               (int)(so__ + (unsigned)(j__ * 16 * K * 2)), 0, 0);                       \
    if (++ikb == KB) { ikb = 0; if (++ich == nchunks) ich = 0; }                                                        \
    if (++islot == NSLOT) islot = 0;                                                                                    \
  } while (0)
  if constexpr (WRES) {
    for (int t = 0; t < KB; ++t) ISSUE_B();                    // the whole [N][K] slice of this wave, once
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
#pragma unroll
    for (int t = 0; t < LA; ++t) ISSUE_B();
  }

  const int r16 = lane & 15, q = lane >> 4;
  const int offA = r16 * 64 + ((q ^ swz(r16)) << 4);
  const int offB = r16 * 64 + ((q ^ swz(r16)) << 4);
  unsigned char* const stg = smem + STAGE + cw * (BMP * SROWW);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
  constexpr int RPP = 64 / VPW;                               // rows per pass of the wave's store loop
  constexpr int PASSES = BMP / RPP;
  const int cvv = lane % VPW, rl = lane / VPW;

  __builtin_amdgcn_s_barrier();                                // operand tile 0 is in LDS
  int rslot = 0, buf = 0;
  for (int tile = first; tile < ntiles; tile += stride, buf ^= 1) {
    const long long p0 = (long long)tile * BMP;
    const unsigned char* abuf = smem + buf * ABYTES + offA;
    for (int ch = 0; ch < nchunks; ++ch) {
      const int nw0 = ch * NCH + cw * FNW * 16;
      const int n = nw0 + cvv * 8;
      f32x4_t acc[FM][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int kb = 0; kb < KB; ++kb) {
        bf16x8_t fa[FM], fb[FN];
        const unsigned char* pa = abuf + kb * (BMP * 64);
        const unsigned char* pb;
        if constexpr (WRES) {
          pb = bring + kb * WSLOT + offB;
        } else {
          // own weight tile landed ((LA-1) newer tiles of FNW DMAs may be in flight); the previous step's MFMAs are issued, i.e. its
          // fragment reads have returned, before the DMA below refills that slot
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(acc[i][j]));
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((LA - 1) * FNW) : "memory");
          pb = bring + rslot * WSLOT + offB;
          if (++rslot == BRING) rslot = 0;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(pb + j * 1024);
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(pa + i * 1024);
        if constexpr (!WRES) ISSUE_B();
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }

      // ---------------- wave-local epilogue: BMP rows x FNW*16 channels ----------------
      if (a.bias) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int nn = nw0 + j * 16 + (lane & 15);
          const float bv = nn < a.N ? a.bias[nn] : 0.f;
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[i][j][rr] += bv;
        }
      }
      {
        if (a.stats || a.xacc.acc) {
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            float s = 0.f, qq = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const bool live = p0 + i * 16 + (lane >> 4) * 4 + rr < a.M;
                const float v = live ? acc[i][j][rr] : 0.f;
                s += v; qq += v * v;
              }
            s += __shfl_xor(s, 16, 64); qq += __shfl_xor(qq, 16, 64);
            s += __shfl_xor(s, 32, 64); qq += __shfl_xor(qq, 32, 64);
            const int nn = nw0 + j * 16 + lane;
            if (lane < 16 && nn < a.N) {
              if (a.xacc.acc) {                             // fire-and-forget exact accumulation (exact_acc.h): no rows, no finalize launch
                long long* xp = a.xacc.acc + (size_t)(tile & (a.xacc.reps - 1)) * (XACC_DIGITS * 2) * a.N + nn;
                xacc_add(xp, 2 * (size_t)a.N, s);
                xacc_add(xp + a.N, 2 * (size_t)a.N, qq);
              } else {
                a.stats[((size_t)tile * 2 + 0) * a.N + nn] = s;
                a.stats[((size_t)tile * 2 + 1) * a.N + nn] = qq;
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int rr = 0; rr < 4; rr += 2) {
            const int row = i * 16 + (lane >> 4) * 4 + rr, col = j * 16 + (lane & 15);
            const unsigned pk = pack_bf16x2(acc[i][j][rr], acc[i][j][rr + 1]);
            reinterpret_cast<bf16_t*>(stg + row * SROWW)[col] = (bf16_t)(pk & 0xffffu);
            reinterpret_cast<bf16_t*>(stg + (row + 1) * SROWW)[col] = (bf16_t)(pk >> 16);
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the wave's own staging writes (LDS serves one wave's requests in order)
#pragma unroll
      for (int u = 0; u < PASSES; ++u) {
        const int row = u * RPP + rl;
        const long long p = p0 + row;
        if (p < a.M && n < a.N)
          *reinterpret_cast<uint4*>(out + p * a.out_ldc + n) = *reinterpret_cast<const uint4*>(stg + row * SROWW + cvv * 16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // staging reads done before the next slice's writes
    }
    __builtin_amdgcn_s_barrier();                               // next operand tile ready; this one may be overwritten
  }
  if constexpr (!WRES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's run-ahead DMAs land before the workgroup's LDS is released
#undef ISSUE_B
}

template <int BMP, int FNW, int NV, bool HASR, bool WRES>
int launch_pw3(PwArgs a, hipStream_t st) {
  const int tiles_m = (int)(((long long)a.M + BMP - 1) / BMP);
  const int nslot = WRES ? a.K / 32 : BRING;
  const int lds = 2 * BMP * a.K * 2 + NCW * nslot * FNW * 1024 + NCW * BMP * (FNW * 32 + 16);
  // per DEVICE: hipFuncSetAttribute applies to the function on the current device, and the CU count is that device's (a process
  // that used a second GPU never got the attribute there)
  static int attr_lds[64] = {};
  static int ncus[64] = {};
  static std::mutex mu;
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MDCV_EARG;
  if (lds > 160 * 1024) return MDCV_EARG;
  auto kern = pw_block_kernel<BMP, FNW, NV, HASR, WRES>;
  int ncu;
  {
    std::lock_guard<std::mutex> g(mu);
    if (!ncus[dev]) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) return MDCV_EARG;
      ncus[dev] = n;
    }
    ncu = ncus[dev];
    if (lds > attr_lds[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
      attr_lds[dev] = lds;
    }
  }
  const int grid = tiles_m < ncu ? tiles_m : ncu;               // persistent: one workgroup per CU walks tiles first, first + grid, ...
  MDCV_LAUNCH(kern, dim3((unsigned)grid), dim3(NT), lds, st, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}


template <int BMP, int FNW, int NV, bool HASR>
int launch_pw2(PwArgs a, hipStream_t st) {
  const int nchunks = (a.N + NCW * FNW * 16 - 1) / (NCW * FNW * 16);
  const int lds_res = 2 * BMP * a.K * 2 + NCW * (a.K / 32) * FNW * 1024 + NCW * BMP * (FNW * 32 + 16);
  if (TUNE().pw_wres && nchunks == 1 && lds_res <= 160 * 1024) return launch_pw3<BMP, FNW, NV, HASR, true>(a, st);
  return launch_pw3<BMP, FNW, NV, HASR, false>(a, st);
}

template <int BMP, int FNW>
int launch_pw(PwArgs a, hipStream_t st) {
  const int nv = BMP * a.K / 2048;                               // 16-byte vectors of a tile per producer thread
  if (nv == 8) return a.in1 ? launch_pw2<BMP, FNW, 8, true>(a, st) : launch_pw2<BMP, FNW, 8, false>(a, st);
  if constexpr (BMP == 64) {                                       // (K = 128 / 64 operands come with 64-pixel tiles only)
    if (nv == 4) return a.in1 ? launch_pw2<BMP, FNW, 4, true>(a, st) : launch_pw2<BMP, FNW, 4, false>(a, st);
    if (nv == 2) return a.in1 ? launch_pw2<BMP, FNW, 2, true>(a, st) : launch_pw2<BMP, FNW, 2, false>(a, st);
  }
  return MDCV_EARG;
}

int dispatch_pw(PwArgs a, hipStream_t st) {
  const int bmp = mdcv_pw_tile_rows(a.K);
  const bool wide = a.N > 128;                                  // 256-channel slices (64 per consumer wave) when there are that many
  if (bmp == 64) return wide ? launch_pw<64, 4>(a, st) : launch_pw<64, 2>(a, st);
  if (bmp == 32) return wide ? launch_pw<32, 4>(a, st) : launch_pw<32, 2>(a, st);
  return wide ? launch_pw<16, 4>(a, st) : launch_pw<16, 2>(a, st);
}

}  // namespace

// pixels per workgroup tile: the resident operand tile is kept at <= 32 KiB so that two workgroups share a CU
int mdcv_pw_tile_rows(int K) {
  if (TUNE().pw_bmp) return TUNE().pw_bmp;
  if (K <= 256) return 64;
  if (K <= 512) return 32;
  return 16;
}

bool mdcv_pw_eligible(int dtype, long long M, int K, int N, int ld0, int ld1, int ldt, int out_ldc) {
  if (dtype != MDCV_BF16) return false;
  if (K < 32 || K > 1024 || (K & 31) || (256 % (K >> 3)) != 0 || K < 64) return false;     // K/8 in {8 .. 128} must divide the 256 producer threads
  if (N < 8 || (N & 7)) return false;
  if ((ld0 & 7) || (ld1 & 7) || (ldt & 7) || (out_ldc & 7)) return false;
  if (M < 1 || M * (long long)((ld0 > ld1 ? ld0 : ld1) > ldt ? (ld0 > ld1 ? ld0 : ld1) : ldt) * 2 >= (1LL << 40)) return false;
  if ((long long)N * K * 2 >= (1LL << 31)) return false;
  return true;
}

extern "C" {

int mdcv_pw_rows(long long M, int K) { return (int)((M + mdcv_pw_tile_rows(K) - 1) / mdcv_pw_tile_rows(K)); }

/* forward: z = act(y * scale + shift) (+ resid) -> z_out (may be NULL) ; out = z . W^T (+ bias) ; stats: [mdcv_pw_rows][2][N] partial sums */
int mdcv_pw_conv_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift, const void* resid, int ldr, int act, float slope,
                     void* z_out, int ldz, const void* w_packed, const float* bias, void* out, int out_ldc, float* stats_partial,
                     long long M, int K, int N, void* stream) {
  if (!y || !w_packed || !out || !mdcv_pw_eligible(dtype, M, K, N, ldy, resid ? ldr : 8, z_out ? ldz : 8, out_ldc)) return MDCV_EARG;
  PwArgs a{};
  a.in0 = y; a.ld0 = ldy; a.in1 = resid; a.ld1 = ldr; a.tout = z_out; a.ldt = ldz;
  a.scale = scale; a.shift = shift; a.act = act; a.slope = act == 2 ? 0.f : slope;   // (ReLU = slope 0, as mdcv_bn_act_fwd)
  a.w = w_packed; a.w_bytes = (unsigned)((long long)N * K * 2); a.bias = bias; a.out = out; a.out_ldc = out_ldc; a.stats = stats_partial;
  a.M = (int)M; a.K = K; a.N = N; a.ysplit = 1;
  return dispatch_pw(a, (hipStream_t)stream);
}

/* the same with the output's statistics added to exact accumulators (exact_acc.h; [reps][3][2][N] 64-bit words, zero before the launch) instead
 * of written as partial rows: the consumer (mdcv_bn_act_fwd_xstats) finishes them in its prologue, no finalize launch. */
int mdcv_pw_conv_fwd_xstats(int dtype, const void* y, int ldy, const float* scale, const float* shift, const void* resid, int ldr, int act,
                            float slope, void* z_out, int ldz, const void* w_packed, const float* bias, void* out, int out_ldc, void* xacc,
                            int reps, long long M, int K, int N, void* stream) {
  if (!y || !w_packed || !out || !xacc || reps < 1 || (reps & (reps - 1)) ||
      !mdcv_pw_eligible(dtype, M, K, N, ldy, resid ? ldr : 8, z_out ? ldz : 8, out_ldc))
    return MDCV_EARG;
  PwArgs a{};
  a.in0 = y; a.ld0 = ldy; a.in1 = resid; a.ld1 = ldr; a.tout = z_out; a.ldt = ldz;
  a.scale = scale; a.shift = shift; a.act = act; a.slope = act == 2 ? 0.f : slope;
  a.w = w_packed; a.w_bytes = (unsigned)((long long)N * K * 2); a.bias = bias; a.out = out; a.out_ldc = out_ldc; a.stats = nullptr;
  a.xacc = XAccArgs{reinterpret_cast<long long*>(xacc), reps};
  a.M = (int)M; a.K = K; a.N = N; a.ysplit = 1;
  return dispatch_pw(a, (hipStream_t)stream);
}

}  // extern "C"

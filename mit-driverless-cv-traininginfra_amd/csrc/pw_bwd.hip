// Backward of a pointwise (1x1, stride 1) convolution in ONE launch.
//
// The reference's 1x1 nn.Conv2d layers (CVC-YOLOv3/models.py:59-65; 34 of YOLOv3's 75 convs) ran their backward as three launches that
// each read dy from HBM: the data gradient dx = dy . W (mdcv_conv2d mode 1), the weight gradient dW = dy^T . x (conv_wgrad_dma_kernel, fp32
// slabs) and the slab reduce.  These layers are bound by neither MFMA (10 % of the step's FLOPs) nor by their minimum HBM traffic
// (0.05 - 0.2 of peak): they are launches.  Here a workgroup owns a run of pixels (a "slab") and a 64-wide slice of input channels and,
// per 32-pixel tile, holds dy [32 x Cout] and x [32 x 64] in LDS once and uses them for BOTH products:
//
//     dx[32 px x 64 ci]   = dy . Wd^T     (A = the layer's Wd slice, resident in LDS for the life of the workgroup; B = the dy tile)
//     dW[Cout x 64 ci]   += dy^T . x      (both fragments by ds_read_b64_tr_b16 from the same tiles; accumulators live across the slab)
//
// dx leaves through the usual epilogue (+ addsrc, + the fused BatchNorm-backward sums of the layer in FRONT of this conv, bn_fuse.h), the
// slab's dW partial is written once at the end to ws[slab][Cout][Cin] (fp32) and summed in fixed slab order by the existing reduce
// kernel (bit-reproducible: no float atomics).  Everything a tile needs -- dy, x, addsrc, the y of the fused sums -- arrives through ONE
// LDS-DMA ring (buffer_load ... lds, counted vmcnt), so the only other vector-memory operations of a wave are the dx stores.
//
// LDS images: [channel block of 32][rows][64 bytes], 16-byte slot q of row r stored at q ^ ((r >> 1) & 2): conflict-free both for the
// ds_read_b128 MFMA fragments of the dx product (rows = pixels, k = channels) and for the transpose reads of the dW product
// (4 pixel rows x 16 channels per 16 lanes).
#include "common.h"
#include "bn_fuse.h"

// (at global scope: a kernel argument type with internal linkage leaves the host stub undefined)
struct PwbArgs {
  const void* dy; const void* x; const void* wd; void* dx; const void* addsrc; float* ws;
  int ldy, ldx, lddx, ldadd;
  int M, Cin, Cout;                      // Cout: channels of dy (padded, multiple of 64); Cin: channels of x / dx (multiple of 64)
  int slab_px, nslabs, slices;
  BnFuseArgs fuse;
};

namespace {

constexpr int NT = 512, NW = 8;          // eight waves
constexpr int BP = 32;                   // pixels per tile = one MFMA K step of the dW product
constexpr int NS = 64;                   // input channels per slice
constexpr int T64 = 2 * 2048 + 128;      // a [32 px][64 ch] image: two channel blocks, the second shifted by 32 banks (epilogue reads of a whole 128-byte row)
constexpr int SPITCH = 136;              // staging row pitch of the dx tile (bf16, 64 channels + 8 bytes)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;


__device__ __forceinline__ int swz(int row) { return (row >> 1) & 2; }
__device__ __forceinline__ unsigned lds_u32(const unsigned char* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ uint2 lds_rd64(unsigned addr) {
  uint2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int OFF> __device__ __forceinline__ s16x4_t lds_tr16(unsigned addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

// CO4 = Cout / 64; S = ring stages; ADD: out = product + addsrc; FUSE: BatchNorm-backward sums of the producer layer (bn_fuse.h)
template <int CO4, int S, bool ADD, bool FUSE>
__global__ __launch_bounds__(NT) void pw_bwd_kernel(PwbArgs a) {
  constexpr int KB = 2 * CO4;                            // 32-channel blocks of dy
  constexpr int NX = 1 + (ADD ? 1 : 0) + (FUSE ? 1 : 0);
  constexpr int WD_BYTES = KB * 4096;
  constexpr int STG = KB * 2048 + NX * T64;
  constexpr int STAGING = WD_BYTES + S * STG;
  constexpr int DUMP = STAGING + BP * SPITCH;
  constexpr int NCHUNK = 2 * KB + 4 * NX;                // 1 KiB DMA pieces per tile
  constexpr int NPW = (NCHUNK + NW - 1) / NW;            // ... per wave (pieces past NCHUNK are zero-fills into the dump)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  // XCD-aware order: the `slices` workgroups of a slab re-read the same dy rows, so they share an XCD (its L2)
  const int xcd = (int)(blockIdx.x & 7), seq = (int)(blockIdx.x >> 3);
  const int slab = xcd + 8 * (seq / a.slices), slice = seq % a.slices;
  if (slab >= a.nslabs) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px0 = slab * a.slab_px;
  const int px_end = min(a.M, px0 + a.slab_px);
  const int ntiles = (px_end - px0 + BP - 1) / BP;
  const int ci0 = slice * NS;

  // rows past the slab's end are past num_records: the DMA zero-fills them (no per-lane predicate)
  const __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wd), 0, (unsigned)a.Cin * (unsigned)a.Cout * 2u, 0x00020000);

  // ---- DMA role of this lane: row (lane >> 2) of a 16-row piece, physical 16-byte slot (lane & 3) = logical slot (lane & 3) ^ swz(row).
  // Piece c = wave + 8 i of a tile: 2 KB pieces of dy, then four each of x / addsrc / y-of-the-fused-sums; pieces past NCHUNK read a
  // zero-length buffer into the dump (every wave issues NPW DMAs per tile, so the counted waits are compile-time constants).
  const int lrow = lane >> 2, lslot = (lane & 3) ^ swz(lrow);
  // (scalars rs0 .. rs5 and macros, not an array / a lambda: hipcc's host pass silently drops a kernel -- no stub, no fat binary, an undefined
  // symbol at dlopen -- whose body passes an ELEMENT of an array of __amdgpu_buffer_rsrc_t to the LDS-DMA builtin)
#define PWB_SETUP(I)                                                                                                               \
  const int c##I = wave + NW * I;                          /* wave-uniform */                                                       \
  const int cc##I = c##I - 2 * KB, t##I = cc##I >> 2, b##I = (cc##I >> 1) & 1, g##I = c##I & 1;                                      \
  const bool isdy##I = c##I < 2 * KB, live##I = c##I < NCHUNK;                                                                       \
  const int k##I = isdy##I ? 0 : (t##I == 0 ? 1 : ((ADD && t##I == 1) ? 2 : 3));        /* 0 dy, 1 x, 2 addsrc, 3 fy */              \
  const void* bp##I = k##I == 0 ? a.dy : (k##I == 1 ? a.x : (k##I == 2 ? a.addsrc : a.fuse.y));                                     \
  const unsigned ld##I = (unsigned)(k##I == 0 ? a.ldy : (k##I == 1 ? a.ldx : (k##I == 2 ? a.ldadd : a.fuse.ldy)));                  \
  const unsigned pitch##I = live##I ? ld##I * 2u : 0u;                                                                              \
  const __amdgpu_buffer_rsrc_t rs##I = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(live##I ? bp##I : a.dy), 0,              \
                                                                         live##I ? (unsigned)px_end * ld##I * 2u : 0u, 0x00020000); \
  const int ch##I = isdy##I ? (c##I >> 1) * 32 : ci0 + b##I * 32;                                                                   \
  const unsigned voff##I = live##I ? (unsigned)(g##I * 16 + lrow) * pitch##I + (unsigned)(ch##I + lslot * 8) * 2u : 0u;             \
  const unsigned ldso##I = !live##I ? (unsigned)(DUMP - WD_BYTES)                                                                   \
                                    : (isdy##I ? (unsigned)((c##I >> 1) * 2048 + g##I * 1024)                                       \
                                               : (unsigned)(KB * 2048 + t##I * T64 + b##I * (2048 + 128) + g##I * 1024));           \
  const unsigned smul##I = live##I ? (unsigned)STG : 0u;
  PWB_SETUP(0) PWB_SETUP(1) PWB_SETUP(2) PWB_SETUP(3) PWB_SETUP(4) PWB_SETUP(5)
#undef PWB_SETUP
  static_assert(NPW <= 6, "six DMA roles per wave");
#define PWB_ISSUE_ONE(I, PROW_, SLOT_)                                                                                             \
  if constexpr (I < NPW)                                                                                                          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs##I, (lds_void_t*)(smem + WD_BYTES + (unsigned)(SLOT_) * smul##I + ldso##I), 16,      \
                                             (int)voff##I, (int)((PROW_) * pitch##I), 0, 0)
#define PWB_ISSUE_TILE(T_, SLOT_)                                                                                                  \
  do {                                                                                                                           \
    const unsigned prow__ = (unsigned)(px0 + (T_) * BP);                                                                         \
    PWB_ISSUE_ONE(0, prow__, SLOT_); PWB_ISSUE_ONE(1, prow__, SLOT_); PWB_ISSUE_ONE(2, prow__, SLOT_);                           \
    PWB_ISSUE_ONE(3, prow__, SLOT_); PWB_ISSUE_ONE(4, prow__, SLOT_); PWB_ISSUE_ONE(5, prow__, SLOT_);                           \
  } while (0)

  // ---- prologue: the Wd slice [64 ci rows][Cout] (KB blocks of [64][64 B]) and the first S - 1 tiles
#pragma unroll
  for (int i = 0; i < (4 * KB) / NW; ++i) {
    const int c = wave + NW * i, kb = c >> 2, g = c & 3;
    const unsigned vo = (unsigned)(ci0 + g * 16 + lrow) * (unsigned)a.Cout * 2u + (unsigned)(kb * 32 + lslot * 8) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rwd, (lds_void_t*)(smem + kb * 4096 + g * 1024), 16, (int)vo, 0, 0, 0);
  }
  int issued = 0;
  for (; issued < S - 1 && issued < ntiles; ++issued) PWB_ISSUE_TILE(issued, issued);

  // ---- fragment addresses
  const int r16 = lane & 15, q4 = lane >> 4;
  const int offF = r16 * 64 + ((q4 ^ swz(r16)) << 4);      // ds_read_b128 fragment: row r16, k slot q4 of a 32-channel block
  const int cb = wave & 3, pb = wave >> 2;                 // dx product: this wave's 16 input channels x 16 pixels of the [64 x 32] tile
  const unsigned char* const pWd = smem + (cb * 16) * 64 + offF;
  // transpose reads: lane t = lane & 15 supplies row (t >> 2) of its 4-row group, 8-byte piece (t & 3) of the 32-byte (16-channel) run
  const int trow = q4 * 4 + (r16 >> 2);
  const int tsub = (r16 & 1) * 8, tq = (r16 & 3) >> 1;
  const int wc = wave >> 1, wn = wave & 1;                 // dW product: co blocks [wc * CO4, +CO4), ci blocks wn * 2 + {0, 1}
  // epilogue role: row tid >> 4, channels (tid & 15) * 4 .. + 4 of the slice
  const int erow = tid >> 4, ecg = tid & 15;
  const int eblk = ecg >> 3, eslot = (ecg & 7) >> 1, esub = (ecg & 1) * 8;
  const int eoff64 = eblk * (2048 + 128) + erow * 64 + ((eslot ^ swz(erow & 15)) << 4) + esub;
  bf16_t* __restrict__ dxp = reinterpret_cast<bf16_t*>(a.dx);
  const unsigned stg_wr = lds_u32(smem) + (unsigned)(STAGING + (pb * 16 + r16) * SPITCH + cb * 32 + q4 * 8);

  float fs[4], fb[4], fm[4], sg[4], sx[4];
  if constexpr (FUSE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = ci0 + ecg * 4 + e;
      fs[e] = a.fuse.scale[n]; fb[e] = a.fuse.shift[n]; fm[e] = a.fuse.mean[n]; sg[e] = 0.f; sx[e] = 0.f;
    }
  }

  f32x4_t accw[CO4][2];
#pragma unroll
  for (int i = 0; i < CO4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) accw[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int slot = 0, islot = issued % S;
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed: at most the S - 2 younger tiles may be in flight (stores are younger still and are NOT counted as allowance:
    // the wait then holds whether or not store acknowledgements overtake loads)
    const int newer = issued - 1 - t;
    if (S > 2 && newer >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S > 2 ? S - 2 : 0) * NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (issued < ntiles) {
      PWB_ISSUE_TILE(issued, islot);
      ++issued;
      islot = islot + 1 == S ? 0 : islot + 1;
    }
    const unsigned char* stg = smem + WD_BYTES + slot * STG;

    // ---------------- dx tile: D[ci][px] = sum_co Wd[ci][co] dy[px][co]
    {
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const unsigned char* pB = stg + (pb * 16) * 64 + offF;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(pWd + kb * 4096);
        const bf16x8_t fbv = *reinterpret_cast<const bf16x8_t*>(pB + kb * 2048);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fbv, acc, 0, 0, 0);
      }
      // lane: pixel pb * 16 + r16, channels cb * 16 + q4 * 4 .. + 4
      uint2 pk;
      pk.x = pack_bf16x2(acc[0], acc[1]); pk.y = pack_bf16x2(acc[2], acc[3]);
      // (asm: behind an LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of a plain LDS store -- it cannot tell the staging rows from the ring)
      asm volatile("ds_write_b64 %0, %1" ::"v"(stg_wr), "v"(pk) : "memory");
    }

    // ---------------- dW partial: D[co][ci] += sum_px dy[px][co] x[px][ci]
    {
      bf16x8_t fx[2], fd[CO4];
      const unsigned xb = lds_u32(stg) + (unsigned)(KB * 2048 + wn * (2048 + 128) + trow * 64 + tsub);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned ad = xb + (unsigned)((((2 * j + tq) ^ swz(trow)) << 4));
        const s16x4_t lo = lds_tr16<0>(ad), hi = lds_tr16<1024>(ad);
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fx[j] = __builtin_bit_cast(bf16x8_t, v);
      }
      const unsigned db = lds_u32(stg) + (unsigned)(trow * 64 + tsub);
#pragma unroll
      for (int i = 0; i < CO4; ++i) {
        const int gblk = wc * CO4 + i;                       // 16-channel block of dy: 32-channel block gblk >> 1, half gblk & 1
        const unsigned ad = db + (unsigned)((gblk >> 1) * 2048) + (unsigned)((((2 * (gblk & 1) + tq) ^ swz(trow)) << 4));
        const s16x4_t lo = lds_tr16<0>(ad), hi = lds_tr16<1024>(ad);
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fd[i] = __builtin_bit_cast(bf16x8_t, v);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fx[j]));
#pragma unroll
      for (int i = 0; i < CO4; ++i) asm volatile("" : "+v"(fd[i]));
#pragma unroll
      for (int i = 0; i < CO4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accw[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[i], fx[j], accw[i][j], 0, 0, 0);
    }

    // ---------------- epilogue of the dx tile: staging -> (+ addsrc) -> HBM, fused sums
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const int p = px0 + t * BP + erow;
      // (asm reads, like the staging store: a plain LDS load behind an LDS-DMA gets a compiler-inserted s_waitcnt vmcnt(0))
      const unsigned sbase = lds_u32(stg) + (unsigned)(KB * 2048 + eoff64);
      uint2 d = lds_rd64(lds_u32(smem) + (unsigned)(STAGING + erow * SPITCH + ecg * 8)), q = d, yq = d;
      if constexpr (ADD) q = lds_rd64(sbase + T64);
      if constexpr (FUSE) yq = lds_rd64(sbase + (NX - 1) * T64);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d), "+v"(q), "+v"(yq)::"memory");
      float v[4];
      v[0] = __uint_as_float(d.x << 16); v[1] = __uint_as_float(d.x & 0xffff0000u);
      v[2] = __uint_as_float(d.y << 16); v[3] = __uint_as_float(d.y & 0xffff0000u);
      if constexpr (ADD) {
        v[0] += __uint_as_float(q.x << 16); v[1] += __uint_as_float(q.x & 0xffff0000u);
        v[2] += __uint_as_float(q.y << 16); v[3] += __uint_as_float(q.y & 0xffff0000u);
        d.x = pack_bf16x2(v[0], v[1]); d.y = pack_bf16x2(v[2], v[3]);
      }
      if (p < px_end) *reinterpret_cast<uint2*>(dxp + (size_t)p * a.lddx + ci0 + ecg * 4) = d;
      if constexpr (FUSE) {
        if (p < px_end) {
          float yv[4];
          yv[0] = __uint_as_float(yq.x << 16); yv[1] = __uint_as_float(yq.x & 0xffff0000u);
          yv[2] = __uint_as_float(yq.y << 16); yv[3] = __uint_as_float(yq.y & 0xffff0000u);
          v[0] = __uint_as_float(d.x << 16); v[1] = __uint_as_float(d.x & 0xffff0000u);       // the sums see dz as stored
          v[2] = __uint_as_float(d.y << 16); v[3] = __uint_as_float(d.y & 0xffff0000u);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pre = __builtin_fmaf(yv[e], fs[e], fb[e]);   // (the sign at the activation boundary must be the apply pass's: mdcv_bn_bwd_dy)
            const float g = (a.fuse.act != 0 && !(pre > 0.f)) ? v[e] * a.fuse.slope : v[e];
            sg[e] += g;
            sx[e] += g * (yv[e] - fm[e]);
          }
        }
      }
    }
    slot = slot + 1 == S ? 0 : slot + 1;
  }

  // ---- the slab's dW partial: ws[slab][co][Cin], this workgroup's 64 columns
  {
    float* __restrict__ wsp = a.ws + (size_t)slab * a.Cout * a.Cin + ci0 + wn * 32 + r16;
#pragma unroll
    for (int i = 0; i < CO4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int co = (wc * CO4 + i) * 16 + q4 * 4 + rr;
          wsp[(size_t)co * a.Cin + j * 16] = accw[i][j][rr];
        }
  }
  if constexpr (FUSE) {
    // rows of one channel group sit 16 lanes apart; then the eight waves meet in LDS (fixed order: bit-reproducible)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sg[e] += __shfl_xor(sg[e], 16, 64); sx[e] += __shfl_xor(sx[e], 16, 64);
      sg[e] += __shfl_xor(sg[e], 32, 64); sx[e] += __shfl_xor(sx[e], 32, 64);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every wave is done with the ring: reuse its first bytes
    float* red = reinterpret_cast<float*>(smem + WD_BYTES);  // [NW][2][64]
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(wave * 2 + 0) * 64 + lane * 4 + e] = sg[e]; red[(wave * 2 + 1) * 64 + lane * 4 + e] = sx[e]; }
    }
    __syncthreads();
    if (tid < 128) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[w * 128 + tid];
      const int which = tid >> 6, c = tid & 63;
      a.fuse.partial[(size_t)(a.fuse.row_base + slab) * 2 * a.Cin + (size_t)which * a.Cin + ci0 + c] = tot;
    }
  }
}

#undef PWB_ISSUE_TILE
#undef PWB_ISSUE_ONE

template <int CO4, int S, bool ADD, bool FUSE>
int launch_pwb(const PwbArgs& a, hipStream_t st) {
  constexpr int KB = 2 * CO4, NX = 1 + (ADD ? 1 : 0) + (FUSE ? 1 : 0);
  constexpr int LDS = KB * 4096 + S * (KB * 2048 + NX * T64) + BP * SPITCH + 1024;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = pw_bwd_kernel<CO4, S, ADD, FUSE>;
  static DynLds dyn_lds;                                     // per device, race-free (common.h)
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(kern), LDS); e != hipSuccess) return (int)e;
  const unsigned grid = (unsigned)(8 * cdiv(a.nslabs, 8) * a.slices);
  MDCV_LAUNCH(kern, dim3(grid), dim3(NT), LDS, st, a);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

template <int CO4, int S>
int launch_pwb2(const PwbArgs& a, hipStream_t st) {
  const bool add = a.addsrc != nullptr, fuse = a.fuse.y != nullptr;
  if (add) return fuse ? launch_pwb<CO4, S, true, true>(a, st) : launch_pwb<CO4, S, true, false>(a, st);
  return fuse ? launch_pwb<CO4, S, false, true>(a, st) : launch_pwb<CO4, S, false, false>(a, st);
}

// compute units of the current device (cached per device: hipGetDeviceProperties costs milliseconds, this runs in front of every launch)
int device_cus() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cus[dev]) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n : 256;
  }
  return cus[dev];
}

int pwb_lds_bytes(int Cout) {
  const int KB = Cout / 32, S = Cout >= 512 ? 2 : 3;
  return KB * 4096 + S * (KB * 2048 + 3 * T64) + BP * SPITCH + 1024;
}

}  // namespace

extern "C" {

/* Number of fp32 slabs [Cout][Cin] the fused 1x1 backward writes for this layer (size `ws` and the fused-sums partial rows with it),
 * or 0 when the layer does not take this form (then: mdcv_conv2d mode 1 + mdcv_conv2d_wgrad). */
int mdcv_pw_bwd_slabs(int dtype, long long M, int Cin, int Cout, int ldy, int ldx, int lddx, int ldadd, int ldfy) {
  if (dtype != MDCV_BF16 || M < 1) return 0;
  if (Cout != 64 && Cout != 128 && Cout != 256 && Cout != 512) return 0;
  if (Cin < 64 || (Cin & 63)) return 0;
  if ((ldy & 7) || (ldx & 7) || (lddx & 3) || (ldadd & 7) || (ldfy & 7)) return 0;
  const long long ldmax = (long long)(ldy > ldx ? ldy : ldx) > (ldadd > ldfy ? ldadd : ldfy) ? (ldy > ldx ? ldy : ldx) : (ldadd > ldfy ? ldadd : ldfy);
  if ((M + 64) * ldmax * 2 >= (1LL << 31) || (long long)Cin * Cout * 2 >= (1LL << 31)) return 0;
  const int ncu = device_cus();
  // Workgroup (slab, slice) runs on XCD slab % 8 (the slices of a slab share the XCD's L2): what one XCD can hold at once -- its CUs x the
  // workgroups of this layer that fit one CU's LDS -- bounds the slabs per XCD; a second round on some XCDs doubled the launch (26^2 768->256:
  // 21 slabs x 12 slices = 36 workgroups on five 32-CU XCDs: 72 us, against 16 slabs: one round)
  const int per_cu = (160 * 1024) / pwb_lds_bytes(Cout);
  const int slices = Cin / NS;
  const int xcd_slots = (ncu / 8) * (per_cu < 1 ? 1 : per_cu);
  int P = 8 * (xcd_slots / slices > 0 ? xcd_slots / slices : 1);
  long long spx = ((M + P - 1) / P + BP - 1) / BP * BP;
  if (spx < 4 * BP) spx = 4 * BP;                               // never less than four tiles per slab
  return (int)((M + spx - 1) / spx);
}

/* dx = dy . Wd (+ addsrc) [M x Cin], fused BatchNorm-backward sums of the layer in front (fy != NULL: partial rows [slabs][2][Cin]), and the
 * slabs of the weight gradient: ws[slabs][Cout][Cin] fp32, to be summed by mdcv_wgrad_reduce.  slabs = mdcv_pw_bwd_slabs(...). */
int mdcv_pw_bwd(int dtype, const void* dy, int ldy, const void* x, int ldx, const void* wd_packed, void* dx, int lddx, const void* addsrc,
                int ldadd, float* ws, int slabs, const void* fy, int ldfy, const float* fscale, const float* fshift, const float* fmean, int fact,
                float fslope, float* fpartial, long long M, int Cin, int Cout, void* stream) {
  if (!dy || !x || !wd_packed || !dx || !ws) return MDCV_EARG;
  if (slabs < 1 || slabs != mdcv_pw_bwd_slabs(dtype, M, Cin, Cout, ldy, ldx, lddx, addsrc ? ldadd : 8, fy ? ldfy : 8)) return MDCV_EARG;
  PwbArgs a{};
  a.dy = dy; a.x = x; a.wd = wd_packed; a.dx = dx; a.addsrc = addsrc; a.ws = ws;
  a.ldy = ldy; a.ldx = ldx; a.lddx = lddx; a.ldadd = ldadd;
  a.M = (int)M; a.Cin = Cin; a.Cout = Cout;
  a.nslabs = slabs; a.slices = Cin / NS;
  a.slab_px = (int)(((M + slabs - 1) / slabs + BP - 1) / BP * BP);      // covers M with `slabs` slabs (a trailing empty slab writes zeros)
  a.fuse = BnFuseArgs{};
  if (fy) {
    if (!fscale || !fshift || !fmean || !fpartial) return MDCV_EARG;
    a.fuse.y = fy; a.fuse.ldy = ldfy; a.fuse.scale = fscale; a.fuse.shift = fshift; a.fuse.mean = fmean; a.fuse.partial = fpartial;
    a.fuse.act = fact; a.fuse.slope = fact == 2 ? 0.f : fslope; a.fuse.row_base = 0;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (Cout) {
    case 64: return launch_pwb2<1, 3>(a, st);
    case 128: return launch_pwb2<2, 3>(a, st);
    case 256: return launch_pwb2<4, 3>(a, st);
    case 512: return launch_pwb2<8, 2>(a, st);
  }
  return MDCV_EARG;
}

}  // extern "C"

// 3x3 / stride 1 / pad = dilation convolution (forward and data gradient) as NINE SHIFTED GEMMs over one LDS-resident activation chunk.
//
// Why: the im2col kernel (conv_igemm.hip) fills LDS with a fresh A tile for every tap, i.e. every activation pixel travels
// L2 -> LDS nine times.  Measured on MI355X that fill path saturates at ~30 GB/s per CU (~12.6 B/clk/CU), which is exactly what
// holds the 3x3 layers at ~650 TFLOP/s (24 KiB of fill per 256x128x32 MACs).  Here the pixels of a tile are filled ONCE per
// 32-channel chunk and all nine taps read them at shifted LDS rows, so the fill per K step drops from 24 KiB to 2.6 + 8 KiB.
//
// Stream layout.  Output positions run over a 1-D stream with one shared zero column and one shared zero row:
//   p = img * (H+1)(W+1) + y * (W+1) + x          (x == W or y == H are junk positions, computed and discarded)
// and the input of tap (kh, kw) at p is the same stream read at p + kh*(W+1) + kw, shifted by one row and one column:
//   t' = p + kh*(W+1) + kw - (W+2)  ->  pixel (y+kh-1, x+kw-1); t' landing on a junk position IS the zero padding.
// (The row/column after each image row / image serves as the left/top padding of the next one.)  So the A operand of tap
// (kh,kw) for stream positions [p0, p0+256) is rows [d, d+256) of ONE LDS chunk holding stream rows [p0, p0+256+2(W+1)+2),
// d = kh*(W+1)+kw (forward) or (2-kh)*(W+1)+(2-kw) (data gradient).  Junk positions cost (H+1)(W+1)/(HW) - 1 extra MACs
// (4 % at 52x52, 8 % at 26x26, 16 % at 13x13).
//
// Dilation 2 (pad 2) uses the same scheme with TWO shared zero columns per image row and two shared zero rows per image:
//   p = img * (H+dil)(W+dil) + y * (W+dil) + x, tap displacement dil * (kh*(W+dil) + kw), halo 2*dil*(W+dil+1) rows.
// Output channels per tile: 128, or one 64- / 32-wide tile column for 64- / 32-channel layers (template parameter BN_; the narrow
// tiles' smaller weight ring also leaves room for rows up to 104 pixels with two workgroups on a CU).
//
// Pipeline.  256(M) x 128(N) tile, 8 waves of 64x64, K step = 32 channels of one tap.  The activation chunk (NPA KiB-chunks
// per wave) is double buffered per 32-channel chunk; the weight tiles (8 KiB per step) go through a 3-slot ring; since
// 9 % 3 == 0 every slot index and every vmcnt immediate is a compile-time constant of the (unrolled) tap index.
// All DMA addresses are loop invariant per lane (voffset) + a scalar (soffset): the K loop has no address VALU at all.
#include "common.h"
#include "conv_shift.h"
#include "bn_fuse.h"

// The file is compiled twice (Makefile: conv_shift_fwd.o with -DMDCV_SHIFT_PART=0, conv_shift_dgrad.o with -DMDCV_SHIFT_PART=1) so that the
// forward and the data-gradient instantiations build in parallel; part 0 also holds the host entry points and the tuning globals.
#ifndef MDCV_SHIFT_PART
#define MDCV_SHIFT_PART 0
#endif
#if MDCV_SHIFT_PART == 0
#else
#endif
int mdcv_shift_launch_dgrad(const ShiftArgs& a, hipStream_t st, unsigned in_bytes, unsigned w_bytes);   // defined by part 1

namespace {

[[maybe_unused]] constexpr int BN = 128;
constexpr int WM = 4;     // default tile width (BN) ; waves: 4 (M) x WN (N), WN = 2 (8 waves, wave tile (BM/4) x 64)
// (per instantiation: BTILE = BN * 64 bytes per weight tile of the ring, SROW = BN * 2 + 16 bytes epilogue staging pitch)
constexpr unsigned OOB = 0x80000000u;

// 16-byte slot of logical k-vector q in a 64-byte tile row: q ^ swz(row).  ds_read_b128 is serviced in 16-lane groups that hold
// rows i..i+3 and i+12..i+15 of one k-vector and rows i+4..i+11 of its neighbour (q^1); with swz = 2*((row>>2)&1) the four rows
// of a residue class (row&3) land in four different slots for EVERY row offset, so the tap-shifted reads are conflict-free too
// (the aligned-only swizzle (-(row>>2))&3 left 27-30 % of the LDS cycles as bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 2; }
typedef __attribute__((address_space(3))) void lds_void_t;

#if defined(MDCV_SHIFT_TS) || defined(MDCV_SHIFT_WG)
#include <cstdio>
__device__ long long g_shift_wg[3 * 4096];   // per workgroup: wall-clock start, end (100 MHz), hardware id
#endif
#ifdef MDCV_SHIFT_TS   /* per-step cycle stamps of one wave never defined in the shipped build */
__device__ long long g_shift_ts[4 * 512];
#define STS(k) do { if (ts_on && ts_i < 512) g_shift_ts[(k) * 512 + ts_i] = (long long)clock64(); } while (0)
#else
#define STS(k)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// The wait in front of a ring barrier also drains THIS wave's LDS reads (lgkmcnt(0)).  The compiler schedules the last MFMAs of a step --
// and the s_waitcnt lgkmcnt that guards their operands -- BELOW the next step's barrier (register-only instructions are not ordered by
// s_barrier or by an asm memory clobber), so a wave could sit behind the barrier with its last fragment reads still queued while another
// wave's LDS-DMA already refilled that ring slot: ds_read / ds_write of different waves are served in issue order, an LDS-DMA write is not
// ordered against them.  Seen as one wrong 16-column fragment of one wave about once per 2000 YOLOv3 steps, only with the weight-gradient
// stream running beside the kernel (scripts/repro_probe.py).
template <int N> __device__ __forceinline__ void wait_vm_reads_done() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

}  // namespace

template <int MODE, int BM, int NPA, int BRING, bool FUSE, int WN, bool EPI = false, int BN_ = 128, int LOOP = 0>
__global__ __launch_bounds__(WM * WN * 64) void mdcv_conv3x3_shift_kernel(ShiftArgs a, unsigned in_bytes, unsigned w_bytes) {
  // output channels per tile: 128, or 64 for 64-channel layers (wave tile (BM/4) x 32; waves 4..7 send their weight DMA to the sink)
  constexpr int BN = BN_, BTILE = BN * 64, SROW = BN * 2 + 16;
  constexpr int NW = WM * WN, TN = BN / WN, FN = TN / 16;
  constexpr int TM = BM / WM, FM = TM / 16;
  static_assert(!FUSE || (WN == 2 && BM % 128 == 0), "the fused sums: 8 waves, one partial row per 128 positions");
  // LDS: [A0: nca KiB][A1: nca KiB][weight ring: BRING x 8 KiB][1 KiB sink for the surplus chunk DMAs]; the epilogue reuses it
  // as [bf16 staging BM x SROW][rowpix BM ints][statistics WM*2*BN floats].
  constexpr int STAGE = BM * SROW;
  constexpr int PIX_OFF = STAGE;                           // int rowpix[BM]
  constexpr int STAT_OFF = STAGE + BM * 4;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int ABYTES = a.nca * 1024;
  const int BBASE = 2 * ABYTES;
  const int SINK = BBASE + BRING * BTILE;

  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.tiles_total) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tile_m = logical / a.tiles_n, tile_n = logical % a.tiles_n;
  const int p0 = a.p_base + tile_m * BM;
  const int lrow = lane >> 2, kv = (lane & 3) ^ swz(lrow);
  // (fills past the last K step use a descriptor with num_records = 0: everything is out of range, the DMA writes zeros
  //  into a slot nobody reads, and the DMA count per step stays constant)

  // ---- loop-invariant DMA offsets
  unsigned avo[NPA];
#pragma unroll
  for (int k = 0; k < NPA; ++k) {
    const int j = (wave + k * NW) * 16 + lrow;             // LDS row of the activation chunk
    int img, yy, xx;
    bool ok;
    if (a.t2d) {                                           // 2-D pixel tile: chunk row j = (row cy, column cx) of the tile with its halo ring
      const int tpi = a.tiles_x * a.tiles_y;
      img = tile_m / tpi;
      const int rt = tile_m - img * tpi, ty = rt / a.tiles_x, tx = rt - ty * a.tiles_x;
      const int cy = j / a.Wq, cx = j - cy * a.Wq;
      if (MODE == 3) { yy = ty * a.TH + cy; xx = tx * (a.Wq - 1) + cx; }        // stride-2 data gradient: the halo is below / right of the tile only
      else { yy = ty * a.TH - 1 + cy; xx = tx * (a.Wq - 2) - 1 + cx; }
      ok = yy >= 0 && xx >= 0;
    } else {
      const int t = p0 + j - a.dil * (a.Wq + 1);
      ok = t >= 0 && t < a.Mq;
      const int tt = ok ? t : 0;
      img = tt / a.Sq;
      const int rem = tt - img * a.Sq;
      yy = rem / a.Wq; xx = rem - yy * a.Wq;
    }
    ok = ok && yy < a.H && xx < a.W && kv * 8 < a.Cin;     // (Cin = 8: the other three k-vectors of a row are zeros, whatever the weight tile holds there)
    avo[k] = ok ? (unsigned)((((img * a.H + yy) * a.W + xx) * a.in_ldc + kv * 8) * 2) : OOB;
  }
  unsigned bvo;                                            // the weight tile is 8 KiB-chunks: waves 8..15 (16-wave variant) fill the sink
  {
    const int n = tile_n * BN + wave * 16 + lrow;
    bvo = (wave < BN / 16 && n < a.Nout) ? (unsigned)((n * a.wrow + kv * 8) * 2) : OOB;
  }
  // ---- per-tap fragment offsets (activation rows shifted by the tap displacement)
  const int r = lane & 15, q = lane >> 4;
  int offA[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    // MODE 3 (stride-2 data gradient, see the epilogue): tap (kh, kw) feeds output parity class (kh != 1, kw != 1) from the dY row / column
    // one further on when kh == 0 / kw == 0
    const int d = MODE == 3 ? (kh == 0 ? a.Wq : 0) + (kw == 0 ? 1 : 0)
                            : a.dil * (MODE == 0 ? kh * a.Wq + kw : (2 - kh) * a.Wq + (2 - kw));
    const int row = wm * TM + d + r;
    offA[tap] = row * 64 + ((q ^ swz(row)) << 4);
  }
  const int offB = (wn * TN + r) * 64 + ((q ^ swz(r)) << 4);

  // descriptors live in the kernel body (device builtins inside a lambda make the host pass drop the kernel stub)
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rin0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 0, 0x00020000);
#define ISSUE_A(RS, CHUNK, BUF)                                                                                       \
  do {                                                                                                              \
    const int base__ = (BUF) * ABYTES;                                                                              \
    const int so__ = (CHUNK) * 64;                                                                                  \
    _Pragma("unroll") for (int k = 0; k < NPA; ++k) {                                                               \
      const int ch__ = wave + k * NW;             /* every wave issues NPA DMAs; chunks past the last go to the sink */ \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lds_void_t*)(smem + (ch__ < a.nca ? base__ + ch__ * 1024 : SINK)), 16, \
                                               (int)avo[k], so__, 0, 0);                                            \
    }                                                                                                               \
  } while (0)
#define ISSUE_B(RS, TAP, CHUNK, SLOT)                                                                               \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lds_void_t*)(smem + (wave < BN / 16 ? BBASE + (SLOT) * BTILE + wave * 1024 : SINK)), 16, (int)bvo, \
                                           ((TAP) * a.Cin + (CHUNK) * 32) * 2, 0, 0)

  constexpr int NCLS = MODE == 3 ? 4 : 1;                   // stride-2 data gradient: one accumulator set per output parity class
#define TAP_CLS(T) (MODE == 3 ? ((((T) / 3) != 1 ? 2 : 0) + (((T) % 3) != 1 ? 1 : 0)) * FM : 0)
  f32x4_t acc[NCLS * FM][FN];
#pragma unroll
  for (int i = 0; i < NCLS * FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Step s = chunk * 9 + tap.  At step s the weight tile s+LA is issued (LA = BRING-1 tiles of lookahead) and, at tap 0,
  // the next activation chunk behind it.  DMAs complete in order, so "tile s has landed" = at most LA-1 newer weight tiles
  // outstanding, plus the chunk (NPA DMAs) while it is younger than tile s, i.e. at taps 1 .. LA.
  constexpr int LA = BRING - 1;
  const int nch = a.nchunks;
#ifdef MDCV_SHIFT_TS
  const bool ts_on = logical == 300 && tid == 64 * 3;
  int ts_i = 0;
#endif
#if defined(MDCV_SHIFT_TS) || defined(MDCV_SHIFT_WG)
  if (tid == 0 && logical < 4096) {
    g_shift_wg[logical] = (long long)wall_clock64();
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_shift_wg[2 * 4096 + logical] = (long long)hwid | ((long long)xcc << 32);
  }
#endif
  ISSUE_A(rin, 0, 0);
#pragma unroll
  for (int t = 0; t < LA; ++t) ISSUE_B(rw, t % 9, t / 9, t);
  // The chunk loop is unrolled over PERIOD chunks so that every ring slot, A buffer and wait count below is a compile-time
  // constant of (cc, tap): 9 * PERIOD is a multiple of BRING, and PERIOD is even or the A buffer index is taken from c.
  constexpr int PERIOD = BRING == 4 ? 4 : 2;              // 3- and 6-slot rings: 18 steps ; 4 slots: 36
  static_assert((9 * PERIOD) % BRING == 0, "ring period");
  if constexpr (LOOP == 1) {
    // PING-PONG form (round 4).  The eight waves are two groups of four (one wave of each group per SIMD); a K step of a wave is a LOAD
    // phase (fragment reads of step s, the DMA of weight tile s+LA, the counted wait for tile s+1) and an MFMA phase (16 MFMAs), each
    // closed by a barrier, and group 1 runs ONE BARRIER behind group 0: in every barrier interval one wave of a SIMD multiplies while its
    // partner reads and issues DMAs, instead of all eight waves reading, then all eight multiplying.
    //   interval 2s   : group 0 LOAD(s)      group 1 MFMA(s-1)
    //   interval 2s+1 : group 0 MFMA(s)      group 1 LOAD(s)
    // RAW: tile s+1 is waited for (own DMAs, counted vmcnt) at the end of LOAD(s) by both groups, i.e. no later than interval 2s+1, and
    // first read by group 0 in interval 2s+2 -- a barrier every wave has passed lies between.  WAR: tile s+LA goes into the slot of tile
    // s-1 (LA = BRING-1), whose last reads (group 1, LOAD(s-1), interval 2s-1) are DONE (lgkmcnt(0)) before the barrier that opens
    // interval 2s, where group 0 issues first.  The same count holds for the activation chunk (issued in LOAD(9c), its buffer last read in
    // LOAD(9c-1)).
    const int grp = wave >> 2;
    wait_vm<LA - 1>();                                     // own share of chunk 0 and weight tile 0 has landed
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int c0 = 0; c0 < nch; c0 += PERIOD) {
#pragma unroll
      for (int cc = 0; cc < PERIOD; ++cc) {
        const int c = c0 + cc;
        if (c < nch) {
          const bool lastc = c == nch - 1;
          const int abase = (cc & 1) * ABYTES;
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int rslot = (cc * 9 + tap) % BRING, wslot = (cc * 9 + tap + LA) % BRING;
            // ---- LOAD phase
            bf16x8_t fa[FM], fb[FN];
            const unsigned char* pa = smem + abase + offA[tap];
            const unsigned char* pb = smem + BBASE + rslot * BTILE + offB;
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(pb + j * 1024);
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(pa + i * 1024);
            {
              const int t2 = (tap + LA) % 9, c2 = c + (tap + LA) / 9;
              if (tap + LA >= 9 && lastc) ISSUE_B(rw0, t2, c2, wslot);
              else ISSUE_B(rw, t2, c2, wslot);
            }
            if (tap == 0) {
              if (lastc) ISSUE_A(rin0, c + 1, (cc + 1) & 1);
              else ISSUE_A(rin, c + 1, (cc + 1) & 1);
            }
            // tile s+1 landed: LA-1 newer weight tiles may be outstanding, and the chunk while it is younger than tile s+1 (taps 0 .. LA-1)
            if (tap <= LA - 1) wait_vm_reads_done<LA - 1 + NPA>(); else wait_vm_reads_done<LA - 1>();
#pragma unroll
            for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(fb[j]));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA phase
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
              for (int j = 0; j < FN; ++j) acc[TAP_CLS(tap) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[TAP_CLS(tap) + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
  for (int c0 = 0; c0 < nch; c0 += PERIOD) {
#pragma unroll
    for (int cc = 0; cc < PERIOD; ++cc) {
      const int c = c0 + cc;
      if (c < nch) {
        const bool lastc = c == nch - 1;
        const int abase = (cc & 1) * ABYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int rslot = (cc * 9 + tap) % BRING, wslot = (cc * 9 + tap + LA) % BRING;
          STS(0);
          // Every MFMA of the previous step is ISSUED before this barrier -- hence each fragment read it consumes has RETURNED (see
          // wait_vm_reads_done above for why that matters; pinning the accumulators costs nothing measurable, draining lgkmcnt here +0.4 %).
#pragma unroll
          for (int i = 0; i < NCLS * FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(acc[i][j]));
          if (tap >= 1 && tap <= LA) wait_vm_reads_done<LA - 1 + NPA>(); else wait_vm_reads_done<LA - 1>();
          STS(1);
          __builtin_amdgcn_s_barrier();
          STS(2);
          bf16x8_t fa[FM], fb[FN];
          const unsigned char* pa = smem + abase + offA[tap];
          const unsigned char* pb = smem + BBASE + rslot * BTILE + offB;
#pragma unroll
          for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(pa + i * 1024);
#pragma unroll
          for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(pb + j * 1024);
          {
            const int t2 = (tap + LA) % 9, c2 = c + (tap + LA) / 9;
            if (tap + LA >= 9 && lastc) ISSUE_B(rw0, t2, c2, wslot);   // past the last K step: zero fill, same DMA count
            else ISSUE_B(rw, t2, c2, wslot);
          }
          if (tap == 0) {
            if (lastc) ISSUE_A(rin0, c + 1, (cc + 1) & 1);
            else ISSUE_A(rin, c + 1, (cc + 1) & 1);
          }
#ifdef MDCV_SHIFT_PRIO
          __builtin_amdgcn_s_setprio(MDCV_SHIFT_PRIO);
#endif
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[TAP_CLS(tap) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[TAP_CLS(tap) + i][j], 0, 0, 0);
#ifdef MDCV_SHIFT_PRIO
          __builtin_amdgcn_s_setprio(0);
#endif
          STS(3);
#ifdef MDCV_SHIFT_TS
          ++ts_i;
#endif
        }
      }
    }
  }
  }
  wait_vm<0>();
  __syncthreads();                                         // the epilogue reuses the pipeline LDS

  if constexpr (MODE == 3) {
    // Stride-2 data gradient: dx[2a + py][2b + px] of the tile's dY positions (a, b), class (py, px) in accumulator set 2 py + px.  For each
    // py the two column classes are interleaved in LDS as OUTPUT rows (row la, 64 pixels 2 lb + px of BN channels), so the stores are whole
    // contiguous output rows (the per-class launches of the im2col path write every 128-byte line as two 64-byte halves at different times).
    static_assert(MODE != 3 || (BM == 256 && !EPI), "stride-2 data gradient: 8 x 32 dY positions per tile");
    const int tpi = a.tiles_x * a.tiles_y;
    const int img = tile_m / tpi, rt = tile_m - img * tpi, ty = rt / a.tiles_x, tx = rt - ty * a.tiles_x;
    const int a0 = ty * a.TH, b0 = tx * (a.Wq - 1);
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    bf16_t* __restrict__ out3 = reinterpret_cast<bf16_t*>(a.out);
    const bf16_t* __restrict__ add3 = reinterpret_cast<const bf16_t*>(a.addsrc);
    constexpr int VPRO3 = BN / 8;                          // 16-byte vectors per output pixel = vectors per thread and row-parity pass
    static_assert((512 * VPRO3) % (NW * 64) == 0 && (NW * 64) % VPRO3 == 0, "every thread keeps one channel vector");
    using Acc3 = BnFuseAcc<bf16_t, BN, NW * 64>;
    Acc3 fz3;
    const bf16_t* __restrict__ fy3 = reinterpret_cast<const bf16_t*>(a.fuse.y);
    if constexpr (FUSE) fz3.init(a.fuse, tile_n * BN + (tid % VPRO3) * 8, a.Nout);
#pragma unroll
    for (int py = 0; py < 2; ++py) {
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int p = wm * TM + i * 16 + (lane >> 4) * 4 + rr, col = wn * TN + j * 16 + (lane & 15);
              const int srow = (p >> 5) * 64 + 2 * (p & 31) + px;
              reinterpret_cast<bf16_t*>(smem + srow * SROW)[col] = f2bf(acc[(2 * py + px) * FM + i][j][rr]);
            }
      __syncthreads();
      // every thread: VPRO3 vectors of this pass; their global operands (addsrc, y of the fused sums) are all issued first
      long long pixv[VPRO3]; uint4 aq[VPRO3], yq[VPRO3];
      const int cv = tid % VPRO3, n = tile_n * BN + cv * 8;
#pragma unroll
      for (int k = 0; k < VPRO3; ++k) {
        const int srow = (tid + k * NW * 64) / VPRO3;
        const int la = srow >> 6, ox = srow & 63;
        const int Y = 2 * (a0 + la) + py, X = 2 * b0 + ox;
        const bool ok = ox < 2 * (a.Wq - 1) && Y < Ho && X < Wo && n < a.Nout;
        pixv[k] = ok ? ((long long)img * Ho + Y) * Wo + X : -1;
        if (ok) {
          if (add3) aq[k] = *reinterpret_cast<const uint4*>(add3 + pixv[k] * a.add_ldc + n);
          if constexpr (FUSE) yq[k] = *reinterpret_cast<const uint4*>(fy3 + pixv[k] * a.fuse.ldy + n);
        }
      }
#pragma unroll
      for (int k = 0; k < VPRO3; ++k) {
        if (pixv[k] >= 0) {
          const int srow = (tid + k * NW * 64) / VPRO3;
          uint4 d = *reinterpret_cast<const uint4*>(smem + srow * SROW + cv * 16);
          float x[8];
          if (add3 || FUSE) ET<bf16_t>::unpack(d, x);
          if (add3) {
            float y[8];
            ET<bf16_t>::unpack(aq[k], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += y[e];
            d = ET<bf16_t>::pack(x);
            if constexpr (FUSE) ET<bf16_t>::unpack(d, x);   // the sums see dz as stored
          }
          if constexpr (FUSE) {
            fz3.add(a.fuse, x, yq[k]);
          }
          *reinterpret_cast<uint4*>(out3 + pixv[k] * a.out_ldc + n) = d;
        }
      }
      __syncthreads();
    }
    if constexpr (FUSE) {                                    // one partial row per tile (mdcv_shift_s2_rows): every wave folds its sums with row swaps
      static_assert(Acc3::kWaveFold, "bf16, at most 16 channel vectors per pixel");   // into 2*BN floats of its own, the waves meet once
      float* ws3 = reinterpret_cast<float*>(smem + 512 * SROW);
      fz3.fold_wave(ws3 + wave * 2 * BN, lane);
      lds_only_barrier();
      for (int t = tid; t < 2 * BN; t += NW * 64) Acc3::write_row(a.fuse, ws3, 2 * BN, t, tile_n * BN, a.Nout, tile_m);
    }
    return;
  }
  // ---------------- epilogue ----------------
  int* rowpix = reinterpret_cast<int*>(smem + PIX_OFF);
  if (tid < BM) {
    int img, y, x;
    bool ok;
    if (a.t2d) {                                           // output position tid of the tile: row tid / Wq, column tid % Wq; the last two columns are junk
      const int tpi = a.tiles_x * a.tiles_y;
      img = tile_m / tpi;
      const int rt = tile_m - img * tpi, ty = rt / a.tiles_x, tx = rt - ty * a.tiles_x;
      const int ly = tid / a.Wq, lx = tid - ly * a.Wq;
      y = ty * a.TH + ly; x = tx * (a.Wq - 2) + lx;
      ok = lx < a.Wq - 2 && ly < a.TH;
    } else {
      const int p = p0 + tid;
      ok = p < a.Mq;
      const int pp = ok ? p : 0;
      img = pp / a.Sq;
      const int rem = pp - img * a.Sq;
      y = rem / a.Wq; x = rem - y * a.Wq;
    }
    ok = ok && y < a.H && x < a.W;
    rowpix[tid] = ok ? (img * a.H + y) * a.W + x : -1;
  }
  const int n0 = tile_n * BN + wn * TN;
  if constexpr (EPI) {                                     // inference instantiation (MODE 0): act(acc * scale + shift), once per tile.  A
    // template parameter, not a runtime branch: even this block-uniform test outside the store loops cost the training step 2 %
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + j * 16 + (lane & 15);
      const float sc = (a.epi.oscale && n < a.Nout) ? a.epi.oscale[n] : 1.f;
      const float bv = (a.bias && n < a.Nout) ? a.bias[n] : 0.f;
      const float sl = a.epi.act == 1 ? a.epi.slope : (a.epi.act == 2 ? 0.f : 1.f);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float v = acc[i][j][rr] * sc + bv;
          acc[i][j][rr] = v > 0.f ? v : v * sl;
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
        for (int rr = 0; rr < 4; ++rr) acc[i][j][rr] += bv;
    }
  }
  __syncthreads();
  float* sstat = reinterpret_cast<float*>(smem + STAT_OFF);
  const bool want_stats = a.stats || a.xacc.acc;
  if (want_stats) {
    bool live[FM][4];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) live[i][rr] = rowpix[wm * TM + i * 16 + (lane >> 4) * 4 + rr] >= 0;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float s = 0.f, qq = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float v = live[i][rr] ? acc[i][j][rr] : 0.f;
          s += v; qq += v * v;
        }
      s += __shfl_xor(s, 16, 64); qq += __shfl_xor(qq, 16, 64);
      s += __shfl_xor(s, 32, 64); qq += __shfl_xor(qq, 32, 64);
      if (lane < 16) {
        sstat[(wm * 2 + 0) * BN + wn * TN + j * 16 + lane] = s;
        sstat[(wm * 2 + 1) * BN + wn * TN + j * 16 + lane] = qq;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; rr += 2) {
        const int row = wm * TM + i * 16 + (lane >> 4) * 4 + rr, col = wn * TN + j * 16 + (lane & 15);
        const unsigned pk = pack_bf16x2(acc[i][j][rr], acc[i][j][rr + 1]);     // one v_cvt_pk_bf16_f32 per two rows
        reinterpret_cast<bf16_t*>(smem + row * SROW)[col] = (bf16_t)(pk & 0xffffu);
        reinterpret_cast<bf16_t*>(smem + (row + 1) * SROW)[col] = (bf16_t)(pk >> 16);
      }
  __syncthreads();
  constexpr int GR = (BM == 128 || BM == 256) ? 128 : BM;   // stream positions per partial-statistics row (192- / 384-row tiles: one row per tile)
  constexpr int G = BM / GR, WPG = WM / G;                 // rows per tile; waves (in M) per row
  if (want_stats && tid < BN * G) {
    const int g = tid / BN, col = tid - g * BN;
    const int n = tile_n * BN + col;
    if (n < a.Nout) {
      float s = 0.f, qq = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < WPG; ++w2) { s += sstat[((g * WPG + w2) * 2 + 0) * BN + col]; qq += sstat[((g * WPG + w2) * 2 + 1) * BN + col]; }
      const size_t srow = (size_t)(p0 / GR) + g;
      if (srow * GR < (size_t)a.Mq) {                       // the last tile may reach past the stream: the caller holds ceil(Mq/GR) rows
        if (a.xacc.acc) {                                   // fire-and-forget exact accumulation (exact_acc.h): no rows, no finalize launch
          long long* xp = a.xacc.acc + (size_t)(srow & (size_t)(a.xacc.reps - 1)) * (XACC_DIGITS * 2) * a.Nout + n;
          xacc_add(xp, 2 * (size_t)a.Nout, s);
          xacc_add(xp + a.Nout, 2 * (size_t)a.Nout, qq);
        } else {
          a.stats[(srow * 2 + 0) * a.Nout + n] = s;         // (an unguarded second row of the last 256-row tile wrote 2*Nout floats past
          a.stats[(srow * 2 + 1) * a.Nout + n] = qq;        //  the buffer whenever ceil(Mq/128) was odd, e.g. batch 32 at 52/26/13)
        }
      }
    }
  }
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
  const bf16_t* __restrict__ addsrc = reinterpret_cast<const bf16_t*>(a.addsrc);
  constexpr int VPRO = BN / 8;
  if constexpr (!FUSE) {
    for (int v = tid; v < BM * VPRO; v += NW * 64) {
      const int row = v / VPRO, cv = v - row * VPRO;
      const int pix = rowpix[row], n = tile_n * BN + cv * 8;
      if (pix >= 0 && n < a.Nout) {
        uint4 d = *reinterpret_cast<const uint4*>(smem + row * SROW + cv * 16);
        if (addsrc) {
          float x[8], y[8];
          ET<bf16_t>::unpack(d, x);
          ET<bf16_t>::unpack(*reinterpret_cast<const uint4*>(addsrc + ((size_t)pix * a.add_ldc + n)), y);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += y[e];
          d = ET<bf16_t>::pack(x);
        }
        *reinterpret_cast<uint4*>(out + ((size_t)pix * a.out_ldc + n)) = d;
      }
    }
  } else {
    // data gradient with the BatchNorm-backward sums of the producer layer folded in (bn_fuse.h)
    using Acc = BnFuseAcc<bf16_t, BN, NW * 64>;
    Acc fz;
    const int cv = tid % VPRO, n = tile_n * BN + cv * 8;
    fz.init(a.fuse, n, a.Nout);
    const bf16_t* __restrict__ fy = reinterpret_cast<const bf16_t*>(a.fuse.y);
    float* fred = reinterpret_cast<float*>(smem + STAT_OFF);
    constexpr int PPG = 128 / Acc::RPP;                    // passes per 128-position group
    // The global loads (addsrc, y) of ALL groups are issued first: they are HBM misses, and with one batch of loads per 128-row
    // group their latency was exposed once per group (0.95 ms per YOLOv3 step over the 66 fused data gradients); the accumulators
    // are dead by now, so the registers are free.
    constexpr int NG = BM / 128;
    int pixv[NG][PPG]; uint4 aq[NG][PPG], yq[NG][PPG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
      for (int u = 0; u < PPG; ++u) {
        const int row = gi * 128 + u * Acc::RPP + tid / VPRO;
        pixv[gi][u] = n < a.Nout ? rowpix[row] : -1;
        if (pixv[gi][u] >= 0) {
          if (addsrc) aq[gi][u] = *reinterpret_cast<const uint4*>(addsrc + ((size_t)pixv[gi][u] * a.add_ldc + n));
          yq[gi][u] = *reinterpret_cast<const uint4*>(fy + ((size_t)pixv[gi][u] * a.fuse.ldy + n));
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
          float x[8];
          uint4 d = dq[u];
          ET<bf16_t>::unpack(d, x);
          if (addsrc) {
            float y[8];
            ET<bf16_t>::unpack(aq[gi][u], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += y[e];
            d = ET<bf16_t>::pack(x);
            ET<bf16_t>::unpack(d, x);                     // the sums see dz as stored
          }
          *reinterpret_cast<uint4*>(out + ((size_t)pixv[gi][u] * a.out_ldc + n)) = d;
          fz.add(a.fuse, x, yq[gi][u]);
        }
      }
      if (p0 + g0 < a.Mq) {                                // block-uniform: rows past the stream do not exist
        if constexpr (Acc::kWaveFold)
          fz.fold_wave(reinterpret_cast<float*>(smem + (g0 + wave * Acc::WROWS) * SROW), lane);   // the rows this wave read in its first pass: dead, private
        else
          fz.flush(a.fuse, fred, tid, tile_n * BN, a.Nout, (p0 + g0) >> 7);
      }
    }
    if constexpr (Acc::kWaveFold) {                          // the waves meet once, behind the tile's last store (bn_fuse.h)
      static_assert(Acc::WROWS * SROW >= 2 * BN * 4 && SROW % 4 == 0, "a wave's dead staging rows hold its 2*BN sums");
      lds_only_barrier();
      for (int t = tid; t < NG * 2 * BN; t += NW * 64) {
        const int gi = t / (2 * BN), g0 = gi * 128;
        if (p0 + g0 < a.Mq)
          Acc::write_row(a.fuse, reinterpret_cast<const float*>(smem + g0 * SROW), Acc::WROWS * SROW / 4, t - gi * 2 * BN, tile_n * BN, a.Nout, (p0 + g0) >> 7);
      }
    }
  }
#if defined(MDCV_SHIFT_TS) || defined(MDCV_SHIFT_WG)
  if (tid == 0 && logical < 4096) g_shift_wg[4096 + logical] = (long long)wall_clock64();
#endif
}

namespace {

template <int MODE, int BM, int NPA, bool FUSE, int WN, bool EPI = false, int BRING = 3, int BN_ = 128, int LOOP = 0>
int launch_shift_f(ShiftArgs a, int p_base, int tiles_m, hipStream_t st, unsigned in_bytes, unsigned w_bytes) {
  constexpr int NW = WM * WN;
  constexpr int BN = BN_, BTILE = BN * 64, SROW = BN * 2 + 16;
  if constexpr (LOOP == 0 && WN == 2 && !EPI && MODE == 0 && !FUSE) {   // ping-pong K loop: forward launches only (measured, see TUNE().shift_loop)
    if (TUNE().shift_loop == 1 || BM > 256 || (TUNE().shift_loop == 2 && tiles_m * a.tiles_n <= 256))
      return launch_shift_f<MODE, BM, NPA, FUSE, WN, EPI, BRING, BN_, 1>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  }
  // A grid that puts one workgroup on a CU has only the ring's lookahead in flight on that CU's L2 -> LDS path (latency-bound fill):
  // such launches (batch 32: the 13x13 and 26x26 data gradients) take a 4-slot weight ring.  Same-box A/B of the YOLOv3 step:
  // +0.45 .. 0.6 % (6 slots +0.35 %; 4 slots on EVERY grid -2.8 %: the 36-step unrolled period and the third workgroup's worth of LDS).
  if constexpr (BRING == 3 && WN == 2 && !EPI) {
    if ((TUNE().shift_ring == 4 && tiles_m * a.tiles_n <= 256) || TUNE().shift_ring == 5)
      return launch_shift_f<MODE, BM, NPA, FUSE, WN, EPI, 4, BN_, LOOP>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  }
  a.p_base = p_base;
  a.tiles_total = tiles_m * a.tiles_n;
  a.xcd_chunk = (a.tiles_total + 7) / 8;
  a.nca = (BM + 2 * a.dil * (a.Wq + 1) + 15) / 16;         // KiB-chunks (16 stream rows each) of one activation chunk
  const int pipe = 2 * a.nca * 1024 + BRING * BTILE + 1024;
  const int epi = MODE == 3 ? 512 * SROW + NW * 2 * BN * 4      // stride-2 data gradient: two column classes of the tile interleaved as output rows (+ fold scratch)
                            : BM * SROW + BM * 4 + WM * 2 * BN * 4;      // staging + position table + statistics (the fused sums fold inside dead staging rows)
  const int lds = pipe > epi ? pipe : epi;
  static DynLds dyn_lds;
  auto kern = mdcv_conv3x3_shift_kernel<MODE, BM, NPA, BRING, FUSE, WN, EPI, BN_, LOOP>;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return (int)e;
  MDCV_LAUNCH(kern, dim3((unsigned)(a.xcd_chunk * 8)), dim3(NW * 64), lds, st, a, in_bytes, w_bytes);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

template <int MODE, int BM, int NPA, int WN, int BN_>
int launch_shift(const ShiftArgs& a, int p_base, int tiles_m, hipStream_t st, unsigned in_bytes, unsigned w_bytes) {
  if constexpr (MODE == 3) {
    if (a.fuse.y) return launch_shift_f<MODE, BM, NPA, true, WN, false, 3, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  }
  if constexpr (MODE == 1 && WN == 2 && BM % 128 == 0) {   // the fused BatchNorm-backward sums exist for 8-wave data gradients only
    if (a.fuse.y) return launch_shift_f<MODE, BM, NPA, true, WN, false, 3, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  }
  if constexpr (MODE == 0 && WN == 2) {                    // inference epilogue: forward, 8-wave tiles
    if (a.epi.oscale || a.epi.act) return launch_shift_f<MODE, BM, NPA, false, WN, true, 3, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  }
  if (a.epi.oscale || a.epi.act) return MDCV_EARG;          // no inference instantiation for this mode / wave layout
  return launch_shift_f<MODE, BM, NPA, false, WN, false, 3, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);
}

// (16-wave workgroups -- 4 x 4 waves of (BM/4) x 32 -- were tried for grids that put at most one workgroup on a CU and measured slower
// everywhere: 26x26 dgrad 54.4 vs 50.4 us, 13x13 forward 56.1 vs 52.2 us; removed in round 3.)
template <int MODE, int BM, int BN_>
int launch_shift_bm(const ShiftArgs& a, int p_base, int tiles_m, hipStream_t st, unsigned in_bytes, unsigned w_bytes) {
  const int nca = (BM + 2 * a.dil * (a.Wq + 1) + 15) / 16;
  const int npa = (nca + 7) / 8;
  if (npa <= 2) return launch_shift<MODE, BM, 2, 2, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  if (npa == 3) return launch_shift<MODE, BM, 3, 2, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  if (npa == 4) return launch_shift<MODE, BM, 4, 2, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);
  return launch_shift<MODE, BM, 5, 2, BN_>(a, p_base, tiles_m, st, in_bytes, w_bytes);   // dilation 2 at 80 pixels per row
}

// Tile plan.  Inside a busy CU the K loop is MFMA-bound whether one or two workgroups share it (a lone workgroup simply runs
// twice as fast), so an isolated launch takes ceil(tiles / 256) x (time of one tile) and loses the CUs left without a tile in
// the last round.  192-row tiles (wave tile 48 x 64) cut that loss and are 7-12 % faster in a tight loop on the 52^2 / 26^2
// layers (scripts/conv_ab.py), but inside the training step the idle CUs are not wasted — the weight-gradient stream fills
// them — and the 192-row tile's lower MFMA density per barrier made the step 0.7 % slower while the conv kernels shared the CUs evenly
// with the rest; since the BatchNorm sums moved into the data gradients the main stream bounds the step and the 192-row plan is
// +0.6 % (2043 -> 2055 img/s, same-box A/B), so it is the default where it saves a round (batch 32: the 26x26 layers); 128-row
// tiles only for grids of at most 128 tiles; plan 6 restores 256-row-only.
int shift_plan_bm(int Mq, int tiles_n, bool fused, int halo = 0, int bn = 128, bool fwd = false) {
  if (halo > 0 && !fused && 2 * ((256 + halo + 15) / 16) * 1024 + 3 * bn * 64 + 1024 > 80 * 1024) {
    // dilation 2 / very wide rows: the halo (2 * dil * (Wq + 1) rows) dominates the LDS footprint; tallest tile that leaves two workgroups on a CU
    for (int bm = 256; bm >= 128; bm -= 64) {
      const int nca = (bm + halo + 15) / 16;
      if (2 * nca * 1024 + 3 * bn * 64 + 1024 <= 80 * 1024) return bm;
    }
    return 256;
  }
  if (bn == 128 && !fused && fwd && halo <= 110) {       // 384-row ping-pong tiles (one workgroup per CU): forward only
    const int t384 = ((Mq + 383) / 384) * tiles_n;
    if (TUNE().shift_big == 384 || (TUNE().shift_loop == 2 && t384 > 192 && t384 <= 256)) return 384;
  }
  if (TUNE().shift_plan == 1) return 256;
  if (TUNE().shift_plan == 2) return 128;
  const int t256 = ((Mq + 255) / 256) * tiles_n;
  if (t256 <= 128) return 128;                       // (measured: 184- and 200-tile grids are still faster as 256-row tiles)
  if (fused || TUNE().shift_plan == 6) return 256;       // plan 6: the old default (no 192-row tiles)
  const int t192 = ((Mq + 191) / 192) * tiles_n;
  const int c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;
  return c192 < c256 ? 192 : 256;
}

template <int MODE, int BN_>
int launch_shift_mode(const ShiftArgs& a, hipStream_t st, unsigned in_bytes, unsigned w_bytes) {
  if constexpr (MODE == 3) {                                 // stride-2 data gradient: one 8 x 32 position tile per workgroup
    return launch_shift_bm<MODE, 256, BN_>(a, 0, a.Mq / 256, st, in_bytes, w_bytes);
  } else {
  if (a.t2d) return launch_shift_bm<MODE, 256, BN_>(a, 0, a.Mq / 256, st, in_bytes, w_bytes);   // one 8 x 32 position tile per workgroup
  const int bm = shift_plan_bm(a.Mq, a.tiles_n, a.fuse.y != nullptr, 2 * a.dil * (a.Wq + 1), BN_, MODE == 0);
  if constexpr (BN_ == 128 && MODE == 0) {
    if (bm == 384) return launch_shift_bm<MODE, 384, BN_>(a, 0, (a.Mq + 383) / 384, st, in_bytes, w_bytes);
  }
  if (bm == 192) return launch_shift_bm<MODE, 192, BN_>(a, 0, (a.Mq + 191) / 192, st, in_bytes, w_bytes);
  const int big_m = (a.Mq + 255) / 256;
  const int nbig_m = bm == 128 ? 0 : big_m;
  if (nbig_m > 0) {
    const int rc = launch_shift_bm<MODE, 256, BN_>(a, 0, nbig_m, st, in_bytes, w_bytes);
    if (rc) return rc;
  }
  const int p_base = nbig_m * 256;
  if (p_base < a.Mq) return launch_shift_bm<MODE, 128, BN_>(a, p_base, (a.Mq - p_base + 127) / 128, st, in_bytes, w_bytes);
  return MDCV_OK;
  }
}

}  // namespace

#if MDCV_SHIFT_PART == 1
int mdcv_shift_launch_dgrad(const ShiftArgs& a, hipStream_t st, unsigned in_bytes, unsigned w_bytes) {
  if (a.t2d == 2) {                                          // stride-2 data gradient (MODE 3): 32- and 64-channel outputs
    if (a.Nout == 32) return launch_shift_mode<3, 32>(a, st, in_bytes, w_bytes);
    if (a.Nout == 64) return launch_shift_mode<3, 64>(a, st, in_bytes, w_bytes);
    return MDCV_EARG;
  }
  if (a.Nout == 32) return launch_shift_mode<1, 32>(a, st, in_bytes, w_bytes);
  if (a.Nout == 64) return launch_shift_mode<1, 64>(a, st, in_bytes, w_bytes);
  // Few positions, many channels (13^2 x 32 images, 1024 -> 512: 25 x 4 tiles of 256 x 128): the plan below would fall back to 128-row tiles,
  // which stream the same 2.4 MB of weights per tile for half the MFMA work.  256 x 64 tiles give the same number of workgroups with half
  // the weight stream each (variant -64 / -63: on / off).
  if (TUNE().shift_n64_wide && !a.t2d && a.Nout % 64 == 0 && ((a.Mq + 255) / 256) * a.tiles_n <= 128) {
    ShiftArgs b = a;
    b.tiles_n = a.Nout / 64;
    return launch_shift_mode<1, 64>(b, st, in_bytes, w_bytes);
  }
  return launch_shift_mode<1, 128>(a, st, in_bytes, w_bytes);
}
#else
// ---- host side (internal linkage across the library's objects: declared in conv_shift.h)
// The 1-D position stream keeps a tile's 256 positions + 2 (W+1) + 2 halo rows in LDS: up to W = 62 the chunk is 384 rows, up to 80 (104 for
// the 64- / 32-wide tiles with their smaller weight ring) two workgroups still fit a CU.  Wider images (YOLOv3's 208^2 layers, the 104^2
// layers with 128 output channels, 152^2 / 304^2 at 608^2) are cut into 2-D pixel tiles instead: 8 rows x 30 columns of outputs per
// workgroup, stored with a one-pixel halo ring as a [10][32] mini-image whose taps are again pure row displacements kh * 32 + kw; the
// two positions per row whose window would wrap are junk (6 %), the chunk is 322 rows however wide the image is.  Same kernel, same
// K loop: only the DMA source addresses and the position -> pixel table of the epilogue differ (ShiftArgs.t2d).
static bool shift_fits_1d(int W, int Nout) {
  return W <= TUNE().shift_wmax || (TUNE().shift_wmax_narrow && Nout <= 64 && W <= TUNE().shift_wmax_narrow) || (TUNE().shift_wmax_n32 && Nout <= 32 && W <= TUNE().shift_wmax_n32);
}
static bool shift_is_2d(int W, int Nout, int dil) { return dil == 1 && TUNE().shift_2d && !shift_fits_1d(W, Nout); }
constexpr int T2D_WQ = 32, T2D_TH = 8;                      // tile = 8 x 32 positions = 256 (30 output columns + 2 junk)
static long long shift_2d_positions(int B, int H, int W) { return (long long)B * ((H + T2D_TH - 1) / T2D_TH) * ((W + T2D_WQ - 3) / (T2D_WQ - 2)) * 256; }

bool mdcv_shift_eligible(int dtype, int B, int H, int W, int Cin, int Nout, int KH, int KW, int stride, int pad, int dil, long long in_ldc) {
  if (dtype != MDCV_BF16 || KH != 3 || KW != 3 || stride != 1 || pad != dil || (dil != 1 && !(dil == 2 && TUNE().shift_dil2))) return false;
  // dilation 2 pays where the halo-heavy chunk is amortised over >= 2 channel chunks and two workgroups still fit a CU (narrow tiles):
  // RektNet data gradient 128->64 412 -> 332 us, 64->32 193 -> 183 us; forward 64->128 (128-wide tile, one workgroup per CU) 281 -> 360 us
  if (dil == 2 && TUNE().shift_dil2 == 1 && !(Nout <= 64 && Cin >= 64)) return false;
  if ((Cin & 31) || ((Nout & 127) && !(Nout == 64 && TUNE().shift_n64) && !(Nout == 32 && TUNE().shift_n64 == 2))) return false;   // 128-wide tiles, or one 64-wide tile column
  if (H < 8 || W < 8) return false;
  if (!shift_fits_1d(W, Nout) && !(TUNE().shift_2d && dil == 1)) return false;   // wider rows: 2-D pixel tiles (dilation 1 only), or not at all
  if ((long long)B * (H + dil) * (W + dil) + 1024 >= (1LL << 30) || shift_2d_positions(B, H, W) >= (1LL << 30)) return false;
  if ((long long)B * H * W * in_ldc * 2 >= (1LL << 31) || (long long)Nout * 9 * Cin * 2 >= (1LL << 31)) return false;
  return true;
}

// 3x3 / stride-2 / pad-1 data gradient with an even output (dx = 2H x 2W from dY = H x W) and 32 or 64 output channels: the two HBM-bound
// layers of YOLOv3 (208 -> 416, 104 -> 208).  mdcv_shift_conv(mode 3).
bool mdcv_shift_s2_dgrad_eligible(int dtype, int B, int H, int W, int Cin, int Nout, long long in_ldc) {
  if (!TUNE().shift_s2 || dtype != MDCV_BF16 || (Cin & 31) || Cin < 32 || !(Nout == 32 || Nout == 64) || H < 1 || W < 1) return false;
  if ((long long)B * ((H + 7) / 8) * ((W + 30) / 31) * 256 >= (1LL << 30)) return false;
  if ((long long)B * H * W * in_ldc * 2 >= (1LL << 31) || (long long)Nout * 9 * Cin * 2 >= (1LL << 31)) return false;
  return true;
}

int mdcv_shift_s2_rows(int B, int H, int W) { return B * ((H + T2D_TH - 1) / T2D_TH) * ((W + T2D_WQ - 2) / (T2D_WQ - 1)); }   // fused-sum rows of mode 3: one per tile

int mdcv_shift_stats_rows(int B, int H, int W, int dil, int Nout) {
  if (shift_is_2d(W, Nout, dil)) return (int)(shift_2d_positions(B, H, W) / 128);
  return (int)(((long long)B * (H + dil) * (W + dil) + 127) / 128);
}
// rows of the FORWARD statistics buffer: one per 128 stream positions, or one per tile when the plan picks 192-row tiles
int mdcv_shift_fwd_stats_rows(int B, int H, int W, int Nout, int dil) {
  if (shift_is_2d(W, Nout, dil)) return (int)(shift_2d_positions(B, H, W) / 128);
  const int Mq = B * (H + dil) * (W + dil);
  const int bn = Nout <= 64 ? Nout : BN;
  const int bm = shift_plan_bm(Mq, Nout <= 64 ? 1 : Nout / BN, false, 2 * dil * (W + dil + 1), bn, true);
  return (bm == 192 || bm == 384) ? (Mq + bm - 1) / bm : (Mq + 127) / 128;
}

int mdcv_shift_conv(int mode, const void* in, int in_ldc, const void* w, void* out, int out_ldc, const float* bias, const void* addsrc,
                    int add_ldc, float* stats, int B, int H, int W, int Cin, int Nout, const BnFuseArgs* fuse, hipStream_t st,
                    const EpiArgs* epi, int dil, const XAccArgs* xacc) {
  ShiftArgs a;
  a.xacc = XAccArgs{nullptr, 1};
  if (xacc) { if (mode != 0 || stats || epi || !xacc->acc || xacc->reps < 1 || (xacc->reps & (xacc->reps - 1))) return MDCV_EARG; a.xacc = *xacc; }
  if (fuse) a.fuse = *fuse; else a.fuse = BnFuseArgs{};
  if (epi) a.epi = *epi; else a.epi = EpiArgs{nullptr, 0, 0.f};
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addsrc = addsrc; a.stats = stats;
  a.in_ldc = in_ldc; a.out_ldc = out_ldc; a.add_ldc = add_ldc;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Nout = Nout;
  a.dil = dil; a.Wq = W + dil; a.Sq = (H + dil) * (W + dil); a.Mq = B * a.Sq;
  a.t2d = 0; a.tiles_x = a.tiles_y = a.TH = 0;
  if (mode == 3) {                                           // stride-2 data gradient: H, W are dY's; tiles of 8 x 31 dY positions + one halo row / column
    a.t2d = 2; a.Wq = T2D_WQ; a.TH = T2D_TH;
    a.tiles_x = (W + T2D_WQ - 2) / (T2D_WQ - 1); a.tiles_y = (H + T2D_TH - 1) / T2D_TH;
    a.Mq = B * a.tiles_x * a.tiles_y * 256; a.Sq = a.tiles_x * a.tiles_y * 256;
  } else if (shift_is_2d(W, Nout, dil)) {
    a.t2d = 1; a.Wq = T2D_WQ; a.TH = T2D_TH;
    a.tiles_x = (W + T2D_WQ - 3) / (T2D_WQ - 2); a.tiles_y = (H + T2D_TH - 1) / T2D_TH;
    a.Mq = (int)shift_2d_positions(B, H, W); a.Sq = a.tiles_x * a.tiles_y * 256;
  }
  a.tiles_n = Nout <= 64 ? 1 : Nout / BN;
  a.tiles_total = 0; a.xcd_chunk = 0; a.nca = 0; a.p_base = 0;
  a.nchunks = (Cin + 31) / 32;
  a.wrow = 9 * Cin;
  const unsigned in_bytes = (unsigned)((long long)B * H * W * in_ldc * 2);
  const unsigned w_bytes = (unsigned)((long long)Nout * 9 * Cin * 2);
  if (mode == 3 && (epi || stats || bias)) return MDCV_EARG;
  if (mode != 0) return mdcv_shift_launch_dgrad(a, st, in_bytes, w_bytes);
  if (Nout == 32) return launch_shift_mode<0, 32>(a, st, in_bytes, w_bytes);
  if (Nout == 64) return launch_shift_mode<0, 64>(a, st, in_bytes, w_bytes);
  return launch_shift_mode<0, 128>(a, st, in_bytes, w_bytes);
}

#ifdef MDCV_SHIFT_TS
extern "C" int mdcv_debug_shift_ts(long long* host4x512) {
  return (int)hipMemcpyFromSymbol(host4x512, HIP_SYMBOL(g_shift_ts), sizeof(long long) * 4 * 512);
}
#endif
#if defined(MDCV_SHIFT_TS) || defined(MDCV_SHIFT_WG)
extern "C" int mdcv_debug_shift_occ(int lds) {
  int nb = -1;
  auto kern = mdcv_conv3x3_shift_kernel<0, 256, 3, 3, false, 2>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 512, (size_t)lds);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
  printf("occupancy(lds=%d) = %d blocks/CU (err %d); numRegs %d sharedSizeBytes %zu maxThreadsPerBlock %d\n", lds, nb, (int)e, fa.numRegs,
         (size_t)fa.sharedSizeBytes, fa.maxThreadsPerBlock);
  return nb;
}
extern "C" int mdcv_debug_shift_wg(long long* host3x4096) {
  return (int)hipMemcpyFromSymbol(host3x4096, HIP_SYMBOL(g_shift_wg), sizeof(long long) * 3 * 4096);
}
#endif

#endif   // MDCV_SHIFT_PART

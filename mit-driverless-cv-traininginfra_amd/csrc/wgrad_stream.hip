// Weight gradient of a 3x3 / stride 1 / "same" convolution (dilation 1 or 2) for the 16..128-channel layers, with the activation
// window of ALL NINE taps resident in an LDS ring.
//
// Why: the generic kernel (conv_igemm.hip, conv_wgrad_dma_kernel / _narrow_kernel) gathers an im2col tile per tap, so every activation
// row travels through the global->LDS path nine times.  For C <= 64 that path, not HBM and not the MFMA pipe, sets the time
// (RektNet 80x80 16->16: 118 us for 105 MB of operands = 0.9 TB/s; 64->64: 287 TFLOP/s), and a 128-wide output-channel tile wastes
// half its MFMAs on Cout = 64.
//
// Here a block owns a run of stream positions and the WHOLE dW[Cout][9][Cin] (Cout, Cin <= 128; 128x128 stays with wgrad_shift.hip).
// Positions run over a padded 1-D stream, p = img*(H+d)(W+d) + y*(W+d) + x with d = dilation shared zero columns per row and d
// zero rows per image: dY'[p] is zero at junk positions, X'[p] is zero on the padding, so tap (kh,kw) of position p is simply
// X'[p + ((kh-1)(W+d) + (kw-1))*d] and no per-tap masking exists.  Per step of BP positions the block DMAs BP new dY rows into
// a stage buffer and BP new activation rows into a RING of RS >= (D+1)*BP + 2*hpad rows (a power of two) (hpad >= d(W+d+1) halo rows each side):
// every activation byte is fetched once per block (plus the 2*hpad-row window at the start of the block's run).
//
// Operands stay in [position][channel] order; MFMA fragments (8 positions per channel) come from ds_read_b64_tr_b16 transpose
// reads.  Rows are 32*NC bytes (NC = channels/16); the 32-byte unit u of row r is stored at unit u ^ ((r & 7) / (8/NC) & (NC-1)), which
// makes any 8 CONSECUTIVE rows (one LDS service group: lanes 0-31 read rows r..r+7 of one 16-channel block) cover all 64 banks
// for every start row, so the tap-shifted reads are conflict-free as well.  K slot (kq, half, i) <-> position kq*4 + i + 16*half.
//
// Waves: 8 = PG position groups x TG tile groups; a wave accumulates A co-tiles x 1 ci-tile x 9 taps (36*A VGPRs) over its share of
// the 32-position sub-steps.  The PG partial sums meet in LDS in a fixed order (deterministic), then go to the split's fp32 slab
// ws[split][Cout][9*Cin] that wgrad_reduce_kk_kernel<9> sums into OIHW.
#include "common.h"
#include "wgrad_stream.h"

#include <type_traits>

namespace {

constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

__device__ __forceinline__ void fast_divmod(int n, int d, float inv, int& q, int& r) {   // 0 <= n < 2^24
  q = (int)((float)n * inv);
  r = n - q * d;
  const int lt = r < 0;  q -= lt; r += lt ? d : 0;
  const int ge = r >= d; q += ge; r -= ge ? d : 0;
}

// Transpose read as inline asm.  With the builtin, the compiler (which cannot tell the ring slots apart) drains the LDS-DMA queue
// with s_waitcnt vmcnt(0) in front of every read that follows a DMA -- also across the loop back-edge -- so fill and compute never
// overlap.  The asm read is invisible to that pass; the kernel orders DMA and reads itself (counted vmcnt + barrier) and waits for
// the reads with wait_frags() below, whose "+v" operands make every MFMA depend on the wait.
template <int OFF> __device__ __forceinline__ s16x4_t lds_tr16(unsigned addr) {      // addr: LDS byte address; OFF: immediate offset
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// plain 4-byte LDS read as inline asm (the DMA-address table of the TBL forms): same reason, and the caller waits with wait_tbl() below
template <int OFF> __device__ __forceinline__ int lds_ld32(unsigned addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int NX, int NY> __device__ __forceinline__ void wait_tbl(int (&nx)[NX], int (&ny)[NY]) {     // all LDS reads of this wave have returned
  static_assert(NX >= 1 && NX <= 2 && NY >= 1 && NY <= 4 && NY != 3, "one or two (dY: up to four) DMA instructions per operand per wave per step");
  if constexpr (NX == 1 && NY == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(ny[0]) :: "memory");
  else if constexpr (NX == 1 && NY == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(ny[0]), "+v"(ny[1]) :: "memory");
  else if constexpr (NX == 2 && NY == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(ny[0]) :: "memory");
  else if constexpr (NX == 2 && NY == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(ny[0]), "+v"(ny[1]) :: "memory");
  else if constexpr (NX == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(ny[0]), "+v"(ny[1]), "+v"(ny[2]), "+v"(ny[3]) :: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(ny[0]), "+v"(ny[1]), "+v"(ny[2]), "+v"(ny[3]) :: "memory");
}
// wait until at most N of this wave's LDS reads are outstanding (they return in order); the MFMAs that use `f...` depend on it
template <int N> __device__ __forceinline__ void wait_lds(bf16x8_t& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lds(bf16x8_t (&fa)[1], bf16x8_t& f) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[0]), "+v"(f) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lds(bf16x8_t (&fa)[2], bf16x8_t& f) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(f) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lds(bf16x8_t (&fa)[8], bf16x8_t& f) {
  asm volatile("s_waitcnt lgkmcnt(%9)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]), "+v"(f) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lds(bf16x8_t (&fa)[4], bf16x8_t& f) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(f) : "n"(N) : "memory");
}

// NCI / NCO: 16-channel blocks of Cin / Cout.  A: co blocks per wave.  PG: position groups.  BP: positions per step (8 KiB of
// activations).  D: steps in flight behind the one being multiplied.  TGRP: taps staged together in the epilogue (9, 3 or 1).
// TILED: the block owns a 16*NCO x 9 x 16*NCI slice of a LARGER dW (Cout = tiles_co * 16*NCO, Cin = tiles_ci * 16*NCI): the 128..1024-channel
// layers of YOLOv3 at 52x52 / 26x26 / 13x13.  Same ring, same fragments; the DMA columns and the slab rows / columns carry the tile offset.
// 128 co x 64 ci x 9 taps per block fills 24 KiB per 64 positions = 393 FLOP per filled byte (generic 128 x 128 im2col tile: 64).
// NW: waves per block (8, or 4 for the light form: 64 co x 64 ci per block, one wave per SIMD).
// TBL: the DMA lanes take the pixel index of their stream position from a per-block TABLE in LDS (built once in the prologue: one entry per position of
// the block's run, JUNK = B*H*W for padding / foreign rows, which the buffer's range check turns into zeros) instead of stepping (x, y, image) forward
// and re-deriving the pixel row per DMA instruction.  Round 6: the light form's step was 72 MFMAs behind ~180 address / select instructions (ISA: eight
// v_mad_u64_u32 among them), issued by the only wave of its SIMD; with the table a DMA instruction costs one ds_read_b32 (prefetched a step ahead), one
// v_mad_u32_u24 and the m0 write.  Same lanes, same ring image: results are bit-identical to the stepping form.
// DIRECT (round 6): the block owns its 16*NCO x 9 x 16*NCI slice of dW OUTRIGHT -- one split, no slab, no reduce launch: the epilogue adds the PG position
// groups' partial sums in LDS (fixed order) in [co][ci][tap] order and writes whole OIHW rows of the gradient itself.  For layers with enough 64 co x 32 ci
// tiles to fill the chip in one split: YOLOv3's 13^2 512->1024 layers (256 tiles).
template <int NCI, int NCO, int A, int PG, int BP, int D, int TGRP, bool TILED = false, int NW = 8, bool TBL = false, bool DIRECT = false>
__global__ __launch_bounds__(NW * 64, (NW == 4 && NCO == 4) ? 2 : 1) void wgrad3x3_stream_kernel(WgradStreamArgs a, unsigned dy_bytes, unsigned x_bytes) {
  constexpr int RBX = NCI * 32, RBY = NCO * 32;            // row bytes
  constexpr int LPRX = 2 * NCI, LPRY = 2 * NCO;            // 16-byte slots (= DMA lanes) per row
  constexpr int RPIX = 64 / LPRX, RPIY = 64 / LPRY;        // rows per 1 KiB DMA instruction
  constexpr int RX = 8 / NCI, RY = 8 / NCO;                // rows per 256-byte bank period
  constexpr int DYI = BP / RPIY / NW;                      // dY DMA instructions per wave per step
  constexpr int XI = BP / RPIX / NW;                       // activation DMA instructions per wave per step
  constexpr int NI = XI + DYI;                             // DMA instructions per wave per step
  constexpr int NSUB = BP / 32 / PG;                       // 32-position sub-steps per wave per step
  constexpr int TG = NW / PG;
  constexpr int YSTAGE = BP * RBY;
  constexpr int CIN = NCI * 16, COUT = NCO * 16;
  static_assert(BP / RPIX == NW * XI && XI >= 1 && (NW * RPIY) % 8 == 0, "whole activation DMAs per wave per step");
  static_assert(TG == (NCO / A) * NCI && NSUB >= 1 && DYI >= 1, "wave grid");
  static_assert(9 % TGRP == 0, "tap groups");

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const ring = smem;                          // the activation ring sits at LDS offset 0, the dY stages behind it
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  int split = logical, tile_co = 0, tile_ci = 0;
  if constexpr (TILED) {
    static_assert(TGRP == 1, "tiled slabs are written one tap at a time");
    if (logical >= a.splits * a.tiles) return;
    split = logical / a.tiles;                               // the tiles of one split are neighbours on an XCD
    const int tl = logical - split * a.tiles;                //   (they read the same pixel rows)
    tile_co = tl / a.tiles_ci; tile_ci = tl - tile_co * a.tiles_ci;
  } else {
    if (logical >= a.splits) return;
  }
  const int slab_index = split;
  const void* dyp = a.dy;
  const void* xp = a.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave % PG, tg = wave / PG;
  const int cog = tg / NCI, cit = tg - cog * NCI;          // this wave's co blocks cog*A .. +A-1 and ci block
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dyp), 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xp), 0, x_bytes, 0x00020000);

  const int RS = a.RS, rmask = a.RS - 1;                    // ring rows: a power of two
  // The ring carries a MIRROR of its first MIR rows behind its end (like the 7x7 kernel): a tap's fragments of one step -- NSUB sub-steps of
  // 32 rows from one start row -- then never wrap, so a step computes ONE wrapped start address per tap and every read is that address +
  // an immediate offset (the per-read add / and pairs were 104 VALU instructions per step in front of 72 MFMAs on a one-wave-per-SIMD kernel).
  constexpr int MIR = NSUB * 32;
  unsigned char* const stages = smem + (RS + MIR) * RBX;
  const unsigned ybase = (unsigned)((RS + MIR) * RBX), bmask = (unsigned)(RS * RBX - 1);
  const int p_begin = split * a.pos_per_split;
  const int p_end = min(a.Mq, p_begin + a.pos_per_split);
  const int p_lo = p_begin - a.hpad;                        // stream position held by ring row 0
  const float inv_sq = 1.0f / (float)a.Sq, inv_wq = 1.0f / (float)a.Wq;
  const unsigned ldy2 = (unsigned)a.dy_ldc * 2u, lx2 = (unsigned)a.x_ldc * 2u;

  // DMA roles.  activations: lane -> row rrx of the chunk, slot sx; it fetches the logical 16-byte column sx ^ 2*g(row).
  const int rrx = lane / LPRX, sx = lane % LPRX;
  const int gxw = ((rrx & 7) / RX) & (NCI - 1);             // chunks start at ring rows that are multiples of 8
  const unsigned lane_x = (unsigned)((sx ^ (gxw << 1)) * 16) + (unsigned)(tile_ci * (NCI * 32));
  const int rry = lane / LPRY, sy = lane % LPRY;
  const int gyw = ((((wave * RPIY) + rry) & 7) / RY) & (NCO - 1);   // chunk c = wave + 8j starts at row c*RPIY; 8j*RPIY is a multiple of 8
  const unsigned lane_y = (unsigned)((sy ^ (gyw << 1)) * 16) + (unsigned)(tile_co * (NCO * 32));

  auto pix_off = [&](int p, unsigned ld2, bool ok) -> unsigned {   // stream position -> byte offset of its pixel row, or OOB (zero fill)
    ok = ok && p >= 0 && p < a.Mq;
    const int pp = ok ? p : 0;
    int img, rem, y, x;
    fast_divmod(pp, a.Sq, inv_sq, img, rem);
    fast_divmod(rem, a.Wq, inv_wq, y, x);
    ok = ok && x < a.W && y < a.H;
    return ok ? __umul24((unsigned)((img * a.H + y) * a.W + x), ld2) : OOB;
  };
  // one activation chunk: RPIX rows starting at stream position p (ring row rho, a multiple of RPIX); rows below MIR also go to the mirror
  auto issue_x = [&](int p, int rho) {
    const unsigned off = pix_off(p + rrx, lx2, true);
    const int vo = (int)(off == OOB ? OOB : off + lane_x);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + rho * RBX), 16, vo, 0, 0, 0);
    if (rho < MIR) {
      asm volatile("" ::: "memory");                         // (two LDS-DMA calls that differ only in the destination must not be folded)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + (RS + rho) * RBX), 16, vo, 0, 0, 0);
    }
  };
  // The steps are issued in order t = 0, 1, 2, ..., each exactly once, and every DMA lane's stream position advances by BP per step: the
  // lane keeps (x, y, image) of its row per DMA instruction and steps them forward (add, compare, conditional subtract) instead of two
  // float divisions per DMA per step.  That address code was ~180 VALU instructions per step IN FRONT of the step's MFMAs -- on the
  // one-wave-per-SIMD forms nothing else issues meanwhile, which is why "fill" and "compute" added up instead of overlapping.
  // Needs Hq > BP / Wq + 1 (one row wrap per step at most; the host only takes this kernel then).
  const int adv_q = BP / a.Wq, adv_r = BP - adv_q * a.Wq, Hq = a.Sq / a.Wq;
  int qx[XI], qy[XI], qi[XI], yx[DYI], yy[DYI], yi[DYI], yrem[DYI];
  auto seed = [&](int p, int& x, int& y, int& img) {        // p >= 0
    int rem;
    fast_divmod(p < (1 << 24) - 1 ? p : (1 << 24) - 1, a.Sq, inv_sq, img, rem);
    fast_divmod(rem, a.Wq, inv_wq, y, x);
  };
  const int nimg = a.Mq / a.Sq;
  if constexpr (!TBL) {
#pragma unroll
    for (int j = 0; j < XI; ++j) seed(p_begin + a.hpad + (wave + NW * j) * RPIX + rrx, qx[j], qy[j], qi[j]);
#pragma unroll
    for (int j = 0; j < DYI; ++j) {
      const int p = p_begin + (wave + NW * j) * RPIY + rry;
      seed(p, yx[j], yy[j], yi[j]);
      yrem[j] = p_end - p;                                   // > 0 while the row belongs to this split
    }
  }
  auto advance = [&](int& x, int& y, int& img) {
    x += adv_r;
    const int c1 = x >= a.Wq;
    x -= c1 ? a.Wq : 0;
    y += adv_q + c1;
    const int c2 = y >= Hq;
    y -= c2 ? Hq : 0;
    img += c2;
  };
  auto row_off = [&](int x, int y, int img, unsigned ld2, bool ok) -> unsigned {
    ok = ok && x < a.W && y < a.H && img < nimg;
    const unsigned pix = __umul24(__umul24((unsigned)img, (unsigned)a.H) + (unsigned)y, (unsigned)a.W) + (unsigned)x;
    return ok ? __umul24(pix, ld2) : OOB;
  };
  // step t (called for t = 0, 1, 2, ... in order): the BP new activation rows (p_begin + t*BP + hpad ..) and the BP dY rows of the step
  auto issue = [&](int t, int rho_new) {
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int chunk = wave + NW * j;
      const int rho = (rho_new + chunk * RPIX) & rmask;
      const unsigned off = row_off(qx[j], qy[j], qi[j], lx2, true);
      const int vo = (int)(off == OOB ? OOB : off + lane_x);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + rho * RBX), 16, vo, 0, 0, 0);
      if (rho < MIR) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + (RS + rho) * RBX), 16, vo, 0, 0, 0);
      }
      advance(qx[j], qy[j], qi[j]);
    }
    unsigned char* sY = stages + (t % (D + 1)) * YSTAGE;
#pragma unroll
    for (int j = 0; j < DYI; ++j) {
      const int chunk = wave + NW * j;
      const unsigned off = row_off(yx[j], yy[j], yi[j], ldy2, yrem[j] > 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(sY + chunk * 1024), 16, (int)(off == OOB ? OOB : off + lane_y), 0, 0, 0);
      advance(yx[j], yy[j], yi[j]);
      yrem[j] -= BP;
    }
  };

  // ---- TBL forms: tbl[i] = pixel index of stream position p_lo + i (JUNK = B*H*W: byte offset == buffer size, so the range check zero-fills the row).
  // ex / ey always hold the (complete) entries of the NEXT step to be issued; the entries of the one after are requested in front of a step's DMA issue
  // and waited for behind it, before the step's first fragment read.
  const unsigned tbl_base = ybase + (unsigned)((D + 1) * YSTAGE);
  int ex[XI], ey[DYI];
  unsigned tx_addr = 0, ty_addr = 0;
  static_assert(!TBL || (XI <= 2 && (DYI <= 2 || DYI == 4)), "table reads are unrolled by hand");
  int tbl_s0 = 0;                                             // first step of the table's current window (a.tbl_steps steps; long runs rebuild it)
  auto fetch = [&](int t, int (&nx)[XI], int (&ny)[DYI]) {
    const unsigned ax = tx_addr + (unsigned)((t - tbl_s0) * (BP * 4)), ay = ty_addr + (unsigned)((t - tbl_s0) * (BP * 4));
    nx[0] = lds_ld32<0>(ax);
    if constexpr (XI > 1) nx[1] = lds_ld32<NW * RPIX * 4>(ax);
    ny[0] = lds_ld32<0>(ay);
    if constexpr (DYI > 1) ny[1] = lds_ld32<NW * RPIY * 4>(ay);
    if constexpr (DYI > 2) { ny[2] = lds_ld32<2 * NW * RPIY * 4>(ay); ny[3] = lds_ld32<3 * NW * RPIY * 4>(ay); }
  };
  auto issue_tbl = [&](int t, int rho_new) {                 // step t from ex / ey
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int chunk = wave + NW * j;
      const int rho = (rho_new + chunk * RPIX) & rmask;
      const int vo = (int)(__umul24((unsigned)ex[j], lx2) + lane_x);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + rho * RBX), 16, vo, 0, 0, 0);
      if (rho < MIR) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + (RS + rho) * RBX), 16, vo, 0, 0, 0);
      }
    }
    unsigned char* sY = stages + (t % (D + 1)) * YSTAGE;
#pragma unroll
    for (int j = 0; j < DYI; ++j) {
      const int chunk = wave + NW * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(sY + chunk * 1024), 16, (int)(__umul24((unsigned)ey[j], ldy2) + lane_y), 0, 0, 0);
    }
  };
  // window of the table that starts at step s0: entry i <-> stream position p_lo + s0 * BP + i, i < tbl_steps * BP + 2 * hpad (the dY rows of a step
  // sit hpad entries, its new activation rows 2 * hpad entries behind the step's first entry).  Called by all waves with no table read in flight.
  auto build_tbl = [&](int s0) {
    int* const tbl = reinterpret_cast<int*>(smem + tbl_base);
    const int ntbl = a.tbl_steps * BP + 2 * a.hpad, junk = nimg * a.H * a.W, pw = p_lo + s0 * BP;
    for (int i = tid; i < ntbl; i += NW * 64) {
      const int p = pw + i;
      bool ok = p >= 0 && p < a.Mq;
      int img, rem, y, x;
      fast_divmod(ok ? p : 0, a.Sq, inv_sq, img, rem);
      fast_divmod(rem, a.Wq, inv_wq, y, x);
      ok = ok && x < a.W && y < a.H;
      tbl[i] = ok ? (img * a.H + y) * a.W + x : junk;
    }
    __syncthreads();
    tbl_s0 = s0;
  };
  if constexpr (TBL) {
    build_tbl(0);
    tx_addr = tbl_base + 4u * (unsigned)(2 * a.hpad + wave * RPIX + rrx);
    ty_addr = tbl_base + 4u * (unsigned)(a.hpad + wave * RPIY + rry);
  }

  // fragment roles
  const int t16 = lane & 15, kq = lane >> 4;
  const int prow = kq * 4 + (t16 >> 2);
  const int sub = (t16 & 1) * 8, qlo = (t16 & 3) >> 1;
  // Every fragment address is  (wave-uniform byte offset) + (lane constant): sub-steps start at ring rows that are multiples of 32,
  // so the row-dependent swizzle of tap k depends only on (displacement_k + prow) & 7 -- a per-tap lane constant.
  const int gy = ((prow & 7) / RY) & (NCO - 1);
  unsigned laneY[A];                                          // dY: row prow of the sub-step, co block cog*A + i
#pragma unroll
  for (int i = 0; i < A; ++i) laneY[i] = ybase + (unsigned)(prow * RBY + ((((2 * (cog * A + i) + qlo) ^ (gy << 1)) << 4) + sub));
  const int xcol = ((2 * cit + qlo) << 4) + sub;            // byte column of this lane's 8 bytes before the swizzle
  unsigned laneX[9];                                          // activations: row prow + displacement, ci block cit
  int dtap[9];                                                // ring-row displacement of each tap, reduced to [0, RS)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    dtap[tap] = (((kh - 1) * a.Wq + (kw - 1)) * a.dil) & rmask;       // |displacement| <= hpad < RS
    const int g = (((dtap[tap] + prow) & 7) / RX) & (NCI - 1);
    laneX[tap] = (unsigned)((dtap[tap] + prow) * RBX + (xcol ^ (g << 5)));
  }
  // per step: xa[tap] = wrapped byte address of this lane's row of the step's first sub-step at tap `tap` (rows up to MIR - 1 further on are
  // contiguous thanks to the mirror); ya[i] = the same in the step's dY stage (linear, no wrap)
  unsigned xa[9], ya[A];
  auto step_addrs = [&](int t, int rho0) {
    const unsigned ystage = (unsigned)((t % (D + 1)) * YSTAGE + pg * NSUB * 32 * RBY);
    const unsigned xs = (unsigned)((rho0 + pg * NSUB * 32) * RBX);
#pragma unroll
    for (int i = 0; i < A; ++i) ya[i] = ystage + laneY[i];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) xa[tap] = (xs + laneX[tap]) & bmask;
  };
  auto fragY = [&](auto N, int i) -> bf16x8_t {                // N: sub-step of the step (integral_constant)
    constexpr int o = decltype(N)::value * 32 * RBY;
    const s16x4_t lo = lds_tr16<o>(ya[i]);
    const s16x4_t hi = lds_tr16<o + 16 * RBY>(ya[i]);
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto fragX = [&](auto N, int tap) -> bf16x8_t {
    constexpr int o = decltype(N)::value * 32 * RBX;
    const s16x4_t lo = lds_tr16<o>(xa[tap]);
    const s16x4_t hi = lds_tr16<o + 16 * RBX>(xa[tap]);      // 16 rows on: same row & 7, same swizzle
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  f32x4_t acc[9][A];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < A; ++i) acc[k][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = (p_end - p_begin + BP - 1) / BP;
  // A sub-step is  reads (A + 9 fragments: address VALU + LDS)  then  9*A MFMAs.  The block's barrier keeps all waves in the same
  // step, and the two waves that share a SIMD (w and w+4) would read together and multiply together, leaving the MFMA pipe idle
  // during every read phase.  So waves 4..7 run half a sub-step late: they multiply the fragments they read in the previous slot
  // first, then read -- one of the pair reads while the other multiplies.  No extra registers: the fragments simply live across
  // the barrier (the data they were read from may be overwritten afterwards).
  // The DMA of step t+D is issued after the step's last reads in program order: the compiler drains the LDS-DMA queue
  // (s_waitcnt vmcnt(0)) in front of any LDS read it can see after a DMA, which would serialise fill and compute.
  bf16x8_t fa[A], fb[9];
#pragma unroll
  for (int i = 0; i < A; ++i) fa[i] = bf16x8_t{};
#pragma unroll
  for (int k = 0; k < 9; ++k) fb[k] = bf16x8_t{};
  constexpr bool kMfma = true;                                 // (wgrad_stream_pipe.inc is generated with this switch)
  auto reads = [&](auto N) {                                   // sub-step N of the step whose addresses step_addrs() prepared
#pragma unroll
    for (int i = 0; i < A; ++i) fa[i] = fragY(N, i);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) fb[tap] = fragX(N, tap);
  };
  auto mfmas = [&]() {
    // reads return in issue order (fa..., fb[0], fb[1], ...): tap k starts as soon as its fragment is in, the rest keeps streaming
    wait_lds<15>(fa, fb[0]);                               // (the counter has 4 bits)
    if constexpr (kMfma)
#pragma unroll
    for (int i = 0; i < A; ++i) acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[0], acc[0][i], 0, 0, 0);
#define MDCV_TAP(K)                                                                                            \
    wait_lds<16 - 2 * K>(fb[K]);                                                                               \
    if constexpr (kMfma) _Pragma("unroll") for (int i = 0; i < A; ++i)                                         \
      acc[K][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[K], acc[K][i], 0, 0, 0);
    MDCV_TAP(1) MDCV_TAP(2) MDCV_TAP(3) MDCV_TAP(4) MDCV_TAP(5) MDCV_TAP(6) MDCV_TAP(7) MDCV_TAP(8)
#undef MDCV_TAP
  };
  const bool late = NSUB == 1 && wave >= NW / 2;             // (two sub-steps per step: the skewed schedule spills)
  // channel-tiled instantiation: the reads run a few fragments ahead of the multiplies (wgrad_stream_pipe.inc, scripts/gen_wgrad_pipeline.py)
  constexpr bool kPipe = TILED && NSUB == 2 && (PG == 1 || PG == 2) && A == 4;
  bf16x8_t fa2[2][A];
  if constexpr (kPipe) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < A; ++i) fa2[n][i] = bf16x8_t{};
  }
  auto step = [&](int t, int rho0, bool more, int t_new, int rho_new) {
    // (the asm reads are invisible to the compiler's LDS-DMA hazard tracking, so the DMA may sit anywhere; it sits where no
    //  fragment is live, to keep the address arithmetic of the DMA out of the 256-register budget)
    if (late) mfmas();                                       // the previous slot's fragments (zeros the first time)
    if constexpr (TBL) {
      const bool next = more && t_new + 1 < nt;
      int nx[XI], ny[DYI];
      if (next && t_new + 1 - tbl_s0 == a.tbl_steps) build_tbl(t_new + 1);     // (block-uniform; every wave's table reads were waited for in its previous step)
      if (next) fetch(t_new + 1, nx, ny);                    // entries of the step after the one issued now: they land under the DMA issue
      if (more) issue_tbl(t_new, rho_new);
      step_addrs(t, rho0);
      if (next) {
        wait_tbl(nx, ny);
#pragma unroll
        for (int j = 0; j < XI; ++j) ex[j] = nx[j];
#pragma unroll
        for (int j = 0; j < DYI; ++j) ey[j] = ny[j];
      }
    } else {
      if (more) issue(t_new, rho_new);
      step_addrs(t, rho0);
    }
    if constexpr (kPipe) {
      using N0 = std::integral_constant<int, 0>;
      using N1 = std::integral_constant<int, 1>;
#include "wgrad_stream_pipe.inc"
    } else {
      static_assert(NSUB <= 4, "sub-steps are unrolled by hand below");
      reads(std::integral_constant<int, 0>{});
      if (!late) mfmas();
      if constexpr (NSUB > 1) { if (late) mfmas(); reads(std::integral_constant<int, 1>{}); if (!late) mfmas(); }
      if constexpr (NSUB > 2) { if (late) mfmas(); reads(std::integral_constant<int, 2>{}); if (!late) mfmas(); }
      if constexpr (NSUB > 3) { if (late) mfmas(); reads(std::integral_constant<int, 3>{}); if (!late) mfmas(); }
    }
  };

  // prologue: the 2*hpad rows below the first step's new rows, then D steps ahead
  int issued = 0;
  int rho_new = (2 * a.hpad) & rmask;                        // ring row of the next step's first new activation row
  if constexpr (TBL) {
    const int nch = 2 * a.hpad / RPIX;
    for (int c = wave; c < nch; c += NW) {
      int e[1] = {lds_ld32<0>(tbl_base + 4u * (unsigned)(c * RPIX + rrx))}, e2[1] = {0};
      wait_tbl(e, e2);
      const int vo = (int)(__umul24((unsigned)e[0], lx2) + lane_x), rho = c * RPIX;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + rho * RBX), 16, vo, 0, 0, 0);
      if (rho < MIR) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + (RS + rho) * RBX), 16, vo, 0, 0, 0);
      }
    }
    if (nt > 0) { fetch(0, ex, ey); wait_tbl(ex, ey); }
    for (; issued < D && issued < nt; ++issued) {
      issue_tbl(issued, rho_new);
      rho_new = (rho_new + BP) & rmask;
      if (issued + 1 < nt) { fetch(issued + 1, ex, ey); wait_tbl(ex, ey); }
    }
  } else {
    const int nch = 2 * a.hpad / RPIX;
    for (int c = wave; c < nch; c += NW) issue_x(p_lo + c * RPIX, c * RPIX);
    for (; issued < D && issued < nt; ++issued) {
      issue(issued, rho_new);
      rho_new = (rho_new + BP) & rmask;
    }
  }
  int rho0 = a.hpad;
  for (int t = 0; t < nt; ++t) {
    // lgkmcnt(0): the waves that run half a sub-step late carry fragment READS across this barrier (they multiply them behind it), and the
    // DMA issued behind the barrier refills the dY stage -- and, when the ring has less than one step of slack, activation rows -- those
    // reads come from.  ds_reads are served in issue order against other waves' ds_writes, not against an LDS-DMA write: the reads must have
    // RETURNED before any wave passes the barrier (same hazard as in conv_shift.hip; scripts/check_ring_barriers.py checks the built code).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (its own statement: on every path to the barrier, whichever vmcnt wait is taken)
    if (issued - 1 - t >= D - 1 && D > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = issued < nt;
    step(t, rho0, more, issued, rho_new);
    if (more) {
      ++issued;
      rho_new = (rho_new + BP) & rmask;
    }
    rho0 = (rho0 + BP) & rmask;
  }
  if (late) mfmas();                                         // the last slot of the late waves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // epilogue: the PG partial sums meet in LDS (fixed order), TGRP taps at a time, then coalesced rows of the split's slab
  float* so = reinterpret_cast<float*>(smem);
  if constexpr (DIRECT) {
    static_assert(TILED && TGRP == 1, "the direct form is a tiled form");
    constexpr int PR = CIN * 9 + 4;                          // floats per staged output-channel row: [ci][tap] + pad
#pragma unroll
    for (int pgi = 0; pgi < PG; ++pgi) {
      if (pg == pgi) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int i = 0; i < A; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int idx = ((cog * A + i) * 16 + kq * 4 + rr) * PR + (cit * 16 + t16) * 9 + tap;
              const float v = acc[tap][i][rr];
              so[idx] = pgi == 0 ? v : so[idx] + v;
            }
      }
      __syncthreads();
    }
    // whole OIHW rows: output channel co of the tile -> CIN * 9 contiguous floats at dw[co][tile_ci * CIN][0][0]
    float* __restrict__ dw = a.dw + ((size_t)tile_co * COUT * a.Cin_real + (size_t)tile_ci * CIN) * 9;
    if (a.Cin_real == a.Cin && a.Cout_real == a.Cout) {
      constexpr int V4 = CIN * 9 / 4;
      for (int v = tid; v < COUT * V4; v += NW * 64) {
        const int row = v / V4, c4 = (v - row * V4) * 4;
        float4 val = *reinterpret_cast<const float4*>(so + row * PR + c4);
        float4* dst = reinterpret_cast<float4*>(dw + (size_t)row * a.Cin_real * 9 + c4);
        if (a.accumulate) { const float4 o = *dst; val.x += o.x; val.y += o.y; val.z += o.z; val.w += o.w; }
        *dst = val;
      }
    } else {                                                 // padded channels are cropped: element by element
      for (int v = tid; v < COUT * CIN * 9; v += NW * 64) {
        const int row = v / (CIN * 9), c = v - row * (CIN * 9), ci = c / 9;
        if (tile_co * COUT + row < a.Cout_real && tile_ci * CIN + ci < a.Cin_real) {
          float* dst = dw + (size_t)row * a.Cin_real * 9 + c;
          *dst = so[row * PR + c] + (a.accumulate ? *dst : 0.f);
        }
      }
    }
    return;
  }
  constexpr int OR = TGRP * CIN + 4;                         // floats per staged row
  float* __restrict__ ws = a.ws + ((size_t)slab_index * a.Cout + (size_t)tile_co * COUT) * a.Ktot + tile_ci * CIN;
#pragma unroll
  for (int g0 = 0; g0 < 9 / TGRP; ++g0) {
#pragma unroll
    for (int pgi = 0; pgi < PG; ++pgi) {
      if (pg == pgi) {
#pragma unroll
        for (int tl = 0; tl < TGRP; ++tl)
#pragma unroll
          for (int i = 0; i < A; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int idx = ((cog * A + i) * 16 + kq * 4 + rr) * OR + tl * CIN + cit * 16 + t16;
              const float v = acc[g0 * TGRP + tl][i][rr];
              so[idx] = pgi == 0 ? v : so[idx] + v;
            }
      }
      __syncthreads();
    }
    constexpr int V4 = TGRP * CIN / 4;                       // float4 per staged row
    for (int v = tid; v < COUT * V4; v += NW * 64) {
      const int row = v / V4, c4 = (v - row * V4) * 4;
      *reinterpret_cast<float4*>(ws + (size_t)row * a.Ktot + g0 * TGRP * a.Cin + c4) = *reinterpret_cast<const float4*>(so + row * OR + c4);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// RektNet's stem: 7x7 / stride 1 / pad 3, 3 input channels (padded to 16 by the caller for this kernel) -> 16 output channels.
// The im2col kernel filled 49 x the activations (163 us alone, 325 us inside the step, for a 79 MB problem).  Same stream / ring scheme
// as above with three shared zero columns and rows; 16-channel rows (32 bytes: 8 consecutive rows = one bank period, no swizzle
// needed); a wave owns one 32-position sub-step per 256-position step and all 49 taps (196 accumulator VGPRs): per kernel row it
// reads 7 fragments, then 7 MFMAs.  The ring carries a 32-row MIRROR of its first rows behind its end, so a fragment's rows
// r .. r+22 never wrap and the 14 reads of a kernel row are ONE address + immediate offsets (the per-read wrap arithmetic was the
// largest cost of the first version).  The 8 per-wave partial sums meet in LDS in fixed order; slab ws[split][16][49*16], summed by
// the generic slab reduce.
__global__ __launch_bounds__(512) void wgrad7x7_stream_kernel(WgradStreamArgs a, unsigned dy_bytes, unsigned x_bytes) {
  constexpr int RB = 32, RPI = 32, BP = 256, YSTAGE = BP * RB, KT = 7, NT = KT * KT, MIRROR = 32;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.splits) return;
  const int split = logical;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, x_bytes, 0x00020000);
  const int RS = a.RS, rmask = a.RS - 1;
  unsigned char* const ring = smem;                          // RS + MIRROR rows
  unsigned char* const stages = smem + (RS + MIRROR) * RB;
  const unsigned ybase = (unsigned)((RS + MIRROR) * RB);
  const int p_begin = split * a.pos_per_split;
  const int p_end = min(a.Mq, p_begin + a.pos_per_split);
  const int p_lo = p_begin - a.hpad;
  const float inv_sq = 1.0f / (float)a.Sq, inv_wq = 1.0f / (float)a.Wq;
  const unsigned ldy2 = (unsigned)a.dy_ldc * 2u, lx2 = (unsigned)a.x_ldc * 2u;
  const int rr = lane >> 1;                                  // DMA: lane -> row rr of the 32-row chunk, 16-byte half lane & 1
  const unsigned lane_c = (unsigned)((lane & 1) * 16);
  auto pix_off = [&](int p, unsigned ld2, bool ok) -> unsigned {
    ok = ok && p >= 0 && p < a.Mq;
    const int pp = ok ? p : 0;
    int img, rem, y, x;
    fast_divmod(pp, a.Sq, inv_sq, img, rem);
    fast_divmod(rem, a.Wq, inv_wq, y, x);
    ok = ok && x < a.W && y < a.H;
    return ok ? __umul24((unsigned)((img * a.H + y) * a.W + x), ld2) : OOB;
  };
  auto issue_x = [&](int p, int rho) {                       // 32 rows at ring row rho (a multiple of 32); ring row 0's chunk also feeds the mirror
    const unsigned off = pix_off(p + rr, lx2, true);
    const int vo = (int)(off == OOB ? OOB : off + lane_c);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + rho * RB), 16, vo, 0, 0, 0);
    if (rho == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + RS * RB), 16, vo, 0, 0, 0);
  };
  auto issue = [&](int t, int rho_new) {                     // step t: 256 new activation rows and 256 dY rows, one chunk of each per wave
    const int p0 = p_begin + t * BP;
    issue_x(p0 + a.hpad + wave * RPI, (rho_new + wave * RPI) & rmask);
    const int p = p0 + wave * RPI + rr;
    const unsigned off = pix_off(p, ldy2, p < p_end);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(stages + (t & 1) * YSTAGE + wave * 1024), 16,
                                             (int)(off == OOB ? OOB : off + lane_c), 0, 0, 0);
  };
  const int t16 = lane & 15, kq = lane >> 4;
  const int prow = kq * 4 + (t16 >> 2);
  const unsigned lcol = (unsigned)((((t16 & 3) >> 1) << 4) + (t16 & 1) * 8);     // 8 bytes of the 32-byte row

  f32x4_t acc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nt = (p_end - p_begin + BP - 1) / BP;
  {
    const int nch = 2 * a.hpad / RPI;                        // the window below the first step's new rows
    for (int c = wave; c < nch; c += 8) issue_x(p_lo + c * RPI, c * RPI);
  }
  if (nt > 0) issue(0, (2 * a.hpad) & rmask);
  int rho0 = a.hpad & rmask, rho_new = (2 * a.hpad + BP) & rmask;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (two steps in flight measured the same: the step is read/VALU-bound)
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nt) issue(t + 1, rho_new);
    rho_new = (rho_new + BP) & rmask;
    // this wave's sub-step: positions p0 + 32*wave .. +31
    const unsigned ya = ybase + (unsigned)((t & 1) * YSTAGE + (wave * 32 + prow) * RB) + lcol;
    bf16x8_t fa;
    {
      const s16x4_t lo = lds_tr16<0>(ya), hi = lds_tr16<16 * RB>(ya);
      const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      fa = __builtin_bit_cast(bf16x8_t, v);
    }
    const int rs = rho0 + wave * 32 + prow - 3;              // ring row of this lane's first position at tap (kh = 3, kw = 0)
#pragma unroll
    for (int kh = 0; kh < KT; ++kh) {
      bf16x8_t fb[KT];
      const unsigned ad = (unsigned)(((rs + (kh - 3) * a.Wq) & rmask) * RB) + lcol;     // rows ad .. ad+22 are contiguous thanks to the mirror
#define MDCV_RD(KW)                                                                                        \
      { const s16x4_t lo = lds_tr16<(KW) * RB>(ad), hi = lds_tr16<((KW) + 16) * RB>(ad);                   \
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                        \
        fb[KW] = __builtin_bit_cast(bf16x8_t, v); }
      MDCV_RD(0) MDCV_RD(1) MDCV_RD(2) MDCV_RD(3) MDCV_RD(4) MDCV_RD(5) MDCV_RD(6)
#undef MDCV_RD
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fb[4]), "+v"(fb[5]), "+v"(fb[6]) :: "memory");
#pragma unroll
      for (int kw = 0; kw < KT; ++kw) acc[kh * KT + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[kw], acc[kh * KT + kw], 0, 0, 0);
    }
    rho0 = (rho0 + BP) & rmask;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // The 8 waves' partial sums (49 tiles of 16x16 each) meet in LDS one kernel row at a time: every wave stores its 7 tiles, then all
  // 512 threads add the eight copies in fixed order (w = 0..7) and write the 16 x 112 floats of that kernel row to the split's slab.
  // (Eight serialised accumulate rounds over all 49 tiles, as in the 3x3 kernel, cost 25-40 us here: 196 read-modify-writes per
  //  lane per round with seven waves idle.)
  float* so = reinterpret_cast<float*>(smem);                // [8 waves][7 kw][16 co][16 ci]
  float* __restrict__ ws = a.ws + (size_t)split * 16 * a.Ktot;
#pragma unroll
  for (int kh = 0; kh < KT; ++kh) {
#pragma unroll
    for (int kw = 0; kw < KT; ++kw)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) so[((wave * KT + kw) * 16 + kq * 4 + r4) * 16 + t16] = acc[kh * KT + kw][r4];
    __syncthreads();
    for (int v = tid; v < 16 * KT * 16; v += 512) {          // v -> (co, kw, ci): 112 contiguous floats per output channel
      const int co = v / (KT * 16), rem = v - co * (KT * 16), kw = rem >> 4, ci = rem & 15;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += so[((w * KT + kw) * 16 + co) * 16 + ci];
      ws[(size_t)co * a.Ktot + (kh * KT + kw) * 16 + ci] = t;
    }
    __syncthreads();
  }
}

struct StreamCfg { int nci, nco, a, pg, bp, d, tgrp; bool tiled = false; int nw = 8; bool direct = false; };
// Tuning hooks (mdcv_conv2d_wgrad_set_variant): ONE per decision that was measured on the training step.
// Light form of the tiled instantiation: 64 co x 64 ci per block, FOUR waves (one per SIMD, 237 VGPRs), prefetch depth 1, 56 KiB of LDS -- about
// half a CU, on 256 blocks.  The 8-wave form owns its CU outright (2 x 234 VGPRs per SIMD, 120 KiB) and runs on 128 blocks so that the main
// stream keeps the other half of the chip; the light form leaves room for main-stream workgroups on EVERY CU instead, and their waves fill
// the issue slots its single wave per SIMD leaves open.  Same-box A/B of the YOLOv3 step: 14.45 -> 14.32 ms (256 blocks; 192: 14.62, 224:
// 14.35, 288: 14.39, 320: 14.54, 384: 14.77; depth 2: 14.60).  Layers with long position streams stay on the 8-wave form (RektNet's 80^2 x
// 256-image layers, 1.68 M positions: -1.3 % with the light form in round 2, neutral since the round-3 address rework).

// (Cin, Cout) -> instantiation; false if unsupported
inline bool stream_cfg(int Cin, int Cout, StreamCfg& c, long long Mq = 0) {     // Mq: padded stream positions (0: unknown -> the 8-wave form)
  if (Cin == 16 && Cout == 16)       c = {1, 1, 1, 8, 256, 2, 9};
  else if (Cin == 16 && Cout == 32)  c = {1, 2, 2, 8, 256, 2, 9};
  else if (Cin == 32 && Cout == 32)  c = {2, 2, 2, 4, 128, 2, 9};
  else if (Cin == 32 && Cout == 64)  c = {2, 4, 4, 4, 128, 2, 3};
  else if (Cin == 64 && Cout == 64)  c = {4, 4, 4, 2, 64, 2, 3};
  else if (Cin == 64 && Cout == 128 && !TUNE().stream_tiled) c = {4, 8, 4, 1, 64, 2, 1};
  else if (TUNE().stream_tiled && Cin % 64 == 0 && Cout % 128 == 0 && Cin <= 4096 && Cout <= 4096) {
    c = {4, 8, 4, 1, 64, 2, 1}; c.tiled = true;
    if (Mq > 0 && Mq <= TUNE().stream_light_maxpos) { c.nco = 4; c.nw = 4; c.d = TUNE().stream_light_depth; }   // light form
    // slab-free form: enough 64 co x 32 ci tiles to fill the chip with ONE split (13^2 512 -> 1024: 256), short position runs (the table window
    // and the [64][32 * 9] staging fit beside the ring)
    const int t32 = (Cout / 64) * (Cin / 32);
    if (c.nw == 4 && TUNE().stream_direct && TUNE().stream_table && t32 >= 192 && t32 <= 512 && Mq <= 16384) { c = {2, 4, 4, 2, 128, 1, 1}; c.tiled = true; c.nw = 4; c.direct = true; }
  }
  else return false;
  return true;
}

inline int stream_hpad(int W, int dil) { return (dil * (W + dil + 1) + 31) / 32 * 32; }
inline int stream_ring_rows(const StreamCfg& c, int W, int dil) {       // >= (D+1)*BP + 2*hpad, a power of two
  int need = (c.d + 1) * c.bp + 2 * stream_hpad(W, dil), rs = 256;
  while (rs < need) rs *= 2;
  return rs;
}
inline int stream_lds(const StreamCfg& c, int W, int dil, int tbl_bytes = 0) {
  const int rs = stream_ring_rows(c, W, dil);
  const int ring = (c.d + 1) * c.bp * c.nco * 32 + (rs + c.bp / c.pg) * c.nci * 32 + tbl_bytes;     // dY stages + ring + its mirror (one wave's sub-steps of a step) + the DMA-address table
  const int stage_out = c.direct ? c.nco * 16 * (c.nci * 16 * 9 + 4) * 4 : c.nco * 16 * (c.tgrp * c.nci * 16 + 4) * 4;
  return ring > stage_out ? ring : stage_out;
}
// TBL forms: one int per stream position of a block's run plus the halo on both sides
// (runs longer than 4096 positions rebuild it window by window: tbl_steps steps of BP positions each)
inline int stream_tbl_steps(int pos_per_split, int bp) { const int nt = pos_per_split / bp, cap = 4096 / bp; return nt < cap ? nt : cap; }
inline int stream_tbl_bytes(int pos_per_split, int bp, int hpad) { return (stream_tbl_steps(pos_per_split, bp) * bp + 2 * hpad) * 4; }

// configuration for a layer geometry: the prefetch depth shrinks until ring + stages fit the 160 KiB of a CU
inline bool stream_cfg_geom(int Cin, int Cout, int W, int dil, StreamCfg& c, long long Mq = 0) {
  if (!stream_cfg(Cin, Cout, c, Mq)) return false;
  while (c.d > 1 && stream_lds(c, W, dil) > 160 * 1024) --c.d;
  return stream_lds(c, W, dil) <= 160 * 1024;
}

template <int NCI, int NCO, int A, int PG, int BP, int D, int TGRP, bool TILED, int NW, bool TBL, bool DIRECT = false>
int launch_stream1(const WgradStreamArgs& a, int lds, unsigned dyb, unsigned xb, hipStream_t st) {
  static DynLds dyn_lds;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(wgrad3x3_stream_kernel<NCI, NCO, A, PG, BP, D, TGRP, TILED, NW, TBL, DIRECT>), lds); e != hipSuccess)
    return (int)e;
  MDCV_LAUNCH((wgrad3x3_stream_kernel<NCI, NCO, A, PG, BP, D, TGRP, TILED, NW, TBL, DIRECT>), dim3((unsigned)(a.xcd_chunk * 8)), dim3(NW * 64), lds, st, a, dyb, xb);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}
template <int NCI, int NCO, int A, int PG, int BP, int D, int TGRP, bool TILED = false, int NW = 8>
int launch_stream(const WgradStreamArgs& a, int lds, bool tbl, unsigned dyb, unsigned xb, hipStream_t st) {
  return tbl ? launch_stream1<NCI, NCO, A, PG, BP, D, TGRP, TILED, NW, true>(a, lds, dyb, xb, st)
             : launch_stream1<NCI, NCO, A, PG, BP, D, TGRP, TILED, NW, false>(a, lds, dyb, xb, st);
}

}  // namespace

bool mdcv_wgrad_stream_eligible(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                                long long dy_ldc, long long x_ldc) {
  if (dtype != MDCV_BF16 || KH != 3 || KW != 3 || stride != 1 || pad != dil || (dil != 1 && dil != 2)) return false;
  StreamCfg c;
  if (H < 4 || W < 4) return false;
  const long long Mq = (long long)B * (H + dil) * (W + dil);
  if (!stream_cfg_geom(Cin, Cout, W, dil, c, Mq)) return false;
  if (c.bp / (W + dil) + 1 >= H + dil) return false;         // the DMA lanes step (x, y, image) forward by BP positions per step: one image wrap at most
  if (Mq + 4096 >= (1LL << 24) || (long long)B * H * W >= (1LL << 24)) return false;          // 24-bit multiplies / float divmod
  if ((long long)B * H * W * dy_ldc * 2 >= (1LL << 31) || (long long)B * H * W * x_ldc * 2 >= (1LL << 31)) return false;
  if (dy_ldc >= (1 << 23) || x_ldc >= (1 << 23)) return false;
  return true;
}

// blocks = splits: two resident blocks per CU for the light configurations (A <= 2), one for A = 4; never less than 4 steps per split
int mdcv_wgrad_stream_splits(int B, int H, int W, int Cin, int Cout, int dil) {
  StreamCfg c;
  const int Mq = B * (H + dil) * (W + dil);
  if (!stream_cfg(Cin, Cout, c, Mq)) return 1;
  if (c.direct) return 1;                                    // the slab-free form: every block owns its slice of dW outright
  int s = TUNE().stream_blocks > 0 ? TUNE().stream_blocks : (c.a <= 2 ? 512 : 256);
  if (c.tiled) s = ((TUNE().stream_blocks > 0 ? TUNE().stream_blocks : (c.nco == 4 ? TUNE().stream_light_blocks : TUNE().stream_tiled_blocks)) + (Cout / (16 * c.nco)) * (Cin / 64) - 1) / ((Cout / (16 * c.nco)) * (Cin / 64));   // blocks = splits x channel tiles
  const int max_s = (Mq + c.bp * 4 - 1) / (c.bp * 4);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  const int pps = ((Mq + s - 1) / s + c.bp - 1) / c.bp * c.bp;
  return (Mq + pps - 1) / pps;
}

bool mdcv_wgrad_stream_splits_ok(int splits, int B, int H, int W, int Cin, int Cout, int dil) {
  StreamCfg c;
  const int Mq = B * (H + dil) * (W + dil);
  if (splits < 1 || !stream_cfg(Cin, Cout, c, Mq)) return false;
  if (c.direct) return splits == 1;
  const int pps = ((Mq + splits - 1) / splits + c.bp - 1) / c.bp * c.bp;
  return (Mq + pps - 1) / pps == splits;
}

int mdcv_wgrad_stream(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int H, int W, int Cin, int Cout,
                      int dil, hipStream_t st, float* dw_oihw, int Cin_real, int Cout_real, int accumulate, int* wrote_dw) {
  if (wrote_dw) *wrote_dw = 0;
  StreamCfg c;
  if (!stream_cfg_geom(Cin, Cout, W, dil, c, (long long)B * (H + dil) * (W + dil))) return MDCV_EARG;
  if (c.bp / (W + dil) + 1 >= H + dil) return MDCV_EARG;
  WgradStreamArgs a;
  a.dy = dy; a.x = x; a.ws = ws; a.dy_ldc = dy_ldc; a.x_ldc = x_ldc;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Ktot = 9 * Cin; a.dil = dil;
  a.Wq = W + dil; a.Sq = (H + dil) * (W + dil); a.Mq = B * a.Sq;
  a.hpad = stream_hpad(W, dil);
  a.RS = stream_ring_rows(c, W, dil);
  a.pos_per_split = ((a.Mq + splits - 1) / splits + c.bp - 1) / c.bp * c.bp;
  if ((a.Mq + a.pos_per_split - 1) / a.pos_per_split != splits) return MDCV_EARG;
  a.splits = splits;
  a.tiles_ci = c.tiled ? Cin / (16 * c.nci) : 1;
  a.tiles = c.tiled ? (Cout / (16 * c.nco)) * a.tiles_ci : 1;
  a.xcd_chunk = (splits * a.tiles + 7) / 8;
  a.dw = dw_oihw; a.Cin_real = Cin_real; a.Cout_real = Cout_real; a.accumulate = accumulate;
  // the DMA-address table (TBL forms) where it fits beside the ring: 12-13 KiB for YOLOv3's layers (one window), at most 16 KiB + halo for longer runs
  a.tbl_steps = stream_tbl_steps(a.pos_per_split, c.bp);
  const int tblb = stream_tbl_bytes(a.pos_per_split, c.bp, a.hpad);
  const bool tbl = TUNE().stream_table && stream_lds(c, W, dil, tblb) <= 160 * 1024;
  const int lds = stream_lds(c, W, dil, tbl ? tblb : 0);
  const unsigned dyb = (unsigned)((long long)B * H * W * dy_ldc * 2), xb = (unsigned)((long long)B * H * W * x_ldc * 2);
  if (c.direct) {                                            // slab-free: the kernel writes the OIHW gradient itself
    if (!tbl || !dw_oihw || !wrote_dw || splits != 1) return MDCV_EARG;
    *wrote_dw = 1;
    return launch_stream1<2, 4, 4, 2, 128, 1, 1, true, 4, true, true>(a, lds, dyb, xb, st);
  }
  if (c.tiled && c.nw == 4) return c.d == 1 ? launch_stream<4, 4, 4, 1, 64, 1, 1, true, 4>(a, lds, tbl, dyb, xb, st) : launch_stream<4, 4, 4, 1, 64, 2, 1, true, 4>(a, lds, tbl, dyb, xb, st);
  if (c.tiled) {
    if (c.d == 1) return launch_stream<4, 8, 4, 1, 64, 1, 1, true>(a, lds, tbl, dyb, xb, st);
    return launch_stream<4, 8, 4, 1, 64, 2, 1, true>(a, lds, tbl, dyb, xb, st);
  }
#define STREAM_CASE(CI, CO, NCI, NCO, A, PG, BP, TGRP)                                            \
  if (Cin == CI && Cout == CO) {                                                                  \
    if (c.d == 1) return launch_stream<NCI, NCO, A, PG, BP, 1, TGRP>(a, lds, tbl, dyb, xb, st);   \
    return launch_stream<NCI, NCO, A, PG, BP, 2, TGRP>(a, lds, tbl, dyb, xb, st);                 \
  }
  STREAM_CASE(16, 16, 1, 1, 1, 8, 256, 9)
  STREAM_CASE(16, 32, 1, 2, 2, 8, 256, 9)
  STREAM_CASE(32, 32, 2, 2, 2, 4, 128, 9)
  STREAM_CASE(32, 64, 2, 4, 4, 4, 128, 3)
  STREAM_CASE(64, 64, 4, 4, 4, 2, 64, 3)
  STREAM_CASE(64, 128, 4, 8, 4, 1, 64, 1)
#undef STREAM_CASE
  return MDCV_EARG;
}

// ---- 7x7 stem (see wgrad7x7_stream_kernel)
static int stem_hpad(int W) { return (3 * (W + 3 + 1) + 31) / 32 * 32; }
static int stem_ring_rows(int W) { int need = 2 * 256 + 2 * stem_hpad(W), rs = 256; while (rs < need) rs *= 2; return rs; }
static int stem_lds(int W) {
  const int ring = (stem_ring_rows(W) + 32) * 32 + 2 * 256 * 32, stage_out = 8 * 7 * 256 * 4;
  return ring > stage_out ? ring : stage_out;
}
bool mdcv_wgrad_stem_eligible(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                              long long dy_ldc, long long x_ldc) {
  if (dtype != MDCV_BF16 || KH != 7 || KW != 7 || stride != 1 || pad != 3 || dil != 1 || Cin != 16 || Cout != 16) return false;
  if (H < 8 || W < 8 || stem_lds(W) > 160 * 1024) return false;
  const long long Mq = (long long)B * (H + 3) * (W + 3);
  if (Mq + 8192 >= (1LL << 24) || (long long)B * H * W >= (1LL << 24)) return false;
  if ((long long)B * H * W * dy_ldc * 2 >= (1LL << 31) || (long long)B * H * W * x_ldc * 2 >= (1LL << 31)) return false;
  return dy_ldc < (1 << 23) && x_ldc < (1 << 23);
}
int mdcv_wgrad_stem_splits(int B, int H, int W) {
  const int Mq = B * (H + 3) * (W + 3);
  int s = 256;
  const int max_s = (Mq + 256 * 4 - 1) / (256 * 4);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  const int pps = ((Mq + s - 1) / s + 255) / 256 * 256;
  return (Mq + pps - 1) / pps;
}
bool mdcv_wgrad_stem_splits_ok(int splits, int B, int H, int W) {
  if (splits < 1) return false;
  const int Mq = B * (H + 3) * (W + 3);
  const int pps = ((Mq + splits - 1) / splits + 255) / 256 * 256;
  return (Mq + pps - 1) / pps == splits;
}
int mdcv_wgrad_stem(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int H, int W, hipStream_t st) {
  WgradStreamArgs a;
  a.dy = dy; a.x = x; a.ws = ws; a.dy_ldc = dy_ldc; a.x_ldc = x_ldc;
  a.H = H; a.W = W; a.Cin = 16; a.Cout = 16; a.Ktot = 49 * 16; a.dil = 1;
  a.Wq = W + 3; a.Sq = (H + 3) * (W + 3); a.Mq = B * a.Sq;
  a.hpad = stem_hpad(W);
  a.RS = stem_ring_rows(W);
  a.pos_per_split = ((a.Mq + splits - 1) / splits + 255) / 256 * 256;
  if ((a.Mq + a.pos_per_split - 1) / a.pos_per_split != splits) return MDCV_EARG;
  a.splits = splits;
  a.tiles = a.tiles_ci = 1;
  a.xcd_chunk = (splits + 7) / 8;
  const int lds = stem_lds(W);
  static DynLds dyn_lds;
  if (hipError_t e = mdcv_dyn_lds(dyn_lds, reinterpret_cast<const void*>(wgrad7x7_stream_kernel), lds); e != hipSuccess) return (int)e;
  const unsigned dyb = (unsigned)((long long)B * H * W * dy_ldc * 2), xb = (unsigned)((long long)B * H * W * x_ldc * 2);
  MDCV_LAUNCH(wgrad7x7_stream_kernel, dim3((unsigned)(a.xcd_chunk * 8)), dim3(512), lds, st, a, dyb, xb);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

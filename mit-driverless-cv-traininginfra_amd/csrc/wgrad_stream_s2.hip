// Weight gradient of a 3x3 / STRIDE 2 / pad 1 convolution (even input, Hin = 2 Hout): the down-sampling layers of Darknet-53
// (reference models.py create_modules: every "stride=2" [convolutional] block), on the LDS-ring scheme of wgrad_stream.hip.
//
// Why: the generic kernel (conv_igemm.hip, conv_wgrad_dma_kernel) gathers an im2col tile per tap -- 250 MB fetched per launch for 60-90 MB of operands,
// 100-135 us alone and 230 us inside the step for 51 GFLOP (the stride-1 layers with the same FLOPs take 55 us alone).
//
// Stride 2 as a stride-1 problem: cut the input into its four PARITY PLANES, XP[py][px][img][oy][ox] = X[img][2 oy + py][2 ox + px] (space to depth, never
// materialised: a plane is an address offset (py W + px) pixels).  Output position (oy, ox) reads input row 2 oy + kh - 1: kh = 1 is plane row py = 0 at
// oy, kh = 2 is py = 1 at oy, kh = 0 is py = 1 at oy - 1; the same for columns.  So over the padded OUTPUT stream p = img (Ho+1)(Wo+1) + oy (Wo+1) + ox
// (one shared zero column per row, one zero row per image, as in the stride-1 kernel) tap (kh, kw) is plane (kh != 1, kw != 1) at displacement
// -(kh == 0) (Wo+1) - (kw == 0): every displacement is <= 0, so the ring keeps a halo on ONE side only.
//   plane (1,1): 4 taps (0,0) (0,2) (2,0) (2,2) | plane (1,0): (0,1) (2,1) | plane (0,1): (1,0) (1,2) | plane (0,0): (1,1)
//
// A ring row holds one stream position = 4 planes x 32 input channels (256 bytes = eight 32-byte units, unit u = 2 plane + channel block; unit u of row r is
// stored at unit u ^ (r & 7): eight consecutive rows of one unit cover all 64 banks, so the displaced transpose reads are conflict-free).  A block owns
// 64 output channels x 32 input channels x 9 taps and a run of stream positions.  EIGHT waves = 2 position groups (32 of a step's 64 positions each) x 4
// roles that split the 18 (tap, channel block) products: the four taps of plane (1,1) for channel block 0 / 1, the five taps of the other planes for channel
// block 0 / 1 -- 16 / 20 MFMAs per step and wave, each behind 4 dY + 5 activation fragments.  (The first version ran four waves, one per SIMD, over both
// position groups: 330 address / read / move instructions per step in front of 40 MFMAs with nothing else to issue meanwhile -- 2 450 clocks per step, 115-122 us
// per layer against 88 for the generic kernel.  With two waves per SIMD the second position group runs half a step late: it multiplies the fragments it read in
// the previous slot while its partner reads.)  Per step the block DMAs 16 KiB of activations and 8 KiB of dY; the DMA lanes take their byte offsets from two
// per-block tables (plane (0,0)'s pixel row / the dY row of a stream position; padding = one-past-the-end, zero-filled by the buffer range check).  Partial
// sums: the two position groups meet in LDS in fixed order, then leave as the split's fp32 slab ws[split][Cout][9 Cin] (wgrad_reduce_kk_kernel<9> sums the
// slabs in fixed order).
#include "common.h"
#include "wgrad_stream.h"

#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ void s2_divmod(int n, int d, float inv, int& q, int& r) {   // 0 <= n < 2^24
  q = (int)((float)n * inv);
  r = n - q * d;
  const int lt = r < 0;  q -= lt; r += lt ? d : 0;
  const int ge = r >= d; q += ge; r -= ge ? d : 0;
}
// LDS reads as inline asm: invisible to the compiler's LDS-DMA hazard pass (which would drain the DMA queue in front of every read); the kernel orders
// DMA and reads itself (vmcnt + barrier) and waits for reads with the "+v" waits below
template <int OFF> __device__ __forceinline__ s16x4_t s2_tr16(unsigned addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF> __device__ __forceinline__ int s2_ld32(unsigned addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void s2_wait_tbl(int (&nx)[2], int& ny) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(ny) :: "memory");
}
template <int N> __device__ __forceinline__ void s2_wait(bf16x8_t (&fa)[4], bf16x8_t& f) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(f) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void s2_wait(bf16x8_t& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory"); }

constexpr int S2_NW = 8, S2_BP = 64, S2_RBX = 256, S2_RBY = 128, S2_MIR = 32, S2_NT = 5;
constexpr int S2_YSTAGE = S2_BP * S2_RBY;                   // 8 KiB of dY per step
constexpr int S2_TBL_STEPS = 16;                            // steps per window of the offset tables

// D: steps in flight behind the one being multiplied (1 or 2)
template <int D>
__global__ __launch_bounds__(S2_NW * 64) void wgrad3x3_s2_stream_kernel(WgradStreamArgs a, unsigned dy_bytes, unsigned x_bytes) {
  constexpr int NW = S2_NW, BP = S2_BP, RBX = S2_RBX, RBY = S2_RBY, MIR = S2_MIR, NT = S2_NT, YSTAGE = S2_YSTAGE;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const ring = smem;
  const int logical = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.splits * a.tiles) return;
  const int split = logical / a.tiles;                       // the tiles of one split are neighbours on an XCD (they read the same pixel rows)
  const int tl = logical - split * a.tiles;
  const int tile_co = tl / a.tiles_ci, tile_ci = tl - tile_co * a.tiles_ci;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cib = wave & 1, role = (wave >> 1) & 1, pg = wave >> 2;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, x_bytes, 0x00020000);

  const int RS = a.RS;                                       // ring rows: (D + 1) BP + hpad (a multiple of 32, not a power of two: wraps are compare-and-subtract)
  const unsigned ring_bytes = (unsigned)(RS * RBX);
  unsigned char* const stages = smem + (RS + MIR) * RBX;
  const unsigned ybase = (unsigned)((RS + MIR) * RBX);
  const unsigned tbl_base = ybase + (unsigned)((D + 1) * YSTAGE);
  const int ntbl = a.tbl_steps * BP + a.hpad;                // entries per table; the dY table sits behind the activation table
  const int p_begin = split * a.pos_per_split;
  const int p_end = min(a.Mq, p_begin + a.pos_per_split);
  const int p_lo = p_begin - a.hpad;                         // stream position held by ring row 0
  const float inv_sq = 1.0f / (float)a.Sq, inv_wq = 1.0f / (float)a.Wq;
  const unsigned ldy2 = (unsigned)a.dy_ldc * 2u, lx2 = (unsigned)a.x_ldc * 2u;
  const int Win = 2 * a.W;

  // DMA roles.  Activations: 16 lanes per 256-byte row, 4 rows per instruction, chunk c = wave + 8 j at rows 4 c .. 4 c + 3 of the step; the lane at
  // physical slot sx of row r fetches logical slot sx ^ 2 (r & 7).  Chunks start at ring rows = 4 c (mod 8): r & 7 = (4 (wave & 1) + rrx) & 7.
  const int rrx = lane >> 4, sx = lane & 15;
  const int gxw = (4 * (wave & 1) + rrx) & 7;
  const int lsx = sx ^ (gxw << 1), ux = lsx >> 1;            // logical unit = 2 plane + channel block
  const unsigned lane_x = (unsigned)(((ux >> 2) * Win + ((ux >> 1) & 1)) * (int)lx2) + (unsigned)(tile_ci * 64 + (ux & 1) * 32 + (lsx & 1) * 16);
  // dY: 8 lanes per 128-byte row, 8 rows per instruction, chunk = wave; unit u of row r at u ^ ((r & 7) >> 1)
  const int rry = lane >> 3, sy = lane & 7;
  const int gyw = (rry >> 1) & 3;
  const unsigned lane_y = (unsigned)((sy ^ (gyw << 1)) * 16) + (unsigned)(tile_co * 128);

  auto wrap = [&](int r) { return r >= RS ? r - RS : r; };
  int ex[2], ey;
  unsigned tx_addr = 0, ty_addr = 0;
  int tbl_s0 = 0;
  // window of the tables that starts at step s0: entry i <-> stream position p_lo + s0 BP + i, i < tbl_steps BP + hpad.  Activation table: byte offset of the
  // pixel (2 oy, 2 ox) -- plane (0,0); the other planes are lane constants -- dY table: byte offset of the pixel (oy, ox); padding: the tensor's size.
  auto build_tbl = [&](int s0) {
    int* const tbl = reinterpret_cast<int*>(smem + tbl_base);
    const int pw = p_lo + s0 * BP;
    for (int i = tid; i < ntbl; i += NW * 64) {
      const int p = pw + i;
      bool ok = p >= 0 && p < a.Mq;
      int img, rem, y, x;
      s2_divmod(ok ? p : 0, a.Sq, inv_sq, img, rem);
      s2_divmod(rem, a.Wq, inv_wq, y, x);
      ok = ok && x < a.W && y < a.H;
      const unsigned pix = (unsigned)((img * a.H + y) * a.W + x);
      tbl[i] = ok ? (int)__umul24(4u * pix - 2u * (unsigned)x, lx2) : (int)x_bytes;
      tbl[ntbl + i] = ok ? (int)__umul24(pix, ldy2) : (int)dy_bytes;
    }
    __syncthreads();
    tbl_s0 = s0;
  };
  auto fetch = [&](int t, int (&nx)[2], int& ny) {
    const unsigned o = (unsigned)((t - tbl_s0) * (BP * 4));
    nx[0] = s2_ld32<0>(tx_addr + o); nx[1] = s2_ld32<NW * 4 * 4>(tx_addr + o);
    ny = s2_ld32<0>(ty_addr + o);
  };
  auto issue_x = [&](int e, int rho) {                       // 4 rows at ring row rho; rows below MIR also feed the mirror behind the ring's end
    const int vo = e + (int)lane_x;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + rho * RBX), 16, vo, 0, 0, 0);
    if (rho < MIR) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(ring + (RS + rho) * RBX), 16, vo, 0, 0, 0);
    }
  };
  auto issue = [&](int t, int rho_new) {                     // step t from ex / ey
    issue_x(ex[0], wrap(rho_new + wave * 4));
    issue_x(ex[1], wrap(rho_new + (wave + NW) * 4));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(stages + (t % (D + 1)) * YSTAGE + wave * 1024), 16, ey + (int)lane_y, 0, 0, 0);
  };

  // fragment roles (as in wgrad_stream.hip): lane -> rows prow + {0, 16} of the wave's 32 positions, 8 bytes of a 32-byte unit
  const int t16 = lane & 15, kq = lane >> 4;
  const int prow = kq * 4 + (t16 >> 2);
  const int sub = (t16 & 1) * 8, qlo = (t16 & 3) >> 1;
  const int gy = ((prow & 7) >> 1) & 3;
  unsigned laneY[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) laneY[i] = ybase + (unsigned)((pg * 32 + prow) * RBY + ((((2 * i + qlo) ^ (gy << 1)) << 4) + sub));
  // this wave's (tap, plane, displacement) list; role 0 has four taps: its fifth slot repeats the fourth and is dropped in the epilogue
  int tapid[NT];
  unsigned laneX[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    int plane, kh, kw;
    if (role == 0) { plane = 3; kh = k < 2 ? 0 : 2; kw = (k & 1) ? 2 : 0; if (k == 4) { kh = 2; kw = 2; } }
    else if (k < 2) { plane = 2; kh = k == 0 ? 0 : 2; kw = 1; }
    else if (k < 4) { plane = 1; kh = 1; kw = k == 2 ? 0 : 2; }
    else { plane = 0; kh = 1; kw = 1; }
    tapid[k] = (role == 0 && k == 4) ? -1 : kh * 3 + kw;
    const int disp = -(kh == 0 ? a.Wq : 0) - (kw == 0 ? 1 : 0);      // |disp| <= hpad < RS
    const int row = wrap(wrap((disp < 0 ? RS + disp : 0) + pg * 32) + prow);
    const int xcol = ((2 * (2 * plane + cib) + qlo) << 4) + sub;
    laneX[k] = (unsigned)(row * RBX + (xcol ^ ((row & 7) << 5)));    // RS is a multiple of 8: row & 7 is the physical row's
  }
  unsigned xa[NT], ya[4];
  auto step_addrs = [&](int t, int rho0) {
    const unsigned ystage = (unsigned)((t % (D + 1)) * YSTAGE), xs = (unsigned)(rho0 * RBX);
#pragma unroll
    for (int i = 0; i < 4; ++i) ya[i] = ystage + laneY[i];
#pragma unroll
    for (int k = 0; k < NT; ++k) { const unsigned s = xs + laneX[k]; xa[k] = s >= ring_bytes ? s - ring_bytes : s; }
  };
  f32x4_t acc[NT][4];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[4], fb[NT];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = bf16x8_t{};
#pragma unroll
  for (int k = 0; k < NT; ++k) fb[k] = bf16x8_t{};
  auto frag = [&](unsigned addr, auto RB) -> bf16x8_t {
    constexpr int rb = decltype(RB)::value;
    const s16x4_t lo = s2_tr16<0>(addr), hi = s2_tr16<16 * rb>(addr);
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto reads = [&]() {                                        // 4 dY + 5 activation fragments = 18 transpose reads
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag(ya[i], std::integral_constant<int, RBY>{});
#pragma unroll
    for (int k = 0; k < NT; ++k) fb[k] = frag(xa[k], std::integral_constant<int, RBX>{});
  };
  auto mfmas = [&]() {                                        // tap k starts as soon as its fragment is in (reads return in issue order)
    s2_wait<8>(fa, fb[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[0], acc[0][i], 0, 0, 0);
#define MDCV_S2_TAP(K)                                                                                      \
    s2_wait<8 - 2 * K>(fb[K]);                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[K][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[K], acc[K][i], 0, 0, 0);
    MDCV_S2_TAP(1) MDCV_S2_TAP(2) MDCV_S2_TAP(3) MDCV_S2_TAP(4)
#undef MDCV_S2_TAP
  };
  // The two waves of a SIMD (w and w + 4: the two position groups) would read together and multiply together; the second group runs half a step late
  // instead: it multiplies the fragments it read in the previous slot first, then reads -- one of the pair reads while the other multiplies.  The fragments
  // live across the barrier (the rows they were read from may be refilled afterwards: the reads have returned, see the lgkmcnt(0) in front of it).
  const bool late = pg == 1;

  const int nt = (p_end - p_begin + BP - 1) / BP;
  build_tbl(0);
  tx_addr = tbl_base + 4u * (unsigned)(a.hpad + wave * 4 + rrx);
  ty_addr = tbl_base + 4u * (unsigned)(ntbl + a.hpad + wave * 8 + rry);
  // prologue: the hpad rows below the first step, then D steps ahead
  for (int c = wave; c < a.hpad / 4; c += NW) {
    int e[2] = {s2_ld32<0>(tbl_base + 4u * (unsigned)(c * 4 + rrx)), 0}, e2 = 0;
    s2_wait_tbl(e, e2);
    issue_x(e[0], c * 4);
  }
  int issued = 0, rho_new = wrap(a.hpad);
  if (nt > 0) { fetch(0, ex, ey); s2_wait_tbl(ex, ey); }
  for (; issued < D && issued < nt; ++issued) {
    issue(issued, rho_new);
    rho_new = wrap(rho_new + BP);
    if (issued + 1 < nt) { fetch(issued + 1, ex, ey); s2_wait_tbl(ex, ey); }
  }
  int rho0 = wrap(a.hpad);
  for (int t = 0; t < nt; ++t) {
    // every fragment read of the previous step has RETURNED (lgkmcnt: the late waves carry theirs across the barrier in registers) and this step's rows have
    // landed (vmcnt: at most the D - 1 younger steps' 3 .. 5 instructions per wave stay in flight) before any wave refills a stage / ring rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (D > 1 && issued - 1 - t >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 3) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (late) mfmas();                                       // the previous slot's fragments (zeros the first time)
    const bool more = issued < nt, next = more && issued + 1 < nt;
    int nx[2], ny;
    if (next && issued + 1 - tbl_s0 == a.tbl_steps) build_tbl(issued + 1);   // (block-uniform; ex / ey already hold step `issued`)
    if (next) fetch(issued + 1, nx, ny);
    if (more) issue(issued, rho_new);
    step_addrs(t, rho0);
    if (next) {
      s2_wait_tbl(nx, ny);
      ex[0] = nx[0]; ex[1] = nx[1]; ey = ny;
    }
    reads();
    if (!late) mfmas();
    if (more) { ++issued; rho_new = wrap(rho_new + BP); }
    rho0 = wrap(rho0 + BP);
  }
  if (late) mfmas();                                         // the last slot of the late waves
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (on every path to the barrier: scripts/check_ring_barriers.py walks the built code path-insensitively)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // epilogue: one tap at a time through LDS (position group 0 stores, group 1 adds: fixed order) -> 64 rows x 32 floats of the split's slab
  // ws[split][Cout][9 Cin] (column = tap Cin + ci)
  float* so = reinterpret_cast<float*>(smem);
  constexpr int OR = 32 + 4;
  float* __restrict__ ws = a.ws + ((size_t)split * a.Cout + (size_t)tile_co * 64) * a.Ktot + tile_ci * 32;
#pragma unroll
  for (int g0 = 0; g0 < 9; ++g0) {
#pragma unroll
    for (int pgi = 0; pgi < 2; ++pgi) {
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        if (pg == pgi && tapid[k] == g0) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int idx = (i * 16 + kq * 4 + rr) * OR + cib * 16 + t16;
              so[idx] = pgi == 0 ? acc[k][i][rr] : so[idx] + acc[k][i][rr];
            }
        }
      }
      __syncthreads();
    }
    {
      const int row = tid >> 3, c4 = (tid & 7) * 4;            // 512 threads: one float4 each
      *reinterpret_cast<float4*>(ws + (size_t)row * a.Ktot + g0 * a.Cin + c4) = *reinterpret_cast<const float4*>(so + row * OR + c4);
    }
    __syncthreads();
  }
}

inline int s2_hpad(int Wo) { return (Wo + 2 + 31) / 32 * 32; }                  // >= Wq + 1 rows below a step
inline int s2_ring_rows(int Wo, int d) { return (d + 1) * S2_BP + s2_hpad(Wo); }
inline int s2_tbl_steps(int pos_per_split) { const int nt = pos_per_split / S2_BP; return nt < S2_TBL_STEPS ? nt : S2_TBL_STEPS; }
inline int s2_lds(int Wo, int pos_per_split, int d) {
  return (s2_ring_rows(Wo, d) + S2_MIR) * S2_RBX + (d + 1) * S2_YSTAGE + 2 * (s2_tbl_steps(pos_per_split) * S2_BP + s2_hpad(Wo)) * 4;
}
inline int s2_depth(int Wo) {
  int d = TUNE().stream_s2_depth < 1 ? 1 : (TUNE().stream_s2_depth > 3 ? 3 : TUNE().stream_s2_depth);
  while (d > 1 && s2_lds(Wo, S2_TBL_STEPS * S2_BP, d) > TUNE().stream_s2_lds) --d;
  return d;
}

}  // namespace

bool mdcv_wgrad_s2_eligible(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride, int pad, int dil,
                            long long dy_ldc, long long x_ldc) {
  if (!TUNE().stream_s2 || dtype != MDCV_BF16 || KH != 3 || KW != 3 || stride != 2 || pad != 1 || dil != 1) return false;
  if (Hin != 2 * Hout || Win != 2 * Wout || Hout < 4 || Wout < 4 || Wout > 255) return false;      // even inputs; ox rides in 8 bits of a table entry
  if ((Cin & 31) || (Cout & 63)) return false;
  if (s2_lds(Wout, S2_TBL_STEPS * S2_BP, 1) > TUNE().stream_s2_lds) return false;     // the block shares its CU with the main queue's workgroups
  const long long Mq = (long long)B * (Hout + 1) * (Wout + 1);
  if (Mq + 4096 >= (1LL << 24) || (long long)B * Hin * Win >= (1LL << 24)) return false;             // 24-bit multiplies / float divmod
  if ((long long)B * Hout * Wout * dy_ldc * 2 >= (1LL << 31) || (long long)B * Hin * Win * x_ldc * 2 + (long long)(Win + 2) * x_ldc * 2 >= (1LL << 31)) return false;
  return dy_ldc < (1 << 23) && x_ldc < (1 << 23);
}

int mdcv_wgrad_s2_splits(int B, int Hout, int Wout, int Cin, int Cout) {
  const int Mq = B * (Hout + 1) * (Wout + 1), tiles = (Cout / 64) * (Cin / 32);
  int s = (TUNE().stream_s2_blocks + tiles - 1) / tiles;
  const int max_s = (Mq + S2_BP * 4 - 1) / (S2_BP * 4);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  const int pps = ((Mq + s - 1) / s + S2_BP - 1) / S2_BP * S2_BP;
  return (Mq + pps - 1) / pps;
}

bool mdcv_wgrad_s2_splits_ok(int splits, int B, int Hout, int Wout) {
  if (splits < 1) return false;
  const int Mq = B * (Hout + 1) * (Wout + 1);
  const int pps = ((Mq + splits - 1) / splits + S2_BP - 1) / S2_BP * S2_BP;
  return (Mq + pps - 1) / pps == splits;
}

int mdcv_wgrad_s2(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int Hout, int Wout, int Cin, int Cout,
                  hipStream_t st) {
  WgradStreamArgs a;
  a.dy = dy; a.x = x; a.ws = ws; a.dy_ldc = dy_ldc; a.x_ldc = x_ldc;
  a.H = Hout; a.W = Wout; a.Cin = Cin; a.Cout = Cout; a.Ktot = 9 * Cin; a.dil = 1;
  a.Wq = Wout + 1; a.Sq = (Hout + 1) * (Wout + 1); a.Mq = B * a.Sq;
  a.hpad = s2_hpad(Wout);
  const int d = s2_depth(Wout);
  a.RS = s2_ring_rows(Wout, d);
  a.pos_per_split = ((a.Mq + splits - 1) / splits + S2_BP - 1) / S2_BP * S2_BP;
  if ((a.Mq + a.pos_per_split - 1) / a.pos_per_split != splits) return MDCV_EARG;
  a.splits = splits;
  a.tiles_ci = Cin / 32;
  a.tiles = (Cout / 64) * a.tiles_ci;
  a.xcd_chunk = (splits * a.tiles + 7) / 8;
  a.tbl_steps = s2_tbl_steps(a.pos_per_split);
  a.dw = nullptr; a.Cin_real = Cin; a.Cout_real = Cout; a.accumulate = 0;
  const int lds = s2_lds(Wout, a.pos_per_split, d);
  static DynLds dyn_lds1, dyn_lds2;
  const unsigned dyb = (unsigned)((long long)B * Hout * Wout * dy_ldc * 2), xb = (unsigned)((long long)B * 4 * Hout * Wout * x_ldc * 2);
#define MDCV_S2_LAUNCH(DD, CACHE)                                                                                                                   \
  {                                                                                                                                                 \
    if (hipError_t e = mdcv_dyn_lds(CACHE, reinterpret_cast<const void*>(wgrad3x3_s2_stream_kernel<DD>), lds); e != hipSuccess) return (int)e;      \
    MDCV_LAUNCH(wgrad3x3_s2_stream_kernel<DD>, dim3((unsigned)(a.xcd_chunk * 8)), dim3(S2_NW * 64), lds, st, a, dyb, xb);                           \
  }
  static DynLds dyn_lds3;
  if (d == 3) MDCV_S2_LAUNCH(3, dyn_lds3)
  else if (d == 2) MDCV_S2_LAUNCH(2, dyn_lds2)
  else MDCV_S2_LAUNCH(1, dyn_lds1)
#undef MDCV_S2_LAUNCH
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

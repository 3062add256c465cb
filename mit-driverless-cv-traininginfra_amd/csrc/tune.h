// Per-CALL tuning of the kernel dispatch (round 5: the C ABI holds no mutable process state).
//
// Every heuristic of the conv / weight-gradient / 1x1-block dispatch that was ever A/B-ed on the training step is a field of MdcvTune with its
// measured default.  A call takes its tuning from its OWN `dtype` argument: bits 0..7 are the dtype (MDCV_F32 / MDCV_BF16), the bits above carry an
// optional signed variant code of the entry point's family (include/mdcv_hip.h: MDCV_TUNED(dtype, code); 0 = defaults) which is applied to a COPY
// of the defaults for the duration of that call (TuneScope: a thread-local pointer, so concurrent callers do not see each other).  Two plans in
// one process can therefore hold different tunings, and a test that walks the variants leaks nothing.
#pragma once

struct MdcvTune {
  // ---- conv_igemm.hip
  int conv_no_ut = 0;   // tuning/A-B: 1 disables the uniform-tap address path
  int conv_tall_narrow = 256;   // 256-row tiles for Nout <= 64 from this many Ki output positions (set_variant 2000 + M_min/1024; 2000 = off): half as
      // many workgroup prologues / epilogues on the 80^2 x 256 and 208^2..416^2 x 32 tensors.  Same-box A/B: RektNet +0.65 %, YOLOv3 +0.2 %
  int conv_deep_narrow = 1;   // 128x64 tiles of the 33..64-channel layers take the 3-stage ring from this many K steps (set_variant 30 + nk_min;
      // 30 = never).  Same-box A/B: RektNet 29.93k -> 30.17k img/s, YOLOv3 +0.3 %
  int conv_deep_small = 8;   // 128x128 and 128x64 tiles take the 3-stage DMA ring from this many K steps (set_variant 60 + nk_min; 60 = never).
      // Same-box A/B of the YOLOv3 step: never 2031, from 4 steps 2045, from 8 2050, from 16 2045, from 32 2034 img/s
  int conv_fuse_narrow = 1;   // (set_variant 92 = off) 1x1 data gradients with fused BatchNorm sums take 128x64 tiles on the 3-stage ring: their store loop (loads of
      // the shortcut gradient and y, sums, partial-row flush) is serial per workgroup, and three or four narrow workgroups per CU overlap it
      // better than one or two wide ones.  Alone (round-2 A/B): 52^2 60 -> 47 us, 104^2 103 -> 78, 26^2 35 -> 32, 13^2 21.5 -> 20
  int conv_variant = -1;   // -1: heuristic ; >= 0: forced tile configuration for wide layers (tuning / A-B benchmarking)
  int conv_tall_s2 = 1;   // (set_variant 18 = off; +0.3 % on the YOLOv3 step) the tall narrow tiles for the parity-class launches too
  int conv_deep_s2 = 1;   // (set_variant 20 = off; +0.6 % on the YOLOv3 step, same-box A/B) 3-stage ring for the parity-class launches of the stride-2 data gradients
  int conv_s2_allcls = 1;   // (set_variant 16 = off) one launch for the four parity classes of a stride-2 data gradient (conv_glds_kernel ALLCLS)
  int conv_s2_split = 512;   // ALLCLS grids below this many 128 x 128 tiles run two workgroups per tile (set_variant 4000 + n; 26->52 and 13->26 at batch 32)
  int conv_s2_split_on = 1;   // (set_variant 14 = off: those layers go back to four class launches; 15 = on)
  int wgrad_slots = 512;   // target block count of the generic weight-gradient kernel (tuning hook 20000 + n)
  int wgrad_bna_stages = 2;   // ring stages of the narrow kernel with the BatchNorm apply in its operand load (34012..34014 -> 2..4), 24 KiB each.  The kernel is
      // HBM-latency-bound with little work per step: it wants blocks, not depth.  416^2 x 32 at batch 32, alone (scripts/bna_ab.py), stages 2 / 3 / 4:
      // 258 / 262 / 391 us at the generic target of 512 blocks (four stages = 96 KiB leave room for ONE block per CU: two rounds),
      // 191 / 267 / 388 us at 768 blocks (three two-stage blocks per CU in one round; the plan asks for that split count, engine.emit_first_conv_bwd)
  int wgrad_variant = 0;   // 0: default dispatch ; 4: generic address path ; 5: wide tile everywhere ; 8 / 9 / 10 / 11: kernel-family choices (use_wgrad_*)
  // ---- conv_shift.hip
  int shift_ring = 4;   // weight-ring depth of sparse grids (<= 256 tiles); 3: off (tuning hook: mdcv_conv2d_set_variant(-3 / -4))
  int shift_wmax_narrow = 104;   // rows up to 104 pixels for the 64- / 32-wide tiles (their smaller weight ring keeps two workgroups on a CU): the data gradients of
      // YOLOv3's 104x104 64->128 layers, +0.3 % on its step (set_variant(-24) off, (-23) on)
  int shift_wmax_n32 = 0;   // tuning (set_variant(-25) -> 208, (-26) -> off): 32-wide tiles on rows up to 208 pixels (128-row tiles)
  int shift_dil2 = 1;   // dilation-2 layers (stream padded with two shared zero columns / rows): 1 = where it pays (below), 2 = every eligible
      // layer (set_variant(-20)), 0 = never (set_variant(-21)), set_variant(-22) restores 1
  int shift_n64 = 2;   // 64- and 32-channel layers run one narrow tile column (set_variant(-18) off / (-17) 64 only / (-19) 64 and 32): RektNet's
      // 64->64 layers 210 -> 168 us forward, 211 -> 153 us data gradient, +0.5 % on its step; the 32->32 layers another +0.35 %
  int shift_wmax = 80;   // widest image row the shift kernel takes (set_variant(-15) -> 62, (-14) -> 80).  Up to 62 the chunk is 384 rows (3 DMAs per
      // wave); 63..80 take a fourth and still fit two workgroups on a CU: RektNet's 128->128 layers at 80x80 +3.1 % on its step,
      // the 76x76 layers of the 608^2 detector +1 % on the joint pipeline (same-box A/B)
  int shift_loop = 2;   // K-loop form of the FORWARD launches (set_variant(-30 - n)): 0 lockstep ; 1 ping-pong wave groups
      // (two groups of four waves one barrier apart: one wave of a SIMD multiplies while its partner reads fragments and issues
      // DMAs) ; 2 (default) ping-pong for grids of at most one workgroup per CU, where no second workgroup fills the
      // read phase (13^2 512->1024 forward 52.5 -> 48.9 us), and 384-row ping-pong tiles where they make ONE round of
      // 193..256 workgroups (26^2 256->512 forward 44.6 -> 42.4 us).  Denser grids: +1..2 % alone, data gradients -5..+3 %.
  int shift_2d = 1;   // images wider than the 1-D stream takes (below) run as 2-D pixel tiles of 8 x 30 outputs (set_variant(-27) off / (-28) on)
  int shift_big = 0;   // A/B: 384 forces the 384-row ping-pong tiles on every forward launch they fit
  int shift_n64_wide = 1;   // data gradients with few positions and > 64 channels on 256 x 64 tiles (see mdcv_shift_launch_dgrad)
  int shift_plan = 0;   // 0 / 5: default plan (192-row tiles where they save a round) ; 1: 256-row tiles only ; 2: 128-row only ; 6: never 192-row
  int shift_s2 = 1;   // set_variant(-29) off / (-60) on
  // ---- pw_block.hip
  int pw_bmp = 0;   // tuning hooks (mdcv_pw_set_variant): forced pixels per tile, 0 = heuristic
  int pw_wres = 1;   // weights resident in LDS where they fit
  // ---- wgrad_stream.hip
  int stream_blocks = 0;   // force the target block count of every form; 0 = defaults below (1000 + blocks/64)
  int stream_tiled = 1;   // channel-tiled instantiation for the wide layers (Cin % 64 == 0, Cout % 128 == 0); 0 = off (1800)
  int stream_light_maxpos = 600000;   // light form up to this many padded stream positions (30003: everywhere, 30005: never, 30002: default)
  int stream_table = 1;            // DMA addresses from a per-block LDS table of pixel indices (34021 on / 34020 off: the lanes step (x, y, image) forward) -- round 6
  int stream_direct = 1;           // slab-free form where one split of 64 co x 32 ci tiles fills the chip (34051 on / 34050 off) -- round 6
  int stream_s2 = 1;               // stride-2 layers on the parity-plane ring kernel (wgrad_stream_s2.hip; 34061 on / 34060 off) -- round 6
  int stream_s2_blocks = 256;      // its block target (35000 + n): 128 / 192 / 256 / 384 / 512 -> 13.04 / 12.99 / 12.99 / 13.00 / 13.03 ms (YOLOv3 step, same box)
  int stream_s2_depth = 2;         // its DMA prefetch depth (34071 .. 34073): alone 99-117 / 79-88 / 78-85 us at depth 1 / 2 / 3, the step 13.49 / 13.48 / 13.48 ms
  int stream_s2_lds = 160 * 1024;  // LDS bound of its block (34100 + KiB; the depth shrinks until it fits).  Same-box A/B of the YOLOv3 step, bound 92 / 100 / 128 / 160 KiB:
      // 13.17 / 13.22 / 13.13 / 13.10 ms against 13.13 with the generic kernel -- room for a main-queue workgroup beside the block buys nothing here
  int stream_light_depth = 1;      // its DMA prefetch depth (steps in flight behind the one being multiplied; 34000 + d)
  int stream_light_blocks = 256;   // its block target (33000 + n)
  int stream_tiled_blocks = 128;   // block target of the 8-wave tiled form (30000 + n).  A block fills its CU, and the weight gradients run BESIDE
      // the main stream: with one block on every CU the main stream's workgroups wait for whole weight-gradient blocks to
      // retire (YOLOv3 step, same-box A/B: 256 blocks 2033, 192: 2080, 128: 2103, 64: 2033 img/s)
};

extern thread_local const MdcvTune* mdcv_t_tune;          // the tuning of the call in progress on this thread (never null: the defaults outside a call)
inline const MdcvTune& TUNE() { return *mdcv_t_tune; }

enum { MDCV_TUNE_CONV = 0, MDCV_TUNE_WGRAD = 1, MDCV_TUNE_PW = 2 };

// ---- variant codes (what mdcv_conv2d_set_variant / mdcv_conv2d_wgrad_set_variant / mdcv_pw_set_variant took until round 4), applied to a copy
// conv family.  0: defaults.
//   1..12  forced tile configuration of wide layers = old codes 0..11 (+1, so that 0 can mean "defaults"): 1-6 register-staged kernels, 7-12 LDS-DMA
//          (128x128 / 128x64 / 256x128 x 2 / 3 stages); 100 + v: the same with the generic address path
//   14/15  sparse stride-2 data gradients: four class launches / one launch with two workgroups per tile     4000+n  'sparse' = below n tiles
//   16/17  stride-2 data gradient as four parity-class launches / one launch        18/19, 20/21  its tall tiles, its 3-stage ring off / on
//   30+n   3-stage ring for 33..64-channel layers from n K steps (30 never)           60+n  the same for 128x128 / 128x64 tiles (60 never)
//   92/93  128x64 tiles for fused 1x1 data gradients off / on                         2000+n  256-row tiles for Nout <= 64 from n Ki positions (2000 off)
//   -3..-26  shift-kernel knobs (below)      -27 / -28  2-D pixel tiles for wide images off / on
//   -29 / -60  stride-2 data gradients with 32 / 64 output channels through the shift kernel off / on
//   -30 / -31 / -32  shift-kernel K loop of forward launches: lockstep / ping-pong everywhere / ping-pong where measured faster (default)
//   -200 / -201  384-row ping-pong tiles: by the plan / forced on every forward launch they fit
//   -63 / -64  3x3 data gradients with few positions and > 64 channels (13^2 layers) on 256 x 64 tiles off / on
inline void mdcv_tune_shift(MdcvTune& t, int ring) {
  if (ring == 63 || ring == 64) { t.shift_n64_wide = ring - 63; return; }
  if (ring == 29 || ring == 60) { t.shift_s2 = ring == 60; return; }
  if (ring == 27 || ring == 28) { t.shift_2d = ring - 27; return; }
  if (ring >= 200 && ring < 300) { t.shift_big = ring == 201 ? 384 : 0; return; }
  if (ring >= 30 && ring <= 59) { t.shift_loop = ring - 30; return; }
  if (ring == 25 || ring == 26) { t.shift_wmax_n32 = ring == 25 ? 208 : 0; return; }
  if (ring == 23 || ring == 24) { t.shift_wmax_narrow = ring == 23 ? 104 : 0; return; }
  if (ring >= 20 && ring <= 22) { t.shift_dil2 = ring == 20 ? 2 : (ring == 21 ? 0 : 1); return; }
  if (ring >= 17 && ring <= 19) { t.shift_n64 = ring == 17 ? 1 : (ring == 18 ? 0 : 2); return; }
  if (ring >= 14 && ring <= 16) { t.shift_wmax = ring == 14 ? 80 : (ring == 15 ? 62 : 104); return; }
  if (ring >= 7) t.shift_plan = ring - 7;                    // -7..-10 -> plan 0..3
  else if (ring >= 3 && ring <= 5) t.shift_ring = ring;
}
inline void mdcv_tune_apply_conv(MdcvTune& t, int v) {
  if (v == 0) return;
  if (v <= -3 && v >= -299) { mdcv_tune_shift(t, -v); return; }
  if (v == 93 || v == 92) { t.conv_fuse_narrow = v == 93; return; }
  if (v >= 60 && v < 92) { t.conv_deep_small = v - 60; return; }
  if (v >= 30 && v < 60) { t.conv_deep_narrow = v - 30; return; }
  if (v == 20 || v == 21) { t.conv_deep_s2 = v - 20; return; }
  if (v == 16 || v == 17) { t.conv_s2_allcls = v - 16; return; }
  if (v == 14 || v == 15) { t.conv_s2_split_on = v - 14; return; }
  if (v >= 4000 && v < 6000) { t.conv_s2_split = v - 4000; return; }
  if (v == 18 || v == 19) { t.conv_tall_s2 = v - 18; return; }
  if (v >= 2000 && v < 3000) { t.conv_tall_narrow = v - 2000; return; }
  if (v >= 100) { t.conv_no_ut = 1; v -= 100; }              // 100 + v: variant v with the generic address path
  if (v >= 1 && v <= 12) t.conv_variant = v - 1;
}
// weight-gradient family.  0: defaults ; 4: generic address path ; 5: wide tile everywhere ; 8 / 9 / 10 / 11: kernel-family choices (use_wgrad_*) ;
// 1000 + 100 d + blocks/64: LDS-ring kernel with d & 8 = untiled and a forced block target ; 20000 + n: block target of the generic kernel ;
// 30000 + n: block target of the 8-wave tiled LDS-ring form (30002: default form choice, 30003: light form everywhere, 30005: never) ;
// 33000 + n: the light form's block target
inline void mdcv_tune_apply_wgrad(MdcvTune& t, int v) {
  if (v == 0) return;
  if (v >= 20000 && v < 30000) { t.wgrad_slots = v - 20000; return; }
  if (v >= 35000 && v < 36000) { t.stream_s2_blocks = v - 35000; return; }
  if (v >= 30000 && v < 40000) {
    const int b = v - 30000;
    if (b >= 4001 && b <= 4002) { t.stream_light_depth = b - 4000; return; }
    if (b >= 4012 && b <= 4014) { t.wgrad_bna_stages = b - 4010; return; }
    if (b == 4020 || b == 4021) { t.stream_table = b - 4020; return; }
    if (b == 4050 || b == 4051) { t.stream_direct = b - 4050; return; }
    if (b == 4060 || b == 4061) { t.stream_s2 = b - 4060; return; }
    if (b >= 4071 && b <= 4073) { t.stream_s2_depth = b - 4070; return; }
    if (b >= 4100 && b <= 4260) { t.stream_s2_lds = (b - 4100) * 1024; return; }
    if (b == 2) t.stream_light_maxpos = 600000;
    else if (b == 3) t.stream_light_maxpos = 1 << 30;
    else if (b == 5) t.stream_light_maxpos = 0;
    else if (b >= 3000 && b < 4000) t.stream_light_blocks = b - 3000;
    else t.stream_tiled_blocks = b > 0 ? b : 128;
    return;
  }
  if (v >= 1000) { const int d = (v - 1000) / 100; t.stream_tiled = !(d & 8); t.stream_blocks = ((v - 1000) % 100) * 64; return; }
  t.wgrad_variant = v;
}
// 1x1 block family.  0: defaults ; 64 / 32 / 16: forced pixels per tile ; 1000 / 1001: weights resident in LDS off / on
inline void mdcv_tune_apply_pw(MdcvTune& t, int v) {
  if (v == 1000 || v == 1001) { t.pw_wres = v - 1000; return; }
  if (v == 64 || v == 32 || v == 16) t.pw_bmp = v;
}

// Scope of one C-ABI call: splits the `dtype` argument into dtype (bits 0..7) and variant code (the signed bits above), applies the code to a copy
// of the defaults and makes that copy the thread's tuning until the call returns.
struct TuneScope {
  MdcvTune t;
  const MdcvTune* prev;
  int dtype;
  TuneScope(int dtype_arg, int family) : prev(mdcv_t_tune), dtype(dtype_arg & 0xff) {
    const int code = dtype_arg >> 8;                         // arithmetic shift: negative codes survive
    if (code != 0) {
      if (family == MDCV_TUNE_CONV) mdcv_tune_apply_conv(t, code);
      else if (family == MDCV_TUNE_WGRAD) mdcv_tune_apply_wgrad(t, code);
      else mdcv_tune_apply_pw(t, code);
      mdcv_t_tune = &t;
    }
  }
  explicit TuneScope(const MdcvTune& forced) : t(forced), prev(mdcv_t_tune), dtype(0) { mdcv_t_tune = &t; }
  ~TuneScope() { mdcv_t_tune = prev; }
  TuneScope(const TuneScope&) = delete;
  TuneScope& operator=(const TuneScope&) = delete;
};
// first statement of an entry point whose dispatch reads TUNE():  MDCV_TUNE_ENTRY(dtype, MDCV_TUNE_CONV);
#define MDCV_TUNE_ENTRY(dtype_var, family) const TuneScope tune_scope__((dtype_var), (family)); (dtype_var) = tune_scope__.dtype

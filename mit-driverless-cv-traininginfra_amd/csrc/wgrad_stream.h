// Internal interface between conv_igemm.hip (mdcv_conv2d_wgrad) and wgrad_stream.hip: 3x3 stride-1 "same" weight gradient
// (dilation 1 or 2) for 16..128-channel layers, activation window kept in an LDS ring.
#pragma once
#include <hip/hip_runtime.h>

struct WgradStreamArgs {
  const void* dy; const void* x; float* ws;
  int dy_ldc, x_ldc;
  int H, W, Cin, Cout, Ktot, dil;
  int Wq, Sq, Mq;                       // W+dil, (H+dil)(W+dil), B*Sq: padded position stream with `dil` shared zero columns / rows
  int hpad, RS;                         // halo rows on each side (multiple of 32) and rows of the activation ring
  int pos_per_split, splits, xcd_chunk;
  int tbl_steps;                        // steps covered by one window of the DMA-address table (TBL forms)
  float* dw; int Cin_real, Cout_real, accumulate;   // DIRECT form: the OIHW gradient itself (no slab), real channel counts, += instead of =
  int tiles, tiles_ci;                  // TILED instantiation: (Cout/128 or /64) x (Cin/64) channel tiles per split, blocks = splits * tiles
};

bool mdcv_wgrad_stream_eligible(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                                long long dy_ldc, long long x_ldc);
int mdcv_wgrad_stream_splits(int B, int H, int W, int Cin, int Cout, int dil);
bool mdcv_wgrad_stream_splits_ok(int splits, int B, int H, int W, int Cin, int Cout, int dil);
// wrote_dw: set to 1 when the launch wrote dw_oihw itself (the slab-free form; splits == 1, `ws` untouched) -- the caller then skips the slab reduce
int mdcv_wgrad_stream(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int H, int W, int Cin, int Cout,
                      int dil, hipStream_t st, float* dw_oihw, int Cin_real, int Cout_real, int accumulate, int* wrote_dw);
// 7x7 / stride 1 / pad 3 stem, 16 (padded) -> 16 channels
bool mdcv_wgrad_stem_eligible(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                              long long dy_ldc, long long x_ldc);
int mdcv_wgrad_stem_splits(int B, int H, int W);
bool mdcv_wgrad_stem_splits_ok(int splits, int B, int H, int W);
int mdcv_wgrad_stem(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int H, int W, hipStream_t st);
// 3x3 / stride 2 / pad 1 with an even input (Hin = 2 Hout, Win = 2 Wout): the four parity planes of the input as one ring (wgrad_stream_s2.hip)
bool mdcv_wgrad_s2_eligible(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride, int pad, int dil,
                            long long dy_ldc, long long x_ldc);
int mdcv_wgrad_s2_splits(int B, int Hout, int Wout, int Cin, int Cout);
bool mdcv_wgrad_s2_splits_ok(int splits, int B, int Hout, int Wout);
int mdcv_wgrad_s2(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int Hout, int Wout, int Cin, int Cout,
                  hipStream_t st);

// Internal interface between conv_igemm.hip (mdcv_conv2d_wgrad) and wgrad_shift.hip (3x3 stride-1 weight gradient, kw taps sharing a tile).
#pragma once
#include <hip/hip_runtime.h>

struct WgradShiftArgs {
  const void* dy; const void* x; float* ws;
  int dy_ldc, x_ldc;
  int H, W, Cin, Cout, Ktot;
  int Wq, Sq, Mq;                       // W+1, (H+1)(W+1), B*Sq: the padded position stream (see conv_shift.hip)
  int tiles_ci, tiles, pos_per_split, blocks_total, xcd_chunk;
};

bool mdcv_wgrad_shift_eligible(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                               long long dy_ldc, long long x_ldc);
int mdcv_wgrad_shift_splits(int B, int H, int W, int Cin, int Cout);
bool mdcv_wgrad_shift_splits_ok(int splits, int B, int H, int W);
int mdcv_wgrad_shift(const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits, int B, int H, int W, int Cin, int Cout,
                     hipStream_t st);

// Internal interface between conv_igemm.hip (dispatch of mdcv_conv2d) and conv_shift.hip (3x3 stride-1 shift-GEMM kernel, dilation 1 or 2).
#pragma once
#include <hip/hip_runtime.h>
#include "bn_fuse.h"
#include "exact_acc.h"

struct ShiftArgs {
  const void* in; const void* w; void* out; const float* bias; const void* addsrc; float* stats;
  int in_ldc, out_ldc, add_ldc;
  int B, H, W, Cin, Nout;
  int Wq, Sq, Mq;                       // W+dil, (H+dil)(W+dil), B*Sq: the padded position stream
  int tiles_n, tiles_total, xcd_chunk;
  int p_base;                           // first stream position of this launch (multiple of 256)
  int dil;                              // dilation (1 or 2): the stream carries `dil` shared zero columns per image row and zero rows per image
  BnFuseArgs fuse;                      // BatchNorm-backward sums folded into the store loop (fuse.y == NULL: off)
  EpiArgs epi;                          // inference epilogue (oscale == NULL and act == 0: off)
  int t2d, tiles_x, tiles_y, TH;        // 2-D pixel tiles (wide images): TH rows x (Wq - 2) columns per tile, Wq = tile width incl. its halo columns
  int nchunks, wrow, nca;               // Cin/32 ; 9*Cin elements per weight row ; KiB-chunks per activation chunk
  XAccArgs xacc;                        // forward statistics added to exact accumulators instead of written as rows (exact_acc.h; acc == NULL: off)
};

bool mdcv_shift_eligible(int dtype, int B, int H, int W, int Cin, int Nout, int KH, int KW, int stride, int pad, int dil, long long in_ldc);
int mdcv_shift_stats_rows(int B, int H, int W, int dil = 1, int Nout = 128);                 // partial rows of the fused data-gradient sums (one per 128 positions)
int mdcv_shift_fwd_stats_rows(int B, int H, int W, int Nout, int dil = 1);  // partial rows of the forward statistics (depends on the tile plan)
int mdcv_shift_conv(int mode, const void* in, int in_ldc, const void* w, void* out, int out_ldc, const float* bias, const void* addsrc,
                    int add_ldc, float* stats, int B, int H, int W, int Cin, int Nout, const BnFuseArgs* fuse, hipStream_t st,
                    const EpiArgs* epi = nullptr, int dil = 1, const XAccArgs* xacc = nullptr);
int mdcv_shift_s2_rows(int B, int H, int W);   // partial rows of mode 3's fused BatchNorm-backward sums (H, W = dY's)
bool mdcv_shift_s2_dgrad_eligible(int dtype, int B, int H, int W, int Cin, int Nout, long long in_ldc);   // mode 3 of mdcv_shift_conv: H, W = dY's

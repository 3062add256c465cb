// Internal interface of pw_block.hip (1x1 conv blocks with the neighbouring BatchNorm pass folded into the operand load).
#pragma once
#include <hip/hip_runtime.h>
#include "bn_fuse.h"
#include "exact_acc.h"

struct PwArgs {
  const void* in0; const void* in1; void* tout;      // forward: y, resid (or NULL), z out ; backward: dz, y, dy out
  const float* scale; const float* shift; const float* cA; const float* cB; const float* cC;
  const void* w; const float* bias; void* out; const void* addsrc; float* stats;
  int ld0, ld1, ldt, out_ldc, add_ldc;
  int M, K, N, act, ysplit;
  unsigned w_bytes;
  float slope;
  BnFuseArgs fuse;
  XAccArgs xacc;       // forward statistics added to exact accumulators instead of written as rows (exact_acc.h; acc == NULL: off)
};

int mdcv_pw_tile_rows(int K);
bool mdcv_pw_eligible(int dtype, long long M, int K, int N, int ld0, int ld1, int ldt, int out_ldc);

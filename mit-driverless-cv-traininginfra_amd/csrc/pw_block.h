// Internal interface of pw_block.hip (1x1 conv blocks with the neighbouring BatchNorm pass folded into the operand load).
#pragma once
#include <hip/hip_runtime.h>
#include "exact_acc.h"

struct PwArgs {
  const void* in0; const void* in1; void* tout;      // y, resid (or NULL), z out
  const float* scale; const float* shift;
  const void* w; const float* bias; void* out; float* stats;
  int ld0, ld1, ldt, out_ldc;
  int M, K, N, act, ysplit;
  unsigned w_bytes;
  float slope;
  XAccArgs xacc;       // forward statistics added to exact accumulators instead of written as rows (exact_acc.h; acc == NULL: off)
};

int mdcv_pw_tile_rows(int K);
bool mdcv_pw_eligible(int dtype, long long M, int K, int N, int ld0, int ld1, int ldt, int out_ldc);

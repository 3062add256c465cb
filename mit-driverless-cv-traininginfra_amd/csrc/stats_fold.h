// Forward BatchNorm statistics without a finalize launch (gfx950).
//
// A conv kernel writes one partial row [2][Nout] (sum, sum of squares) per 128 output positions; mdcv_bn_stats_finalize used to sum the rows
// and form scale / shift in a launch of its own between the conv and the BatchNorm-apply pass: 72 launches of ~6 us on the critical path of
// the YOLOv3 forward, worth 0.85 ms of a 13.8 ms step (what-if timing, scripts/ab_step.py "Xbn_stats_finalize").  Here the rows of a layer are
// cut into groups of G; the workgroup that completes a group (an agent-scope counter per group and channel tile) sums the group's rows in
// row order into one "super row", and the CONSUMER's prologue (bn_act_fwd_fold_kernel, csrc/elementwise.hip) sums the few super rows and
// does the finalize arithmetic itself, redundantly per workgroup.  Deterministic: every sum has a fixed order whoever performs it.
//
// Hand-off (guides/cdna_hip_programming.md §6 G16, counter form): rows are stored write-through (relaxed agent-scope atomic stores = sc1),
// every wave drains vmcnt, the workgroup meets, ONE lane adds the number of rows it wrote to the group's counter; the workgroup that sees the
// group complete reads the rows with agent-scope (sc1) loads.  Per-XCD L2s are not coherent: plain stores / loads here would be stale.
// The counters are zeroed by one memset at the head of every forward list (engine.Plan).
#pragma once
#include <hip/hip_runtime.h>

struct StatsFoldArgs {
  float* super;        // [ngroups][2][Nout]; NULL: off (plain rows, finalize launch)
  unsigned* cnt;       // [ngroups][tiles_n], zero before the launch
  int G, rows;         // rows per group (even), rows of the layer
};

__device__ __forceinline__ void sf_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sf_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// After this workgroup's rows [srow0, srow0 + nrows) of channel tile tile_n were written with sf_store.  Returns (uniformly) the group this
// workgroup completed, or -1.  flag: an LDS word nobody else uses between the two barriers inside.
__device__ __forceinline__ int sf_arrive(const StatsFoldArgs& f, int srow0, int nrows, int tile_n, int tiles_n, volatile int* flag, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    int done = -1;
    if (nrows > 0) {
      const int grp = srow0 / f.G;
      const unsigned expect = (unsigned)min(f.G, f.rows - grp * f.G);
      const unsigned old = __hip_atomic_fetch_add(f.cnt + (size_t)grp * tiles_n + tile_n, (unsigned)nrows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + (unsigned)nrows == expect) done = grp;
    }
    *flag = done;
  }
  __syncthreads();
  return *flag;
}

// The completing workgroup: super[grp] = sum of the group's rows for the channels [n0, n0 + ncols) of its tile, 2 * ncols <= NT / 2.  The NT
// threads are cut into row slices (thread t: column t % (2 ncols), slice t / (2 ncols)); a slice's rows are loaded eight at a time (all in
// flight together: one memory round trip per eight rows, not per row) and summed in row order, the slices meet in LDS and are added in slice
// order -- a fixed order whoever performs it.  red: NT floats of LDS nobody else uses any more.
template <int NT>
__device__ __forceinline__ void sf_fold(const StatsFoldArgs& f, const float* stats, int grp, int n0, int ncols, int Nout, int tid, float* red) {
  const int r0 = grp * f.G, r1 = min(r0 + f.G, f.rows);
  const int cols = 2 * ncols, nsl = NT / cols;             // ncols = 32 / 64 / 128: a power of two
  const int t = tid % cols, sl = tid / cols;
  const int which = t / ncols, n = n0 + (t - which * ncols);
  const int per = (r1 - r0 + nsl - 1) / nsl;               // consecutive rows per slice
  const int ra = r0 + sl * per, rb = min(ra + per, r1);
  float s = 0.f;
  if (n < Nout && sl < nsl) {
    const float* p = stats + ((size_t)which) * Nout + n;
    for (int r = ra; r < rb; r += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r + u < rb ? sf_load(p + (size_t)(r + u) * 2 * Nout) : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  __syncthreads();
  if (sl < nsl) red[sl * cols + t] = s;
  __syncthreads();
  if (tid < cols && n < Nout) {
    float tot = 0.f;
    for (int k = 0; k < nsl; ++k) tot += red[k * cols + tid];
    f.super[((size_t)grp * 2 + which) * Nout + n] = tot;
  }
}

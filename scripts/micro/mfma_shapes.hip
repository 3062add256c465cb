// every MFMA kernel of the library issues v_mfma_f32_16x16x32_bf16; is v_mfma_f32_32x32x16_bf16 an untried axis?
// The K loop of the 3x3 kernels in miniature: a wave owns a 64 x 64 output tile; per K = 32 step it reads its A and B fragments from LDS
// (8 x ds_read_b128 either way: the LDS bytes per MAC depend on the WAVE TILE, not on the MFMA shape) and multiplies them with
//   SHAPE 0: 16 x v_mfma_f32_16x16x32_bf16 (16 cycles each)      SHAPE 1: 8 x v_mfma_f32_32x32x16_bf16 (32 cycles each)
// -- the same 256 MFMA-pipe cycles, half the issue slots.  Measured: TFLOP/s of the chip with W waves per SIMD (one or two workgroups of 4 / 8 waves
// per CU), with and without the fragment reads, with and without a barrier per step (the lockstep form of conv_shift.hip).
//   hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip -o mfma_shapes && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int OFF> __device__ __forceinline__ bf16x8_t rd(unsigned a) {
  u32x4_t v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return __builtin_bit_cast(bf16x8_t, v);
}

// READS: fragments come from LDS every step (else: registers, loaded once).  BAR: one s_barrier per step.  NW: waves per workgroup.
template <int SHAPE, bool READS, bool BAR, int NW>
__global__ __launch_bounds__(NW * 64) void k(int steps, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u;      // 32 KiB of small bf16 values
  __syncthreads();
  // 64-byte rows (32 channels of one K step); 16x16x32: lane -> (row lane & 15, 16-byte slot lane >> 4), swizzle 2 * ((row >> 2) & 1)
  //                                            32x32x16: lane -> (row lane & 31, slot 2 h + (lane >> 5)),   swizzle (row >> 2) & 3   (conflict-free for b128 groups)
  unsigned aA, aB;
  if (SHAPE == 0) {
    const int r = lane & 15, q = lane >> 4;
    aA = (unsigned)((wave & 3) * 4096 + r * 64 + ((q ^ ((r >> 1) & 2)) << 4));
    aB = (unsigned)(16384 + (wave & 1) * 4096 + r * 64 + ((q ^ ((r >> 1) & 2)) << 4));
  } else {
    const int r = lane & 31, q = lane >> 5;
    aA = (unsigned)((wave & 3) * 4096 + r * 64 + ((q ^ ((r >> 2) & 3)) << 4));
    aB = (unsigned)(16384 + (wave & 1) * 4096 + r * 64 + ((q ^ ((r >> 2) & 3)) << 4));
  }
  f32x4_t acc[4][4];
  f32x16_t acc32[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  bf16x8_t fa[4], fb[4];
  fa[0] = rd<0>(aA); fa[1] = rd<1024>(aA); fa[2] = rd<2048>(aA); fa[3] = rd<3072>(aA);
  fb[0] = rd<0>(aB); fb[1] = rd<1024>(aB); fb[2] = rd<2048>(aB); fb[3] = rd<3072>(aB);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) :: "memory");
  for (int s = 0; s < steps; ++s) {
    if constexpr (BAR) __builtin_amdgcn_s_barrier();
    if constexpr (READS) {
      // 16x16x32: fa[i] = rows 16 i .. of the wave tile, all 32 k ; 32x32x16: fa[2 m + h] = rows 32 m .., k half h (slot offset 32 bytes = +2 slots)
      if (SHAPE == 0) {
        fa[0] = rd<0>(aA); fa[1] = rd<1024>(aA); fa[2] = rd<2048>(aA); fa[3] = rd<3072>(aA);
        fb[0] = rd<0>(aB); fb[1] = rd<1024>(aB); fb[2] = rd<2048>(aB); fb[3] = rd<3072>(aB);
      } else {
        fa[0] = rd<0>(aA); fa[1] = rd<0>(aA ^ 32u); fa[2] = rd<2048>(aA); fa[3] = rd<2048>(aA ^ 32u);
        fb[0] = rd<0>(aB); fb[1] = rd<0>(aB ^ 32u); fb[2] = rd<2048>(aB); fb[3] = rd<2048>(aB ^ 32u);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) :: "memory");
    }
    if (SHAPE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc32[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 * m + h], fb[2 * n + h], acc32[m][n], 0, 0, 0);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) t += acc32[i][j][0] + acc32[i][j][15];
  if (t == 12345.678f) sink[threadIdx.x] = t;
}

template <int SHAPE, bool READS, bool BAR, int NW>
int run(const char* name, int blocks, float* sink) {
  const int steps = 4096;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, READS, BAR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<SHAPE, READS, BAR, NW>), dim3(blocks), dim3(NW * 64), 32768, 0, steps, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double flop = 2.0 * 64 * 64 * 32 * (double)steps * NW * blocks;
  printf("%-58s blocks %4d x %d waves: %7.3f ms  %7.1f TFLOP/s\n", name, blocks, NW, best, flop / best / 1e9);
  return 0;
}

int main() {
  float* sink; CK(hipMalloc(&sink, 4096));
  // one wave per SIMD (4-wave workgroups, one per CU), two (8-wave workgroups), four (two 8-wave workgroups per CU)
  run<0, false, false, 4>("16x16x32 registers only", 256, sink);
  run<1, false, false, 4>("32x32x16 registers only", 256, sink);
  run<0, false, false, 8>("16x16x32 registers only", 256, sink);
  run<1, false, false, 8>("32x32x16 registers only", 256, sink);
  run<0, true, false, 4>("16x16x32 + 8 ds_read_b128 per step", 256, sink);
  run<1, true, false, 4>("32x32x16 + 8 ds_read_b128 per step", 256, sink);
  run<0, true, false, 8>("16x16x32 + 8 ds_read_b128 per step", 256, sink);
  run<1, true, false, 8>("32x32x16 + 8 ds_read_b128 per step", 256, sink);
  run<0, true, false, 8>("16x16x32 + reads, two workgroups per CU", 512, sink);
  run<1, true, false, 8>("32x32x16 + reads, two workgroups per CU", 512, sink);
  run<0, true, true, 8>("16x16x32 + reads + barrier per step (lockstep K loop)", 256, sink);
  run<1, true, true, 8>("32x32x16 + reads + barrier per step (lockstep K loop)", 256, sink);
  run<0, true, true, 8>("16x16x32 + reads + barrier, two workgroups per CU", 512, sink);
  run<1, true, true, 8>("32x32x16 + reads + barrier, two workgroups per CU", 512, sink);
  return 0;
}

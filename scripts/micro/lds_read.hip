// How fast does one CU serve LDS *reads* of the kinds the weight-gradient kernels issue, and do they overlap with MFMAs of the same wave?
// The 3x3 weight gradient (wgrad_stream.hip) spends 55.8 us on "reads + MFMAs, no fill" where its MFMAs alone are 21 us: this separates the
// issue rate of ds_read_b64_tr_b16 from that of plain ds_read_b64 / b128 and from the read -> MFMA dependency.
//   hipcc --offload-arch=gfx950 -O3 lds_read.hip -o lds_read && ./lds_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int OFF> __device__ __forceinline__ s16x4_t rd_tr(unsigned a) {
  s16x4_t v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory"); return v;
}
template <int OFF> __device__ __forceinline__ s16x4_t rd_b64(unsigned a) {
  s16x4_t v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory"); return v;
}
template <int OFF> __device__ __forceinline__ u32x4_t rd_b128(unsigned a) {
  u32x4_t v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory"); return v;
}

// KIND 0: ds_read_b64_tr_b16 with the kernel's address pattern (16 rows x 32 B per instruction, row pitch PITCH, swizzled 32-byte units)
// KIND 1: ds_read_b64 lane-linear (512 contiguous bytes)     KIND 2: ds_read_b128 lane-linear (1 KiB)
// KIND 3: ds_read_b64 with the tr pattern's addresses        KIND 4: ds_read_b128, 16 rows x 64 B (8 consecutive channels... a K-major fragment)
// MF: MFMAs per 26 reads (0 or 36); the MFMAs consume the fragments read one group earlier when DEP.
template <int KIND, int NW, int MF, bool DEP, int PITCH>
__global__ __launch_bounds__(NW * 64) void read_kernel(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u;
  __syncthreads();
  const int t16 = lane & 15, kq = lane >> 4;
  const int prow = kq * 4 + (t16 >> 2);
  const int sub = (t16 & 1) * 8, qlo = (t16 & 3) >> 1;
  constexpr int NC = PITCH / 32;
  const int g = ((prow & 7) / (8 / NC)) & (NC - 1);
  unsigned addr;
  if (KIND == 0 || KIND == 3) addr = (unsigned)(prow * PITCH + ((((2 * (wave & (NC - 1)) + qlo) ^ (g << 1)) << 4) + sub));
  else if (KIND == 1) addr = (unsigned)(lane * 8 + wave * 512);
  else if (KIND == 2) addr = (unsigned)(lane * 16 + wave * 1024);
  else addr = (unsigned)((lane >> 2) * PITCH + (lane & 3) * 16);
  f32x4_t acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  s16x4_t f[26];
  u32x4_t q[13];
#pragma unroll
  for (int k = 0; k < 26; ++k) f[k] = s16x4_t{0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 13; ++k) q[k] = u32x4_t{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    // 13 fragments = 26 b64 reads (or 13 b128 reads) per group, 36 MFMAs per group
    if constexpr (DEP && MF > 0) {
      // consume the previous group's fragments first (software pipeline of depth one group)
#pragma unroll
      for (int k = 0; k < MF; ++k) {
        bf16x8_t a_, b_;
        if constexpr (KIND == 2 || KIND == 4) { a_ = __builtin_bit_cast(bf16x8_t, q[k & 3]); b_ = __builtin_bit_cast(bf16x8_t, q[4 + k / 4]); }
        else {
          const s16x8_t av = {f[2 * (k & 3)][0], f[2 * (k & 3)][1], f[2 * (k & 3)][2], f[2 * (k & 3)][3], f[2 * (k & 3) + 1][0], f[2 * (k & 3) + 1][1], f[2 * (k & 3) + 1][2], f[2 * (k & 3) + 1][3]};
          const int j = 8 + 2 * (k / 4);
          const s16x8_t bv = {f[j][0], f[j][1], f[j][2], f[j][3], f[j + 1][0], f[j + 1][1], f[j + 1][2], f[j + 1][3]};
          a_ = __builtin_bit_cast(bf16x8_t, av); b_ = __builtin_bit_cast(bf16x8_t, bv);
        }
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, acc[k], 0, 0, 0);
      }
    }
    if constexpr (KIND == 2 || KIND == 4) {
#define R(K) q[K] = rd_b128<(K) * 2048>(addr);
      R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12)
#undef R
    } else if constexpr (KIND == 1 || KIND == 3) {
#define R(K) f[K] = rd_b64<(K) * 2048>(addr);
      R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15) R(16) R(17) R(18) R(19) R(20) R(21) R(22) R(23) R(24) R(25)
#undef R
    } else {
#define R(K) f[K] = rd_tr<(K) * 2048>(addr);
      R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15) R(16) R(17) R(18) R(19) R(20) R(21) R(22) R(23) R(24) R(25)
#undef R
    }
    if constexpr (!DEP && MF > 0) {
      // independent MFMAs between issue and wait: pure co-issue
      bf16x8_t a_ = {}, b_ = {};
#pragma unroll
      for (int k = 0; k < MF; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, acc[k], 0, 0, 0);
    }
    if constexpr (KIND == 2 || KIND == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]) :: "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]), "+v"(f[16]), "+v"(f[17]), "+v"(f[18]), "+v"(f[19]), "+v"(f[20]), "+v"(f[21]), "+v"(f[22]), "+v"(f[23]), "+v"(f[24]), "+v"(f[25]) :: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 36; ++k) s += acc[k][0];
#pragma unroll
  for (int k = 0; k < 26; ++k) s += (float)f[k][0];
#pragma unroll
  for (int k = 0; k < 13; ++k) s += (float)q[k][0];
  if (s == 1234.5f) sink[blockIdx.x] = s;
}

template <int KIND, int NW, int MF, bool DEP, int PITCH>
int run(const char* name, int blocks_per_cu, int cus, float* sink, double mhz) {
  const int iters = 4000;
  const size_t lds = 65536;
  auto k = read_kernel<KIND, NW, MF, DEP, PITCH>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = cus * blocks_per_cu;
  k<<<blocks, NW * 64, lds, 0>>>(100, sink);
  CK(hipEventRecord(e0, 0));
  k<<<blocks, NW * 64, lds, 0>>>(iters, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double clk = ms * 1e-3 * mhz * 1e6;                 // clocks of the run
  const double groups = (double)iters * NW * blocks_per_cu; // read groups per CU
  const int reads = (KIND == 2 || KIND == 4) ? 13 : 26;
  const double bytes = groups * reads * ((KIND == 2 || KIND == 4) ? 1024.0 : 512.0);
  printf("%-44s blocks/CU %d waves %d: %8.1f clk per group per wave-slot, %6.2f clk per read instr per CU, %6.1f B/clk/CU, MFMA %5.2f clk each per SIMD\n", name, blocks_per_cu,
         NW, clk / iters, clk / (groups * reads), bytes / clk, MF ? clk / ((double)iters * MF * (NW * blocks_per_cu / 4.0)) : 0.0);
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  printf("%s: %d CUs, %.0f MHz\n", p.name, cus, mhz);
  float* sink; CK(hipMalloc(&sink, 1 << 20));
#define RUN(KIND, NW, MF, DEP, PITCH, BPC) if (run<KIND, NW, MF, DEP, PITCH>(#KIND " NW" #NW " MF" #MF " " #DEP " pitch" #PITCH, BPC, cus, sink, mhz)) return 1;
  printf("kinds: 0 tr_b64 (kernel pattern), 1 b64 linear, 2 b128 linear, 3 b64 at the tr addresses, 4 b128 16 rows x 64 B\n");
  RUN(0, 4, 0, false, 128, 1) RUN(0, 4, 0, false, 128, 2) RUN(0, 8, 0, false, 128, 1) RUN(0, 4, 0, false, 256, 1)
  RUN(1, 4, 0, false, 128, 1) RUN(1, 4, 0, false, 128, 2)
  RUN(2, 4, 0, false, 128, 1) RUN(2, 4, 0, false, 128, 2)
  RUN(3, 4, 0, false, 128, 1) RUN(4, 4, 0, false, 128, 1) RUN(4, 4, 0, false, 256, 1)
  printf("-- MFMAs only-ish (reads still issued), independent of the reads\n");
  RUN(0, 4, 36, false, 128, 1) RUN(0, 4, 36, false, 128, 2) RUN(1, 4, 36, false, 128, 1) RUN(2, 4, 36, false, 128, 1)
  printf("-- MFMAs consume the previous group's fragments\n");
  RUN(0, 4, 36, true, 128, 1) RUN(0, 4, 36, true, 128, 2) RUN(0, 8, 36, true, 128, 1) RUN(1, 4, 36, true, 128, 1) RUN(2, 4, 36, true, 128, 1) RUN(2, 4, 36, true, 128, 2)
  return 0;
}

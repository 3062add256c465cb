// How fast can one CU fill its LDS with LDS-DMA (buffer_load_dwordx4 ... lds), as a function of the bytes it keeps in flight and of
// where the data lives (a 64 KiB window per block that stays in L2, a per-XCD shared 2 MiB window, or a unique HBM stream)?
// The conv / weight-gradient kernels were designed around "12.6 B/clk/CU"; this separates a per-CU path limit from latency x bytes in flight.
//   hipcc --offload-arch=gfx950 -O3 lds_fill.hip -o lds_fill && ./lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;

// NW waves per block; every wave keeps DEPTH 1-KiB DMA instructions in flight; each block moves `iters` x NW KiB.
// mode 0: block-private 64 KiB window (L2 / MALL resident after the first touch), 1: unique stream (HBM), 2: one 2 MiB window shared by all blocks
template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const unsigned char* src, size_t bytes, int iters, int mode, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (unsigned)(bytes > 0xfffffff0u ? 0xfffffff0u : bytes), 0x00020000);
  const size_t per_block = (size_t)iters * NW * 1024;
  size_t base;
  unsigned wrap;
  if (mode == 0) { base = (size_t)blockIdx.x * 65536; wrap = 65536; }
  else if (mode == 1) { base = (size_t)blockIdx.x * per_block; wrap = 0xffffffffu; }
  else { base = 0; wrap = 2u << 20; }
  unsigned off = (unsigned)(wave * 1024 + lane * 16);
  for (int i = 0; i < iters; ++i) {
    const unsigned o = mode == 1 ? off : (off & (wrap - 1));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(smem + ((i % DEPTH) * NW + wave) * 1024), 16, (int)(base + o), 0, 0, 0);
    asm volatile("" ::: "memory");
    off += NW * 1024;
    if (i >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem);
}

// the same with plain global_load_dwordx4 into registers (what a register-staged operand path would use)
template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void load_kernel(const unsigned char* src, size_t bytes, int iters, int mode, unsigned* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per_block = (size_t)iters * NW * 1024;
  size_t base;
  unsigned wrap;
  if (mode == 0) { base = (size_t)blockIdx.x * 65536; wrap = 65536; }
  else if (mode == 1) { base = (size_t)blockIdx.x * per_block; wrap = 0xffffffffu; }
  else { base = 0; wrap = 2u << 20; }
  unsigned off = (unsigned)(wave * 1024 + lane * 16);
  uint4 acc = {0, 0, 0, 0};
  uint4 v[DEPTH];
  for (int i = 0; i < iters; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned o = mode == 1 ? off : (off & (wrap - 1));
      v[d] = *reinterpret_cast<const uint4*>(src + base + o);
      off += NW * 1024;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
  }
  if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc.x;
}

template <int NW, int DEPTH, bool DMA>
int run(const unsigned char* d, size_t bytes, unsigned* sink, int blocks, int mode, double clk_ghz) {
  const int kib_per_block = 8192;                       // 8 MiB per block
  const int iters = kib_per_block / NW;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (DMA) hipLaunchKernelGGL((fill_kernel<NW, DEPTH>), dim3(blocks), dim3(NW * 64), NW * DEPTH * 1024, 0, d, bytes, iters, mode, sink);
    else hipLaunchKernelGGL((load_kernel<NW, DEPTH>), dim3(blocks), dim3(NW * 64), 0, 0, d, bytes, iters, mode, sink);
  };
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 3;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double sec = ms * 1e-3 / reps;
  const double total = (double)blocks * kib_per_block * 1024.0;
  const int cus = blocks < 256 ? blocks : 256;
  printf("%s mode %d blocks %4d waves %d depth %2d (%3d KiB in flight/block): %7.2f TB/s  %6.1f GB/s/CU  %5.1f B/clk/CU @%.1f GHz\n", DMA ? "lds-dma" : "gload  ",
         mode, blocks, NW, DEPTH, NW * DEPTH, total / sec / 1e12, total / sec / cus / 1e9, total / sec / cus / (clk_ghz * 1e9), clk_ghz);
  return 0;
}

int main() {
  const size_t bytes = (size_t)3 << 30;                 // 3 GiB (2 GiB of it addressable through one buffer descriptor offset)
  unsigned char* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
  unsigned* sink; CK(hipMalloc(&sink, 4096 * 4));
  const double clk = 2.4;
  for (int mode = 0; mode < 3; ++mode) {
    const int blocks_list[3] = {256, 512, 1};
    for (int bi = 0; bi < 3; ++bi) {
      const int blocks = blocks_list[bi];
      if (mode == 1 && (size_t)blocks * 8192 * 1024 > ((size_t)2 << 30)) continue;
      run<4, 1, true>(d, bytes, sink, blocks, mode, clk);
      run<4, 2, true>(d, bytes, sink, blocks, mode, clk);
      run<4, 4, true>(d, bytes, sink, blocks, mode, clk);
      run<4, 8, true>(d, bytes, sink, blocks, mode, clk);
      run<4, 16, true>(d, bytes, sink, blocks, mode, clk);
      run<8, 2, true>(d, bytes, sink, blocks, mode, clk);
      run<8, 4, true>(d, bytes, sink, blocks, mode, clk);
      run<8, 8, true>(d, bytes, sink, blocks, mode, clk);
      run<4, 4, false>(d, bytes, sink, blocks, mode, clk);
      run<4, 8, false>(d, bytes, sink, blocks, mode, clk);
      run<8, 8, false>(d, bytes, sink, blocks, mode, clk);
    }
  }
  return 0;
}

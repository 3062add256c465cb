// Launch-boundary cost of dependent kernels on one stream vs the same chain replayed from a hipGraph (gfx950).
//   hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap && ./launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
__global__ void wide(float* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  float* d; size_t big = 64u << 20;  CK(hipMalloc(&d, big * 4)); CK(hipMemset(d, 0, big * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 1000;
  for (int mode = 0; mode < 6; ++mode) {     // 0: 1 block ; 1: 1024 blocks small ; 2: 32 MB pass ; 3: 256 MB pass ; 4: 32 MB pass + tiny ; 5: 256 MB pass + tiny
    auto launch = [&](hipStream_t s) {
      if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s, d, 256);
      else if (mode == 1) hipLaunchKernelGGL(tiny, dim3(1024), dim3(256), 0, s, d, 1024 * 256);
      else if (mode == 2) hipLaunchKernelGGL(wide, dim3(2048), dim3(256), 0, s, d, (size_t)(8u << 20));
      else if (mode == 3) hipLaunchKernelGGL(wide, dim3(2048), dim3(256), 0, s, d, (size_t)(64u << 20));
      else if (mode == 4) { hipLaunchKernelGGL(wide, dim3(2048), dim3(256), 0, s, d, (size_t)(8u << 20)); hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, 64 * 256); }
      else { hipLaunchKernelGGL(wide, dim3(2048), dim3(256), 0, s, d, (size_t)(64u << 20)); hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, 64 * 256); }
    };
    for (int i = 0; i < 50; ++i) launch(st);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) launch(st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d stream : %.2f us per launch\n", mode, 1e3 * ms / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d graph  : %.2f us per launch\n", mode, 1e3 * ms / N);
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, st));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    printf("mode %d graph  : host hipGraphLaunch call %.1f us for %d nodes\n", mode, std::chrono::duration<double, std::micro>(t1 - t0).count(), N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}

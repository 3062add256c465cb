// What a side-queue fork costs the MAIN queue between two dependent kernels A -> B, by mechanism (gfx950):
//   0 nothing (no side work)            1 hipEventRecord on main + hipStreamWaitEvent on side (no-timing event)
//   2 the same with a device-scope (hipEventDisableSystemFence) event      3 event on A's dispatch packet (hipExtLaunchKernelGGL stop event)
//   4 a one-thread kernel on main that stores a sequence number to signal memory + hipStreamWaitValue64(GTE) on side
//   5 hipStreamWriteValue64 on main + hipStreamWaitValue64 on side
// A and B are ~20 us kernels over 256 CUs; the side kernel C (~10 us, 64 blocks) follows every fork.  Reported: main-queue time per A+B pair.
//   hipcc --offload-arch=gfx950 -O3 fork_cost.hip -o fork_cost && ./fork_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void work(float* p, int iters) {
  float v = p[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void post(unsigned long long* flag, unsigned long long v) {
  __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  float *a, *c; CK(hipMalloc(&a, 1 << 24)); CK(hipMalloc(&c, 1 << 24));
  CK(hipMemset(a, 0, 1 << 24)); CK(hipMemset(c, 0, 1 << 24));
  unsigned long long* flag = nullptr;
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  hipStream_t m, s; CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int R = 64;
  hipEvent_t ring[R], ringd[R];
  for (int i = 0; i < R; ++i) { CK(hipEventCreateWithFlags(&ring[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ringd[i], hipEventDisableTiming | hipEventDisableSystemFence)); }
  const int N = 400, IT = 600;
  unsigned long long seq = 0;
  for (int mode = 0; mode < 6; ++mode) {
    if (mode >= 4 && !can) continue;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, m));
      for (int i = 0; i < N; ++i) {
        if (mode == 3) hipExtLaunchKernelGGL(work, dim3(1024), dim3(256), 0, m, nullptr, ring[i % R], 0, a, IT);
        else hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, m, a, IT);
        if (mode == 1) { CK(hipEventRecord(ring[i % R], m)); CK(hipStreamWaitEvent(s, ring[i % R], 0)); }
        if (mode == 2) { CK(hipEventRecord(ringd[i % R], m)); CK(hipStreamWaitEvent(s, ringd[i % R], 0)); }
        if (mode == 3) CK(hipStreamWaitEvent(s, ring[i % R], 0));
        if (mode == 4) { ++seq; hipLaunchKernelGGL(post, dim3(1), dim3(1), 0, m, flag, seq); CK(hipStreamWaitValue64(s, flag, seq, hipStreamWaitValueGte, ~0ull)); }
        if (mode == 5) { ++seq; CK(hipStreamWriteValue64(m, flag, seq, 0)); CK(hipStreamWaitValue64(s, flag, seq, hipStreamWaitValueGte, ~0ull)); }
        hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, c, IT / 2);          // (mode 0: the side kernel runs free)
        hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, m, a, IT);
      }
      CK(hipEventRecord(e1, m));
      CK(hipEventSynchronize(e1));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("mode %d: %.2f us of main queue per A+B pair\n", mode, 1e3 * ms / N);
    }
  }
  return 0;
}

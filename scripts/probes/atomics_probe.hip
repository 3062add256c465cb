// How fast are fire-and-forget agent-scope 64-bit integer atomics into a small hot array (BatchNorm statistics accumulators)?
// grid of NB workgroups x 256 threads; each workgroup performs R rounds of: spin ~T ns, then every thread adds to acc[(round-dependent) tid % NACC + k * NACC]
// for k < D digits.  Reports launch time with / without the atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>   // 0: no atomics, 1: int64 agent-scope no-return, 2: fp32 agent atomics, 3: plain stores of rows
__global__ __launch_bounds__(256) void probe(long long* acc, float* rows, int nacc, int digits, int rounds, int spin) {
  const int tid = threadIdx.x;
  float x = (float)tid;
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;          // dependent chain ~ 4 cycles each
    const long long v = (long long)(x) | 1;
    if (MODE == 1) {
      for (int k = 0; k < digits; ++k)
        for (int c = tid; c < nacc; c += 256)
          __hip_atomic_fetch_add(acc + (size_t)k * nacc + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 2) {
      for (int k = 0; k < digits; ++k)
        for (int c = tid; c < nacc; c += 256)
          __hip_atomic_fetch_add(reinterpret_cast<float*>(acc) + (size_t)k * nacc + c, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 3) {
      for (int c = tid; c < nacc; c += 256) rows[((size_t)blockIdx.x * rounds + r) * nacc + c] = x;
    }
  }
  if (x == 12345.678f) acc[0] = 1;
}

int main() {
  long long* acc; float* rows;
  CK(hipMalloc(&acc, 1 << 20)); CK(hipMemset(acc, 0, 1 << 20));
  CK(hipMalloc(&rows, 256u << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nbs[] = {256, 704, 2048};
  const int naccs[] = {128, 512, 2048};           // 2 sums x 64 / 256 / 1024 channels
  for (int nb : nbs) for (int nacc : naccs) for (int digits : {1, 2, 3}) for (int spin : {500, 4000}) {
    const int rounds = 4;
    float t[4];
    for (int mode = 0; mode < 4; ++mode) {
      if ((size_t)nb * rounds * nacc * 4 > (256u << 20)) { t[mode] = -1; continue; }
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) {
          if (mode == 0) probe<0><<<nb, 256>>>(acc, rows, nacc, digits, rounds, spin);
          if (mode == 1) probe<1><<<nb, 256>>>(acc, rows, nacc, digits, rounds, spin);
          if (mode == 2) probe<2><<<nb, 256>>>(acc, rows, nacc, digits, rounds, spin);
          if (mode == 3) probe<3><<<nb, 256>>>(acc, rows, nacc, digits, rounds, spin);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[mode], e0, e1));
      }
    }
    printf("nb %4d nacc %4d digits %d spin %4d : none %.1f us  i64-atomics %.1f us  f32-atomics %.1f us  row-stores %.1f us   (atomics per launch %d)\n", nb, nacc, digits, spin,
           t[0] * 100, t[1] * 100, t[2] * 100, t[3] * 100, nb * rounds * nacc * digits);
  }
  return 0;
}

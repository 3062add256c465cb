// Fused flat-buffer optimizers for gfx950 (HBM-bound: one pass over params / grads / state).
// Replaces torch.optim.Adam / SGD.step() as called from CVC-YOLOv3/train.py:180-187,72 and RektNet/train_eval.py:263,72
// (222 / 54 small tensors per step in the reference -> one launch over the flat parameter buffer).
// Update rules are torch's (Adam: bias-corrected, eps added after sqrt(v_hat); SGD: momentum buffer, dampening 0).
#include <mutex>
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                                   float bc1, float bc2_sqrt, float grad_scale) {
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = G[e] * grad_scale + weight_decay * P[e];
      M[e] = beta1 * M[e] + (1.f - beta1) * gr;
      V[e] = beta2 * V[e] + (1.f - beta2) * gr * gr;
      P[e] -= (lr / bc1) * M[e] / (sqrtf(V[e]) / bc2_sqrt + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gr = g[i] * grad_scale + weight_decay * p[i];
    m[i] = beta1 * m[i] + (1.f - beta1) * gr;
    v[i] = beta2 * v[i] + (1.f - beta2) * gr * gr;
    p[i] -= (lr / bc1) * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n, float lr,
                                                  float momentum, float weight_decay, int first_step, float grad_scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gr = g[i] * grad_scale + weight_decay * p[i];
    if (momentum != 0.f) {
      const float b = first_step ? gr : momentum * buf[i] + gr;
      buf[i] = b;
      gr = b;
    }
    p[i] -= lr * gr;
  }
}

}  // namespace

extern "C" {

int mdcv_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return MDCV_EARG;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return MDCV_EARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  long long g = (n / 4 + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
  MDCV_LAUNCH(adam_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2), grad_scale);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

int mdcv_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, int step, float lr, float momentum, float weight_decay,
                  float grad_scale, void* stream) {
  if (!params || !grads || (momentum != 0.f && !momentum_buf) || step < 1) return MDCV_EARG;
  long long g = (n + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
  MDCV_LAUNCH(sgd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf, n, lr, momentum, weight_decay,
                     step == 1 ? 1 : 0, grad_scale);
  MDCV_CHECK_LAUNCH();
  return MDCV_OK;
}

// ---- device / runtime helpers (plumbing for the Python host)
int mdcv_device_info(int* cu_count, int* wave_size, long long* hbm_bytes, char* arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return (int)e;
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev); if (e != hipSuccess) return (int)e;
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (hbm_bytes) *hbm_bytes = (long long)p.totalGlobalMem;
  if (arch && arch_len > 0) { int i = 0; for (; i < arch_len - 1 && p.gcnArchName[i]; ++i) arch[i] = p.gcnArchName[i]; arch[i] = 0; }
  return MDCV_OK;
}

// HIP-event timing on a caller stream (bench.py measures kernels on the stream they are launched on)
int mdcv_event_create(void** ev) { hipEvent_t e; hipError_t r = hipEventCreate(&e); *ev = (void*)e; return (int)r; }
int mdcv_event_record(void* ev, void* stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
int mdcv_event_sync(void* ev) { return (int)hipEventSynchronize((hipEvent_t)ev); }
int mdcv_event_elapsed_ms(void* start, void* stop, float* ms) { return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop); }
int mdcv_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }

// Cross-stream ordering inside one device: everything enqueued on `from` so far happens before what is enqueued on `to` afterwards.
// One event record + one stream wait, on events of a per-device ring created WITHOUT timing and (device_scope != 0) with
// hipEventReleaseToDevice: the default event releases to SYSTEM scope, which a consumer kernel on the same GPU does not need.
// Re-recording a ring event is legal once its wait is enqueued (the wait captured the record it saw) -- so record AND wait happen under
// the ring's lock: with more than RING forks in flight from other threads a slot could otherwise be re-recorded between this thread's
// record and its wait.  The ring is that of the device OWNING `from` (hipStreamGetDevice; the legacy NULL stream counts for the current
// device), not of whatever device is current.  Scope: the device-scope release orders the producer's writes for consumers ON THE SAME
// GPU only; a host or peer-GPU reader needs device_scope == 0.
int mdcv_stream_fork(void* from, void* to, int device_scope) {
  constexpr int RING = 64, MAXDEV = 16;
  static hipEvent_t ring[MAXDEV][2][RING];
  static unsigned head[MAXDEV][2];
  static std::mutex mu;
  int dev = 0, cur = 0;
  hipError_t e = hipGetDevice(&cur); if (e != hipSuccess) return (int)e;
  dev = cur;
  if (from) {
    hipDevice_t sd;
    if (hipStreamGetDevice((hipStream_t)from, &sd) == hipSuccess) dev = (int)sd;
    else (void)hipGetLastError();
  }
  if (dev < 0 || dev >= MAXDEV) return MDCV_EARG;
  const int sc = device_scope ? 1 : 0;
  std::lock_guard<std::mutex> g(mu);
  const unsigned i = head[dev][sc]++ % RING;
  if (!ring[dev][sc][i]) {
    if (dev != cur) { e = hipSetDevice(dev); if (e != hipSuccess) return (int)e; }     // events belong to the device current at creation
    e = hipEventCreateWithFlags(&ring[dev][sc][i], hipEventDisableTiming | (sc ? hipEventReleaseToDevice : 0u));
    if (dev != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) { ring[dev][sc][i] = nullptr; return (int)e; }
  }
  const hipEvent_t ev = ring[dev][sc][i];
  e = hipEventRecord(ev, (hipStream_t)from); if (e != hipSuccess) return (int)e;
  return (int)hipStreamWaitEvent((hipStream_t)to, ev, 0);
}

// The same ordering without a marker packet in the producer's queue: the NEXT kernel this library launches (it must be on `from`, and the
// caller must launch exactly one before mdcv_stream_fork_wait) carries a ring event as the stop event of its own dispatch packet; `to`
// then waits for that event.  An event record between two dependent kernels of one queue costs ~7 us of that queue (rocprofv3 trace of
// the YOLOv3 backward: 7.2 - 7.8 us between a kernel and its successor wherever a fork sat between them, 0.0 - 0.6 us elsewhere).
int mdcv_stream_fork_arm(void* from, int device_scope, void** ev_out) {
  constexpr int RING = 64, MAXDEV = 16;
  static hipEvent_t ring[MAXDEV][2][RING];
  static unsigned head[MAXDEV][2];
  static std::mutex mu;
  if (!ev_out) return MDCV_EARG;
  int dev = 0, cur = 0;
  hipError_t e = hipGetDevice(&cur); if (e != hipSuccess) return (int)e;
  dev = cur;
  if (from) {
    hipDevice_t sd;
    if (hipStreamGetDevice((hipStream_t)from, &sd) == hipSuccess) dev = (int)sd;
    else (void)hipGetLastError();
  }
  if (dev < 0 || dev >= MAXDEV) return MDCV_EARG;
  const int sc = device_scope ? 1 : 0;
  std::lock_guard<std::mutex> g(mu);
  const unsigned i = head[dev][sc]++ % RING;
  if (!ring[dev][sc][i]) {
    if (dev != cur) { e = hipSetDevice(dev); if (e != hipSuccess) return (int)e; }
    e = hipEventCreateWithFlags(&ring[dev][sc][i], hipEventDisableTiming | (sc ? hipEventReleaseToDevice : 0u));
    if (dev != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) { ring[dev][sc][i] = nullptr; return (int)e; }
  }
  mdcv_t_arm = ring[dev][sc][i];
  *ev_out = (void*)ring[dev][sc][i];
  return MDCV_OK;
}
int mdcv_stream_fork_wait(void* to, void* ev) {
  if (!ev) return MDCV_EARG;
  if (mdcv_t_arm) { mdcv_t_arm = nullptr; return MDCV_EARG; }      // nothing was launched since the arm: the event was never bound
  return (int)hipStreamWaitEvent((hipStream_t)to, (hipEvent_t)ev, 0);
}

// hipGraph capture of a launch sequence issued through this library on `stream`
int mdcv_graph_begin(void* stream) { return (int)hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal); }
int mdcv_graph_end(void* stream, void** graph_exec) {
  hipGraph_t g; hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g); if (e != hipSuccess) return (int)e;
  hipGraphExec_t ge; e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  const hipError_t ed = hipGraphDestroy(g);               // the template graph is not needed once instantiated (or on failure)
  if (e != hipSuccess) return (int)e;
  if (ed != hipSuccess) return (int)ed;
  *graph_exec = (void*)ge;
  return MDCV_OK;
}
int mdcv_graph_launch(void* graph_exec, void* stream) { return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream); }
int mdcv_graph_destroy(void* graph_exec) { return (int)hipGraphExecDestroy((hipGraphExec_t)graph_exec); }

}  // extern "C"

// ---- on-box peak probe (SURVEY 8d "confirm on the box"): bench.py prices the kernels against the guide's spec peaks (2.5 PFLOP/s dense bf16, 8 TB/s)
// and prints what THIS box sustains beside them.  MFMA: every wave runs `iters` rounds of eight independent v_mfma_f32_16x16x32_bf16 on register
// operands (no memory in the loop); 2 * 16*16*32 FLOP each.  HBM: the caller times a large device-to-device copy.
namespace {
__global__ __launch_bounds__(256) void probe_mfma_kernel(float* __restrict__ sink, int iters) {
  typedef __attribute__((ext_vector_type(8))) __bf16 b8;
  const float seed = (float)(threadIdx.x & 15) * 0.001f;
  b8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + 0.01f * e); b[e] = (__bf16)(0.5f - seed); }
  f32x4_t acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // (inline asm on fixed registers: with the builtin, hipcc rotated the eight accumulators through ~50 v_accvgpr moves per round and the
  //  probe read half the rate)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 12345.678f) sink[0] = s;                          // (keeps the loop alive; never true)
}
}  // namespace

extern "C" {
/* enqueue the MFMA probe: blocks x 4 waves x iters x 8 MFMAs of 16384 FLOP; returns the FLOP count through *flops (host side) */
int mdcv_probe_mfma(int blocks, int iters, float* sink, double* flops, void* stream) {
  if (blocks < 1 || iters < 1 || !sink) return MDCV_EARG;
  MDCV_LAUNCH(probe_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
  MDCV_CHECK_LAUNCH();
  if (flops) *flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16.0 * 16.0 * 32.0;
  return MDCV_OK;
}
}  // extern "C"

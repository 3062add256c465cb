// Runtime plumbing of libmdcv_hip.so that is not a kernel:
//   * the in-library kernel profiler behind MDCV_LAUNCH (common.h): a start / stop HIP event pair bound to the dispatch of
//     EVERY kernel this library launches (hipExtLaunchKernel: the kernel's own begin / end timestamps on the stream it runs on),
//     with the kernel's symbol as rocprofv3 prints it (bench.py's `roofline` /
//     `roofline_kernels` are built from these; the rocprofv3 kernel trace of the same command under profiles/ must agree);
//   * the gradient exchange over RCCL for hosts that do not go through torch.distributed (SURVEY.md §8b `comm_init`,
//     `allreduce_sum`): the one collective of the data-parallel path (CVC-YOLOv3/train.py:193-195 nn.DataParallel's
//     gradient reduction, as one all-reduce(SUM) of the flat fp32 gradient buffer).
#include "common.h"

#include <cxxabi.h>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

int mdcv_g_prof = 0;
static const MdcvTune kDefaultTune;                       // every knob at its measured default (tune.h)
thread_local const MdcvTune* mdcv_t_tune = &kDefaultTune;
thread_local hipEvent_t mdcv_t_arm = nullptr;   // (per thread: arm and launch happen on one thread)

namespace {
struct Rec { const void* fn; hipEvent_t e0, e1; };
std::vector<Rec> g_recs;
std::mutex g_mu;
std::map<const void*, std::string> g_names;

const std::string& kernel_name(const void* fn) {
  auto it = g_names.find(fn);
  if (it != g_names.end()) return it->second;
  std::string s;
  const char* raw = fn ? hipKernelNameRefByPtr(fn, nullptr) : nullptr;
  (void)hipGetLastError();
  if (raw && *raw) {
    int status = 1;
    char* dem = abi::__cxa_demangle(raw, nullptr, nullptr, &status);
    s = (status == 0 && dem) ? dem : raw;
    free(dem);
  } else {
    char buf[48];
    snprintf(buf, sizeof buf, "kernel@%p", fn);
    s = buf;
  }
  return g_names.emplace(fn, s).first->second;
}
}  // namespace

void mdcv_prof_new(const void* fn, hipEvent_t* e0, hipEvent_t* e1) {
  Rec r{fn, nullptr, nullptr};
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { *e0 = *e1 = nullptr; return; }
  *e0 = r.e0; *e1 = r.e1;
  std::lock_guard<std::mutex> g(g_mu);
  g_recs.push_back(r);
}

// ---------------------------------------------------------------------------------------------- RCCL, bound at run time
// librccl is dlopen'ed by soname so that a process that already holds one (torch links its own copy) shares that instance.
namespace {
struct RcclId { char internal[128]; };                     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef int (*fn_get_id)(RcclId*);
typedef int (*fn_init_rank)(void**, int, RcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
struct Rccl { void* h = nullptr; fn_get_id get_id; fn_init_rank init_rank; fn_allreduce allreduce; fn_destroy destroy; bool ok = false; };
Rccl g_rccl;
std::once_flag g_rccl_once;

bool rccl_load() {
  std::call_once(g_rccl_once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) { g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (g_rccl.h) break; }
    if (!g_rccl.h) for (const char* n : names) { g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.h) break; }
    if (!g_rccl.h) return;
    g_rccl.get_id = (fn_get_id)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.ok = g_rccl.get_id && g_rccl.init_rank && g_rccl.allreduce && g_rccl.destroy;
  });
  return g_rccl.ok;
}
constexpr int kNcclFloat = 7, kNcclSum = 0;                  // rccl.h: ncclFloat32 = 7, ncclSum = 0
constexpr int MDCV_ENOLIB = -2;
}  // namespace

extern "C" {

int mdcv_profile_begin(void) {
  std::lock_guard<std::mutex> g(g_mu);
  for (auto& r : g_recs) { if (r.e0) (void)hipEventDestroy(r.e0); if (r.e1) (void)hipEventDestroy(r.e1); }
  g_recs.clear();
  mdcv_g_prof = 1;
  return MDCV_OK;
}

int mdcv_profile_count(void) {
  std::lock_guard<std::mutex> g(g_mu);
  return (int)g_recs.size();
}

int mdcv_profile_stop(void) {
  mdcv_g_prof = 0;
  std::lock_guard<std::mutex> g(g_mu);
  for (auto& r : g_recs)
    if (r.e1) { hipError_t e = hipEventSynchronize(r.e1); if (e != hipSuccess) return -(int)e - 1000; }
  return (int)g_recs.size();
}

int mdcv_profile_read(int i, float* ms, char* name, int name_len) {
  std::lock_guard<std::mutex> g(g_mu);
  if (i < 0 || i >= (int)g_recs.size() || !ms) return MDCV_EARG;
  const Rec& r = g_recs[i];
  *ms = 0.f;
  if (r.e0 && r.e1) { hipError_t e = hipEventElapsedTime(ms, r.e0, r.e1); if (e != hipSuccess) return (int)e; }
  if (name && name_len > 0) {
    const std::string& s = kernel_name(r.fn);
    const int n = (int)s.size() < name_len - 1 ? (int)s.size() : name_len - 1;
    memcpy(name, s.data(), n);
    name[n] = 0;
  }
  return MDCV_OK;
}

int mdcv_comm_unique_id(void* id128) {
  if (!id128) return MDCV_EARG;
  if (!rccl_load()) return MDCV_ENOLIB;
  return g_rccl.get_id((RcclId*)id128);
}

int mdcv_comm_init(void** comm, int nranks, const void* id128, int rank) {
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return MDCV_EARG;
  if (!rccl_load()) return MDCV_ENOLIB;
  RcclId id;
  memcpy(&id, id128, sizeof id);
  return g_rccl.init_rank(comm, nranks, id, rank);
}

int mdcv_comm_allreduce_sum(void* comm, float* buf, long long n, void* stream) {
  if (!comm || !buf || n < 0) return MDCV_EARG;
  if (!rccl_load()) return MDCV_ENOLIB;
  if (n == 0) return MDCV_OK;
  return g_rccl.allreduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, comm, (hipStream_t)stream);
}

int mdcv_comm_destroy(void* comm) {
  if (!comm) return MDCV_EARG;
  if (!rccl_load()) return MDCV_ENOLIB;
  return g_rccl.destroy(comm);
}

}  // extern "C"

import ctypes, subprocess, os, numpy as np, torch
src = r'''
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void probe(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x;
  int addr_elems;
  if (mode == 0) addr_elems = l * 4;
  else { int g = l >> 4, t = l & 15; addr_elems = g * 1024 + (t >> 2) * 128 + (t & 3) * 4; }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + addr_elems));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
extern "C" int run(unsigned short* out, int mode) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, mode); return (int)hipDeviceSynchronize(); }
'''
os.makedirs("/tmp/tr", exist_ok=True)
open("/tmp/tr/tr.hip", "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "/tmp/tr/tr.hip", "-o", "/tmp/tr/libtr.so"])
lib = ctypes.CDLL("/tmp/tr/libtr.so")
for mode in (0, 1):
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    rc = lib.run(ctypes.c_void_p(out.data_ptr()), mode)
    v = out.cpu().numpy().astype(np.uint16).reshape(64, 4)
    print("mode", mode, "rc", rc)
    for l in list(range(0, 20)) + [32, 33, 48, 63]:
        print("  lane", l, v[l].tolist())

import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from mdcv import _lib
L = _lib.lib()
f = L.cdll.mdcv_debug_shift_occ; f.argtypes = [ctypes.c_int]
for lds in (32768, 65536, 66560, 74752, 77824, 79872, 81920):
    f(lds)

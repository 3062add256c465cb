"""ctypes binding of libmdcv_hip.so.

Prototypes are parsed from ``include/mdcv_hip.h`` so the header is the single source of truth for the C ABI.
There is NO CPU fallback: if the shared library is missing, or an entry point is asked to run without a GPU, this
module fails loudly (the oracle under ``oracle/`` is test infrastructure and is never imported from here).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "mdcv_hip.h")
LIB_PATH = os.environ.get("MDCV_LIB") or os.path.join(_HERE, "libmdcv_hip.so")      # MDCV_LIB: another BUILD of the same library (timing ablations, scripts/wgrad_ab.py)

F32, BF16 = 0, 1


def tuned(dtype, code):
    """dtype argument carrying a per-call variant code (include/mdcv_hip.h MDCV_TUNED; csrc/tune.h lists the codes; 0 = defaults)"""
    return (dtype & 0xff) | (int(code) * 256)
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
}


def _ctype(decl):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_void_p            # every pointer crosses as an address (torch .data_ptr() / int / None)
    decl = re.sub(r"\b(const|unsigned)\b", "", decl).strip()
    base = " ".join(decl.split()[:-1]) if len(decl.split()) > 1 else decl
    return _CT[base]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every ``mdcv_*`` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long long)\s+(mdcv_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argl = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        names = [re.split(r"[\s\*]+", a)[-1] for a in argl]
        protos[name] = (_CT[ret], [_ctype(a) for a in argl], names)
    return protos


class MdcvError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise MdcvError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {os.path.join(_HERE, 'csrc')}`). There is no CPU fallback for the product path.")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It must be in the process
        # BEFORE our library is dlopen'ed so that both bind to the same runtime instance; two runtimes in one process
        # do not share devices/streams (kernels launched through the second one fail with hipErrorNoDevice).
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, argt, _) in self.protos.items():
            fn = getattr(self.cdll, name)       # AttributeError here == header/library mismatch
            fn.restype = ret
            fn.argtypes = argt
            setattr(self, name[len("mdcv_"):], fn)

    def check(self, rc, what=""):
        if rc != 0:
            raise MdcvError(f"libmdcv_hip: {what} failed with code {rc}" + (" (bad argument)" if rc == -1 else " (hipError_t)"))


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def require_gpu(t=None):
    import torch
    if not torch.cuda.is_available():
        raise MdcvError("MDCV HIP path needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    if t is not None and not t.is_cuda:
        raise MdcvError("MDCV HIP path got a CPU tensor: move the model and inputs to the GPU (`.to('cuda')`)")

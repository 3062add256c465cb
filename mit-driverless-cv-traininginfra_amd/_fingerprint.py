"""Fingerprint of everything that decides which kernels a training step launches and what they do: the HIP sources and the
Python files that lower a network into a launch plan.  Profile artefacts under profiles/ (PMC HBM traffic, MFMA-busy counters)
record the fingerprint they were measured with; bench.py refuses to quote one whose fingerprint is not the running tree's
(there is no .git on the GPU box, so a commit hash cannot serve)."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def fingerprint_files():
    pats = ["csrc/*.hip", "csrc/*.h", "csrc/*.inc", "csrc/Makefile", "engine.py", "_lib.py", "yolo/models.py", "rektnet/keypoint_net.py",
            "rektnet/resnet.py", "optim.py", "pipeline.py", "parallel.py"]
    out = []
    for p in pats:
        out += sorted(glob.glob(os.path.join(_HERE, p)))
    return out


def kernel_fingerprint():
    h = hashlib.sha256()
    for f in fingerprint_files():
        h.update(os.path.relpath(f, _HERE).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_fingerprint())

"""Builds libdcn_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree."""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(HERE, "libdcn_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "dcn_hip.h"),
                                                                           os.path.abspath(__file__)]:
        h.update(f.encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_library(force=False, verbose=False, extra_flags=()):
    """hipcc --offload-arch=gfx950 every csrc/*.hip (in parallel) and link libdcn_hip.so next to this file."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "stamp")
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC] + list(extra_flags)
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + src)
        if verbose and out:
            sys.stdout.write(out.decode())
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs)
    open(stamp, "w").write(dig)
    return OUT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))

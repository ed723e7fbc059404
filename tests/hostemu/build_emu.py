#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Builds tests/hostemu/_build/libdcn_emu.so: the product's HIP sources
(pytorch-dense-correspondence_amd/csrc/*.hip, unmodified) compiled for the HOST against the stand-in
<hip/hip_runtime.h> of this directory, so CPU tests can drive the kernels' logic through the same C ABI.
Never used by the product, bench.py or the -m gpu tests."""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pytorch-dense-correspondence_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libdcn_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "hipemu_runtime.cpp")]


def _digest():
    h = hashlib.sha256()
    for f in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
            os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "dcn_hip.h"), __file__]:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    stamp = os.path.join(OUT_DIR, "stamp")
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        cmd = [CLANG, "-x", "c++", "-std=c++17", "-O2", "-g0", "-fPIC", "-DDCN_HOSTEMU_BUILD=1", "-mf16c",
               "-Wno-unused-value", "-Wno-ignored-attributes", "-Wno-unknown-attributes",
               "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("host-emulation compile failed: " + src)
    cmd = [CLANG, "-shared", "-o", OUT] + objs + ["-lpthread"]
    subprocess.check_call(cmd)
    open(stamp, "w").write(dig)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

#!/usr/bin/env python3
"""Per-kernel resource table of the shipped libdcn_hip.so: the gfx950 code objects are cut out of the library's
``.hip_fatbin`` section (clang offload bundles) and their ``amdhsa.kernels`` notes read with llvm-readelf --
registers, LDS, scratch (``.private_segment_fixed_size``), spill counts.  Used by tests/test_kernel_resources.py (no MFMA
kernel may use scratch) and by hand:

    python tools/kernel_resources.py [--scratch-only] [path/to/libdcn_hip.so]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = os.environ.get("DCN_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "pytorch-dense-correspondence_amd", "dcn_hip", "libdcn_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size", ".uses_dynamic_stack")


def code_objects(lib, workdir):
    fat = os.path.join(workdir, "fatbin.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    d = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            break
        off = i + len(MAGIC)
        (num,) = struct.unpack_from("<Q", d, off)
        off += 8
        for _ in range(num):
            o, s, ts = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + ts].decode()
            off += ts
            if "gfx950" in triple and s > 0:
                p = os.path.join(workdir, "co_%d.elf" % len(out))
                open(p, "wb").write(d[i + o:i + o + s])
                out.append(p)
        pos = i + len(MAGIC)
    return out


def demangle(names):
    try:
        p = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True)
        return p.stdout.split("\n")[:len(names)]
    except Exception:
        return list(names)


def kernels(lib=DEFAULT):
    """[{name, demangled, vgpr_count, ..., private_segment_fixed_size, group_segment_fixed_size}]"""
    rows = []
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(lib, wd):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True,
                                   check=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r"\s+(-\s+)?(\.[a-z_]+):\s+(.*)$", line)
                if not m:
                    continue
                key, val = m.group(2), m.group(3).strip().strip("'")
                if key == ".agpr_count" or (m.group(1) and key in FIELDS) or (cur is None and key in FIELDS):
                    pass
                if key not in FIELDS:
                    continue
                if cur is None or key in cur:
                    cur = {}
                    rows.append(cur)
                cur[key] = val
    rows = [r for r in rows if ".name" in r and ".private_segment_fixed_size" in r]
    for r, dm in zip(rows, demangle([r[".name"] for r in rows])):
        r["demangled"] = dm
    out = []
    for r in rows:
        e = {"name": r[".name"], "demangled": r["demangled"]}
        for k in FIELDS[1:]:
            v = r.get(k, "0")
            e[k[1:]] = int(v) if re.fullmatch(r"-?\d+", v) else v
        out.append(e)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = kernels(args[0] if args else DEFAULT)
    only = "--scratch-only" in sys.argv
    print("%6s %5s %5s %7s %7s %8s  %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds  kernel"))
    for r in sorted(rows, key=lambda r: r["demangled"]):
        if only and not r["private_segment_fixed_size"]:
            continue
        print("%6d %5d %5d %7d %7d %8d  %6d  %s" % (r["vgpr_count"], r["agpr_count"], r["sgpr_count"], r["vgpr_spill_count"],
                                                   r["sgpr_spill_count"], r["private_segment_fixed_size"],
                                                   r["group_segment_fixed_size"], r["demangled"][:150]))
    print(len(rows), "kernels;", sum(1 for r in rows if r["private_segment_fixed_size"]), "with scratch")


if __name__ == "__main__":
    main()

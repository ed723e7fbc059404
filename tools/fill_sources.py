#!/usr/bin/env python3
"""Where do the small torch fill / copy kernels of a training step come from?  (rocprofv3: ~32 FillFunctor launches + ~29 copyBuffer
per config-2 step.)  torch.profiler with Python stacks over three steps of the bench's Job; prints the call sites by launch count."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

class A: pass
a = A(); a.separate_forwards = "--separate" in sys.argv; a.monolithic_allreduce = False; a.torch_adam = False; a.hip_graph = False; a.force_dist = False
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "config2"])
dev = torch.device("cuda:0")
job = bench.Job(a, wl, wl["B"], dev, 0, False)
for it in range(5):
    job.step(it)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for it in range(3):
        job.step(5 + it)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::add_", "aten::index_fill_", "aten::clone", "aten::_foreach_add_", "aten::div_", "aten::mul_") and ev.device_type == torch.autograd.DeviceType.CPU:
        stack = [s for s in (ev.stack or []) if "site-packages/torch" not in s and "dist-packages/torch" not in s][:3]
        cnt[(ev.name, " <- ".join(s.split("/")[-1] for s in stack))] += 1
for (name, st), n in cnt.most_common(40):
    print("%5.1f per step  %-18s %s" % (n / 3.0, name, st[:200]))

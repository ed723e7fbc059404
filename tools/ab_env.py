#!/usr/bin/env python3
"""Same-box, same-process A/B of environment switches (csrc/dcn_tuning.h) on the whole training step.

    python tools/ab_env.py --workload config2 --env DCN_BN_REVERSE=0,1,2,3 [--env NAME=a,b ...] --reps 3 --steps 20
    python tools/ab_env.py --workload config1 --graph 0,1 --separate 0,1

Every combination of the given values is one variant; the variants are measured in turn, `reps` times (alternating, so that
clock / temperature drift hits all of them alike), each time on a fresh Job (plans are rebuilt: some switches change what a
plan reserves).  The library re-reads the environment through dcn_reload_env() -- no new process, no torch import per run.
Prints one line per measurement and a summary (median ms/step per variant)."""
import argparse
import itertools
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts the package on the path)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--env", action="append", default=[], help="NAME=v1,v2,...")
    ap.add_argument("--graph", default="0", help="0, 1 or 0,1: replay forward + loss + backward from a captured hipGraph")
    ap.add_argument("--separate", default="0", help="0, 1 or 0,1: two forward calls instead of forward_pair")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0)
    a = ap.parse_args()
    from dcn_hip import _lib, backbone as bb
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    names, values = [], []
    for e in a.env:
        n, v = e.split("=", 1)
        names.append(n)
        values.append(v.split(","))
    names += ["graph", "separate"]
    values += [a.graph.split(","), a.separate.split(",")]
    variants = list(itertools.product(*values))
    wl = dict(bench.WORKLOADS[a.workload])
    if a.batch:
        wl["B"] = a.batch
    res = {v: [] for v in variants}
    for rep in range(a.reps):
        for v in variants:
            kv = dict(zip(names, v))
            for n in names[:-2]:
                os.environ[n] = kv[n]
            _lib.get().dcn_reload_env()
            bb._PLANS.clear()
            args = argparse.Namespace(separate_forwards=kv["separate"] == "1", monolithic_allreduce=False, torch_adam=False,
                                      hip_graph=kv["graph"] == "1")
            job = bench.Job(args, wl, wl["B"], dev, 0, False)
            sec, loss = job.timed(a.warmup, a.steps, 0, False)
            ms = 1e3 * sec / a.steps
            res[v].append(ms)
            print("ab %s rep=%d  %.3f ms/step  %.1f images/s  host %.2f ms  loss %.5f  %s" % (
                " ".join("%s=%s" % t for t in kv.items()), rep, ms, 2 * wl["B"] * 1e3 / ms, job.host_enqueue_ms, float(loss.item()),
                job.graph_note if kv["graph"] == "1" else ""), flush=True)
            del job
            torch.cuda.empty_cache()
    print("# summary (%s, median of %d): " % (a.workload, a.reps))
    base = statistics.median(res[variants[0]])
    for v in variants:
        m = statistics.median(res[v])
        print("#   %-60s %.3f ms/step  %.1f images/s  (%+.2f %% vs first)" % (
            " ".join("%s=%s" % t for t in zip(names, v)), m, 2 * wl["B"] * 1e3 / m, 100.0 * (base / m - 1.0)))


if __name__ == "__main__":
    main()

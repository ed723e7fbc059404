#!/usr/bin/env python3
"""Why does the config-1 two-call variant run host-bound (35 ms/step) inside the default bench process and at 12.7 ms alone?
Replays the bench's order of jobs and prints, per job, ms/step, host enqueue time and the caching allocator's device
allocation counters (a step that calls hipMalloc / hipFree is host-bound and synchronising)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def run(name, wlname, sep, warm, steps):
    wl = dict(bench.WORKLOADS[wlname])
    args = argparse.Namespace(separate_forwards=sep, monolithic_allreduce=False, torch_adam=False, hip_graph=False)
    job = bench.Job(args, wl, wl["B"], torch.device("cuda", 0), 0, False)
    s0 = torch.cuda.memory_stats()
    sec, _ = job.timed(warm, steps, 0, False)
    s1 = torch.cuda.memory_stats()
    print("%-34s %.3f ms/step  host %.2f ms  device mallocs during run %d  frees %d  retries %d  reserved %.1f GB" % (
        name, 1e3 * sec / steps, job.host_enqueue_ms, s1["num_device_alloc"] - s0["num_device_alloc"],
        s1["num_device_free"] - s0["num_device_free"], s1["num_alloc_retries"] - s0["num_alloc_retries"],
        s1["reserved_bytes.all.current"] / 2 ** 30), flush=True)
    return job


def main():
    from dcn_hip import backbone as bb
    bb.set_conv_mode("f16x3")
    if len(sys.argv) > 1 and sys.argv[1] == "config5":
        # config 5 (ResNet50-8s 1280 x 960, B = 2) was seen at 52 ... 63 images/s run to run: same job five times in one process,
        # arenas released in between, with the allocator counters and the SMI clocks next to each run
        import subprocess
        for k in range(5):
            j = run("config5 pair, run %d" % k, "config5", False, 4 if k else 8, 8)
            del j
            torch.cuda.empty_cache()
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=20).stdout.decode()
                print("   " + " | ".join(l.strip() for l in out.splitlines() if any(t in l for t in ("sclk", "mclk", "Power", "Temperature (Sensor junction)")))[:400], flush=True)
            except Exception as e:  # noqa: BLE001
                print("   rocm-smi unavailable:", e)
        return
    keep = os.environ.get("KEEP_HEADLINE", "1") == "1"
    head = run("config2 pair (headline)", "config2", False, 5, 10)
    if not keep:
        del head
        torch.cuda.empty_cache()
    j = run("config1 separate (first)", "config1", True, 8, 20); del j; torch.cuda.empty_cache()
    j = run("config2 separate", "config2", True, 4, 10); del j; torch.cuda.empty_cache()
    j = run("config1 pair", "config1", False, 8, 20); del j; torch.cuda.empty_cache()
    j = run("config1 separate (after pair)", "config1", True, 8, 20); del j; torch.cuda.empty_cache()
    j = run("config1 separate (again)", "config1", True, 8, 20); del j; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

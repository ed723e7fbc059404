"""bench.py's multi-GPU control flow, executed on the CPU box: `python bench.py --gpus 2 --dry-run-cpu`
goes through respawn_under_torchrun -> torch.distributed.run -> two ranks -> rendezvous (gloo) -> per-rank jobs on the
host-emulated kernels -> bucketed gradient all-reduce during backward -> barrier / max-over-ranks timing -> the weak- and
strong-scaling legs -> ONE JSON line from rank 0.  The numbers mean nothing; that every branch of the 8-GPU run executes does."""
import json
import os
import subprocess
import sys

from helpers import ROOT


def test_two_rank_dry_run_through_the_respawn_path():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--dry-run-cpu", "--profile-steps", "0"], capture_output=True, text=True, env=env, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert "DRY RUN" in out["data"] and "DRY RUN" in out["metric"]
    assert out["config"]["parallelism"] == "dp2" and "hostemu" in out["config"]["library"]
    comm = out["communication"]
    assert comm["backend"] == "gloo" and comm["ranks"] == 2
    assert comm["bucketed_steps"] >= 2 and comm["bucket_collectives"] == 3 * comm["bucketed_steps"] and comm["monolithic_steps"] == 0
    assert comm["allreduce_bytes"] > 0 and comm["allreduce_ms"] > 0
    assert len(out["host_enqueue_ms_per_rank"]) == 2 and all(v > 0 for v in out["host_enqueue_ms_per_rank"])
    assert out["variants"]["weak_scaling"]["one_gpu_same_box"] > 0
    assert out["variants"]["strong_scaling_global64"]["global_pairs"] == 4
    assert list(out)[-1] == "summary" and out["summary"]["communication"]["ranks"] == 2
    assert out["value"] > 0 and abs(out["value"] - 2 * 2 * 2 * 1 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]

#!/usr/bin/env python3
"""Training throughput of the dense-correspondence hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the reference's training iteration (dense_correspondence/training/training.py:325-346)
over one synthetic batch already resident in HBM:
    optimizer.zero_grad() -> dcn.forward(img_a), dcn.forward(img_b) (by default as ONE grouped engine call, forward_pair:
    identical values, batch-norm statistics per image batch; --separate-forwards for two calls) -> process_network_output x2
    -> loss_composer (match + masked + background non-match lists, hard-negative scaling) -> backward, with the gradient
    buckets all-reduced over RCCL on a communication stream as the backbone's backward completes them (N > 1) ->
    Adam step (lr 1e-4, weight decay 1e-4, training.yaml:3,6).
Workload at N = 1: BASELINE.json configs[1] -- B = 4 image pairs (8 images / step), 640x480, D = 3, Resnet34_8s, 5000 match
+ 2500 masked + 2500 background non-match pixel pairs per image pair.  At N > 1: BASELINE configs[3] -- global batch
8 N image pairs, every rank runs B = 8 pairs of its own (weak scaling, BN statistics per rank: the reference has no
SyncBN); `value` is the whole-job images / second.  The same line carries, as `variants`, the measurements that complete
the picture: the fp32-MFMA arithmetic and configs[3]'s per-GPU share on ONE GPU (N = 1), the fixed-global-batch-64
strong-scaling point (N > 1), and `allreduce_ms` / `communication` (collective alone, exposed per step, RCCL rank count).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel = the gather-GEMM
convolution, forward + dgrad: conv_gemm_hl_kernel on the wide layers and conv_gemm_f16_kernel on the others in the default
split-fp16 arithmetic / conv_gemm_kernel with --conv-mode fp32; algorithmic FLOPs and per-launch durations from HIP events
recorded by the engine on the launch stream; `roofline.hl_kernel` is the wide-layer kernel alone) and
`cpu_baseline` (the oracle's step on this box's host cores, bounded sample).  `--workload pairgen` measures the device
pair generator instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch-dense-correspondence_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# synthetic-input constants.  Kept here so the measured path never imports `oracle`;
# tests/test_abi.py (test_synth_constants_consistent_with_bench) checks they equal oracle/synth.py.
DEFAULT_IMAGE_MEAN = [0.5573105812072754, 0.37420374155044556, 0.37020164728164673]
DEFAULT_IMAGE_STD_DEV = [0.24336038529872894, 0.2987397611141205, 0.31875079870224]
LOSS_CONFIG = {"M_masked": 0.5, "M_background": 0.5, "M_pixel": 50, "match_loss_weight": 1.0,
               "non_match_loss_weight": 1.0, "use_l2_pixel_loss_on_masked_non_matches": False,
               "use_l2_pixel_loss_on_background_non_matches": False, "scale_by_hard_negatives": True,
               "scale_by_hard_negatives_DIFFERENT_OBJECT": True, "alpha_triplet": 0.1}
WORKLOADS = {
    "config2": dict(B=4, H=480, W=640, D=3, Pm=5000, Pk=2500, Pg=2500, backbone="Resnet34_8s",
                    desc="BASELINE configs[1]: B=4 pairs 640x480 D=3 Resnet34_8s 5000/2500/2500 pairs"),
    "config1": dict(B=1, H=480, W=640, D=3, Pm=1000, Pk=500, Pg=500, backbone="Resnet34_8s",
                    desc="BASELINE configs[0]: B=1 pair 640x480 D=3 Resnet34_8s 1000/500/500 pairs"),
    "config3": dict(B=32, H=480, W=640, D=16, Pm=10000, Pk=50000, Pg=50000, backbone="Resnet34_8s",
                    desc="BASELINE configs[2]: B=32 pairs 640x480 D=16 Resnet34_8s 10000/50000/50000 pairs"),
    "config4": dict(B=8, H=480, W=640, D=3, Pm=5000, Pk=2500, Pg=2500, backbone="Resnet34_8s",
                    desc="BASELINE configs[3] per-GPU share: B=8 pairs 640x480 D=3 Resnet34_8s"),
    "config5": dict(B=2, H=960, W=1280, D=32, Pm=2500, Pk=5000, Pg=5000, backbone="Resnet50_8s", masked=True,
                    desc="BASELINE configs[4] per-GPU share: B=2 pairs 1280x960 D=32 Resnet50_8s, masked / background "
                         "non-match sampling (2500 matches x 2 masked + 2 background non-matches)"),
    "tiny": dict(B=1, H=96, W=128, D=3, Pm=500, Pk=250, Pg=250, backbone="Resnet34_8s", desc="smoke-size workload"),
    # --dry-run-cpu only: a Resnet18_8s of base width 8 at 64 x 64 (what the host-emulated kernels step through in seconds)
    "dryrun": dict(B=2, H=64, W=64, D=3, Pm=60, Pk=30, Pg=30, backbone="DryRunNet", base_width=8,
                   desc="DRY RUN: B=2 pairs 64x64 D=3 Resnet18_8s of base width 8, host-emulated kernels"),
}
F16_MFMA_PEAK_TFLOPS = 2516.6   # v_mfma_f32_32x32x16_f16: 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz ("~2.5 PF dense")
F16X3_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3.0
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz




def make_masked_index_lists(B, H, W, Pm, per_match, gen):
    """BASELINE config 5 "masked-background non-match sampling" (SURVEY.md section 8d): a random elliptic object mask
    (~15 % of the image) per image; matches lie on the mask in both images; for every match ``per_match`` masked
    non-matches (b index ON the mask) and ``per_match`` background non-matches (b index OFF the mask), with the a index
    repeated per match -- the grouped layout of spartan_dataset_masked.py:841-858."""
    out = []
    ys = torch.arange(H).view(H, 1).float()
    xs = torch.arange(W).view(1, W).float()
    for _ in range(B):
        masks = []
        for _side in range(2):
            cy = (0.3 + 0.4 * torch.rand(1, generator=gen)) * H
            cx = (0.3 + 0.4 * torch.rand(1, generator=gen)) * W
            ry, rx = 0.22 * H, 0.22 * W
            masks.append((((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 <= 1.0).reshape(-1))
        on_a, on_b = masks[0].nonzero().reshape(-1), masks[1].nonzero().reshape(-1)
        off_b = (~masks[1]).nonzero().reshape(-1)
        pick = lambda pool, n: pool[torch.randint(0, pool.numel(), (n,), generator=gen)]
        ma, mb = pick(on_a, Pm), pick(on_b, Pm)
        rep = ma.repeat_interleave(per_match)
        d = {"matches_a": ma, "matches_b": mb,
             "masked_non_matches_a": rep.clone(), "masked_non_matches_b": pick(on_b, Pm * per_match),
             "background_non_matches_a": rep.clone(), "background_non_matches_b": pick(off_b, Pm * per_match),
             "blind_non_matches_a": torch.tensor([-1], dtype=torch.int64),
             "blind_non_matches_b": torch.tensor([-1], dtype=torch.int64)}
        out.append(d)
    return out


def make_batch(B, H, W, Pm, Pk, Pg, seed, masked=False):
    """Seeded synthetic batch: mean/std-normalised uniform images, uniformly drawn int64 pixel pairs."""
    gen = torch.Generator().manual_seed(seed)
    mean = torch.tensor(DEFAULT_IMAGE_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(DEFAULT_IMAGE_STD_DEV).view(1, 3, 1, 1)
    img_a = (torch.rand(B, 3, H, W, generator=gen) - mean) / std
    img_b = (torch.rand(B, 3, H, W, generator=gen) - mean) / std
    if masked:
        return img_a, img_b, make_masked_index_lists(B, H, W, Pm, Pk // Pm, gen)
    lists = []
    for _ in range(B):
        d = {}
        for name, n in (("matches", Pm), ("masked_non_matches", Pk), ("background_non_matches", Pg)):
            for side in ("a", "b"):
                d[name + "_" + side] = (torch.randint(0, H * W, (n,), generator=gen, dtype=torch.int64) if n > 0
                                        else torch.tensor([-1], dtype=torch.int64))
        d["blind_non_matches_a"] = torch.tensor([-1], dtype=torch.int64)
        d["blind_non_matches_b"] = torch.tensor([-1], dtype=torch.int64)
        lists.append(d)
    return img_a, img_b, lists


def as_tuples(lists):
    keys = ("matches_a", "matches_b", "masked_non_matches_a", "masked_non_matches_b", "background_non_matches_a",
            "background_non_matches_b", "blind_non_matches_a", "blind_non_matches_b")
    return [tuple(L[k] for k in keys) for L in lists]


def usable_cpus():
    """Host cores this process may really use: min(affinity mask, cgroup CPU quota).  (os.cpu_count() reports the
    machine's logical CPUs -- 256 on the GPU box -- while the container is capped by cgroup cpu.max; running torch
    with more threads than the quota oversubscribes it by an order of magnitude.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(wl, steps, warmup):
    """The oracle's training step (oracle/step.py, the CPU restatement of the reference) on this box's host cores.
    Bounded sample (~10-30 s of CPU work): the workload's FULL batch when it has at most 4 image pairs (config 1, config 2),
    otherwise ONE image pair of its shapes (BN statistics then over 1 image instead of B: same FLOPs per image)."""
    from oracle import resnet_dilated_oracle, step as ostep, synth
    cores = usable_cpus()
    torch.set_num_threads(cores)
    model = resnet_dilated_oracle.build(wl["backbone"], wl["D"], seed=0)
    model.train()
    Bc = wl["B"] if wl["B"] <= 4 else 1
    img_a, img_b, lists = synth.make_batch(Bc, wl["H"], wl["W"], wl["Pm"], wl["Pk"], wl["Pg"], seed=1,
                                           masked=wl.get("masked", False))
    sec = ostep.time_cpu_step(model, img_a, img_b, lists, synth.LOSS_CONFIG, steps=steps, warmup=warmup)
    return {"value": 2.0 * Bc / sec, "unit": "images/s", "cores": cores, "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": "oracle (py3 CPU restatement of training.py:325-346, torch %s, fp32, forward + loss + backward + Adam) on "
                      "%s of the workload's shapes (%dx%d, D=%d, %d/%d/%d pixel pairs per image pair), %d warm-up + median "
                      "of up to %d steps (30 s budget), %.2f s/step"
                      % (torch.__version__.split("+")[0], "the full batch of %d image pairs" % Bc if Bc == wl["B"] else
                         "1 of the %d image pairs" % wl["B"], wl["W"], wl["H"], wl["D"], wl["Pm"], wl["Pk"], wl["Pg"],
                         warmup, steps, sec)}


def pairgen_bench(args):
    """`--workload pairgen`: device pair generation at the training configuration's sizes
    (training.yaml:9,17-21: 10 000 matching attempts, 150 masked + 150 background non-matches per match) on a seeded
    synthetic scene; cpu_baseline = the oracle (the reference's CPU algorithm, restated) on the host cores."""
    import numpy as np
    from dense_correspondence.correspondence_tools import correspondence_finder as cf
    H, W = 480, 640
    rng = np.random.RandomState(2)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)

    def surf():
        d = 900 + 150 * np.sin(xs / (60 + 40 * rng.rand())) + 120 * np.cos(ys / (50 + 30 * rng.rand())) + 40 * rng.rand()
        d[rng.rand(H, W) < 0.02] = 0
        return d.astype(np.uint16)

    def pose(ry, t):
        T = np.eye(4)
        T[:3, :3] = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
        T[:3, 3] = t
        return T
    depth_a, depth_b = surf(), surf()
    pose_a, pose_b = pose(0.01, [0.02, 0.0, 0.01]), pose(-0.06, [0.08, -0.03, 0.05])
    mask_np = np.zeros((H, W), np.float32)
    mask_np[150:380, 180:470] = 1.0
    mask = torch.tensor(mask_np).cuda()
    da = torch.from_numpy(depth_a.view(np.int16)).cuda()
    db = torch.from_numpy(depth_b.view(np.int16)).cuda()

    def one_pair():
        uv_a, uv_b = cf.batch_find_pixel_correspondences(da, pose_a, db, pose_b, num_attempts=10000, img_a_mask=mask)
        cf.create_non_correspondences(uv_b, (H, W), 150, img_b_mask=mask)
        cf.create_non_correspondences(uv_b, (H, W), 150, img_b_mask=1 - mask)
        return uv_a[0].numel()

    for _ in range(args.warmup):
        nmatch = one_pair()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pair()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    out = {"metric": "image pairs/sec, device pair generation 640x480 (10000 attempts, 2 x 150 non-matches per match)",
           "value": 1e3 / ms, "unit": "image pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/int64",
           "data": "synthetic", "config": {"workload": "pairgen", "matches_per_pair": nmatch,
                                           "non_match_samples_per_pair": 300 * nmatch, "host_syncs_per_pair": 4},
           "roofline": None, "cpu_baseline": None}
    if args.cpu_baseline_steps > 0:
        from oracle import correspondence_oracle as co
        cores = usable_cpus()
        torch.set_num_threads(cores)
        mk = torch.tensor(mask_np)
        t0 = time.perf_counter()
        for _ in range(args.cpu_baseline_steps):
            lst = torch.nonzero(mk.reshape(-1)).squeeze(1)
            sel = lst[torch.floor(torch.rand(10000) * lst.numel()).long()]
            ua, _ = co.find_correspondences_for_candidates(depth_a, pose_a, depth_b, pose_b, sel % W, sel // W)
            n = ua[0].numel() * 150
            co.create_non_correspondences(ua[0].numel(), (H, W), 150, mk, torch.rand(n))
            co.create_non_correspondences(ua[0].numel(), (H, W), 150, 1 - mk, torch.rand(n))
        sec = (time.perf_counter() - t0) / args.cpu_baseline_steps
        out["cpu_baseline"] = {"value": 1.0 / sec, "unit": "image pairs/s", "cores": cores, "kind": "port",
                               "sample": "oracle/correspondence_oracle.py (correspondence_finder.py:276-619 restated, torch CPU "
                                         "ops), same scene and sizes, mean of %d pairs, %.3f s/pair" % (args.cpu_baseline_steps, sec)}
    print(json.dumps(out), flush=True)


class _HostEvent(object):
    """--dry-run-cpu: torch.cuda.Event stand-in (host clock)."""

    def __init__(self, enable_timing=True):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


def enter_cpu_dry_run(args):
    """`--dry-run-cpu`: NOT a measurement.  The whole bench choreography -- respawn under torch.distributed.run, rendezvous,
    per-rank jobs, bucketed gradient all-reduce, barriers, max-over-ranks timing, the weak- / strong-scaling legs, the JSON
    line -- on a box without a GPU: host-emulated kernels (tests/hostemu, the same kernel sources compiled for the host), the
    `gloo` backend, a narrow network.  It exists so that the multi-GPU control flow, which the authoring container cannot run
    on RCCL, is executed by the CPU test-suite (tests/test_bench_dry_run.py); every number it prints is meaningless."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hostemu"))
    import build_emu
    from dcn_hip import _lib
    _lib.load(build_emu.build())
    import pytorch_segmentation_detection.models.resnet_dilated as rd

    class DryRunNet(rd.Resnet18_8s):
        def __init__(self, num_classes):
            super(DryRunNet, self).__init__(num_classes=num_classes, base_width=8)
    DryRunNet.arch, DryRunNet.attr = rd.Resnet18_8s.arch, rd.Resnet18_8s.attr
    rd.DryRunNet = DryRunNet
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    torch.cuda.Event = _HostEvent
    args.workload = "dryrun"
    args.cpu_baseline_steps = 0
    args.hip_graph = False
    args.conv_mode = "f16x3"


def respawn_under_torchrun(n, dry_run_cpu=False):
    """`python bench.py --gpus N` with no torch.distributed.run around it: start the N ranks ourselves (one process per
    GPU, RCCL over xGMI) and relay rank 0's JSON line."""
    import socket
    import subprocess
    if not dry_run_cpu and (not torch.cuda.is_available() or torch.cuda.device_count() < n):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" %
                         (n, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


class Job(object):
    """One workload on this rank: model, optimizer, flat gradients, resident synthetic batch, and the step closure."""

    def __init__(self, args, wl, B, dev, rank, use_dist, separate_forwards=None):
        from dcn_hip.distributed import FlatGradients, broadcast_module
        self.separate = separate = args.separate_forwards if separate_forwards is None else bool(separate_forwards)
        from dcn_hip.loss import PairLists
        from dcn_hip.optim import Adam
        from dense_correspondence.loss_functions import loss_composer
        from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
        from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
        self.args, self.wl, self.B, self.dev = args, wl, B, dev
        H, W, D = wl["H"], wl["W"], wl["D"]
        torch.manual_seed(0)
        cfg = {"descriptor_dimension": D, "image_width": W, "image_height": H,
               "backbone": {"model_class": "Resnet", "resnet_name": wl["backbone"]}}
        self.dcn = dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False)  # .cuda().train() like network.py:435
        dcn.to(dev)
        broadcast_module(dcn)
        self.pcl = pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=LOSS_CONFIG)
        # --force-dist with ONE rank: the bucketed schedule anyway, its all-reduces issued over the one-rank RCCL group, so that
        # a single GPU runs the communication-stream path the 8-GPU job runs (communication.bucketed_steps > 0)
        one_rank_dist = use_dist and (not dist.is_initialized() or dist.get_world_size() == 1)
        self.bucketed_default = False if args.monolithic_allreduce else (True if one_rank_dist else None)
        self.grads = grads = FlatGradients(dcn, bucketed=self.bucketed_default, single_rank_collectives=one_rank_dist)
        self.opt = opt = grads.attach((torch.optim.Adam if args.torch_adam else Adam)(
            dcn.parameters(), lr=1.0e-4, weight_decay=1.0e-4))   # training.py:133-145
        img_a, img_b, lists = make_batch(B, H, W, wl["Pm"], wl["Pk"], wl["Pg"], seed=1 + rank, masked=wl.get("masked", False))
        self.img_a, self.img_b = img_a, img_b = img_a.to(dev), img_b.to(dev)
        self.pair_lists = pair_lists = PairLists.from_lists(as_tuples(lists), dev, hw=H * W)
        self.match_type = match_type = 0  # SINGLE_OBJECT_WITHIN_SCENE
        self.comm = True

        def forward_backward():
            opt.zero_grad()              # training.py:325 (the attached flat buffer is zeroed in place, one kernel)
            if separate:                 # literally training.py:329-333
                ya, yb = dcn.forward(img_a), dcn.forward(img_b)
            else:                        # the same two network calls as ONE grouped launch sequence (BN statistics per image batch)
                ya, yb = dcn.forward_pair(img_a, img_b)
            pa = dcn.process_network_output(ya, B)
            pb = dcn.process_network_output(yb, B)
            loss, terms, hard = loss_composer.get_loss_batched(pcl, match_type, pa, pb, pair_lists)
            loss.backward()              # (world > 1: the gradient buckets are all-reduced on the communication stream from
            return loss                  #  inside this call, as the backbone's backward completes them)
        self.forward_backward = forward_backward

        # The ~600 launches of forward + loss + backward are captured ONCE into a hipGraph (inputs, lists, parameters and
        # the flat gradient buffer are static tensors; a training loop copies each new batch into them) and replayed per
        # step; the gradient all-reduce and the optimizer stay eager.  Falls back to eager launches if capture fails.
        self.graph, self.static_loss, self.graph_note = None, None, "off"
        if args.hip_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        forward_backward()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.static_loss = forward_backward()
                self.graph_note = "forward + loss + backward replayed from one captured hipGraph"
            except Exception as e:  # noqa: BLE001 -- any capture problem: run eagerly, say so in the JSON
                self.graph, self.graph_note = None, "capture failed (%s): eager launches" % (str(e).splitlines()[0][:120],)
                torch.cuda.synchronize()

    def step(self, it, eager=False):
        if it % 250 == 0 and it > 0:  # training.yaml:4-5 step decay
            for g in self.opt.param_groups:
                g["lr"] *= 0.9
        if self.graph is not None and not eager:
            self.graph.replay()
            loss = self.static_loss
        else:
            loss = self.forward_backward()
        if self.comm:
            self.grads.all_reduce_mean()   # joins the communication stream (bucketed) or ONE collective (monolithic)
        self.opt.step()
        return loss

    def timed(self, warmup, steps, first_it, use_dist):
        """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks.  -> (seconds, last loss)"""
        loss = None
        for it in range(warmup):
            loss = self.step(first_it + it)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(steps):
            loss = self.step(first_it + warmup + it)
        self.host_enqueue_ms = 1e3 * (time.perf_counter() - t0) / steps   # (host time to ENQUEUE a step; the GPU runs behind)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # (outside the timed region) the same enqueue against an IDLE GPU, one step at a time with a synchronisation in
        # between: the launch queue never fills, so this is the host's own cost of a step -- Python, bindings, the engine's
        # C++ and the HIP runtime's launch path -- without the back-pressure that host_enqueue_ms includes when the GPU is
        # the slower side.  Two steps; the smaller one.
        idle = []
        for it in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            self.step(first_it + warmup + steps + it)
            idle.append(1e3 * (time.perf_counter() - t1))
        torch.cuda.synchronize()
        self.host_enqueue_idle_ms = min(idle)
        return elapsed, loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS) + ["pairgen"],
                    help="default: config2 (BASELINE configs[1]) on one GPU, config4 (configs[3]: B = 8 pairs per GPU) on several")
    ap.add_argument("--batch", type=int, default=0, help="override image pairs per GPU per step")
    ap.add_argument("--cpu-baseline-steps", type=int, default=5, help="0 disables the CPU baseline leg")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra steps with per-launch HIP events (roofline)")
    ap.add_argument("--conv-mode", default="f16x3", choices=["f16x3", "fp32"],
                    help="convolution arithmetic (include/dcn_hip.h): split-fp16 on the fp16 MFMA pipe with fp32-level "
                         "accuracy (default), or fp32 MFMA")
    ap.add_argument("--hip-graph", action="store_true",
                    help="replay forward + loss + backward (~350 launches) from one captured hipGraph instead of launching "
                         "them every step; measured: no gain at B = 4 and 4 %% SLOWER at B = 1 (profiles/r4a_config1_graph_separate_ab.txt) "
                         "-- the step is kernel-bound, the host stays ahead of the GPU")
    ap.add_argument("--separate-forwards", action="store_true",
                    help="forward(img_a) and forward(img_b) as two engine calls (default: one grouped call with identical results)")
    ap.add_argument("--torch-adam", action="store_true", help="optimizer.step() through torch.optim.Adam instead of dcn_adam_step")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even with one rank (smoke-tests the collective path)")
    ap.add_argument("--monolithic-allreduce", action="store_true",
                    help="ONE all-reduce after backward instead of the bucketed, overlapped schedule")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="NOT a measurement: run the bench's control flow (respawn, rendezvous, bucketed all-reduce, barriers, "
                         "scaling legs, JSON line) on CPU with host-emulated kernels and gloo -- see enter_cpu_dry_run")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra measurements (fp32-MFMA mode, config 4 on one "
                                                                "GPU, strong-scaling point) that ride on the JSON line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus, args.dry_run_cpu)   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run_cpu:
        enter_cpu_dry_run(args)
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the dense-correspondence hot path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    rccl = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dry_run_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()
        import ctypes
        ctypes.CDLL(None).fflush(None)  # RCCL's version banner sits in the C stdio buffer: emit it now, not after the JSON line
        rccl = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                "nccl_version": None if args.dry_run_cpu else ".".join(str(v) for v in torch.cuda.nccl.version())}

    if args.workload is None:
        args.workload = "config2" if world == 1 else "config4"
    if args.workload == "pairgen":
        if world > 1:
            raise SystemExit("--workload pairgen is a single-GPU measurement")
        return pairgen_bench(args)
    from dcn_hip import _lib, backbone as bb
    bb.set_conv_mode(args.conv_mode)
    from dense_correspondence.loss_functions import loss_composer
    info = _lib.library_info()
    assert args.dry_run_cpu or not info["hostemu"], "bench.py must run the gfx950 library"

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    B, H, W, D = wl["B"], wl["H"], wl["W"], wl["D"]
    job = Job(args, wl, B, dev, rank, use_dist)
    dcn, pcl, grads, pair_lists, match_type = job.dcn, job.pcl, job.grads, job.pair_lists, job.match_type
    img_a, img_b = job.img_a, job.img_b
    graph = job.graph
    graph_note = job.graph_note
    step = job.step

    elapsed, loss = job.timed(args.warmup, args.steps, 0, use_dist)
    host_enqueue_ms = job.host_enqueue_ms
    final_loss = float(loss.item())
    # every rank's host time to enqueue a step (8 ranks share the box's usable cores: a rank whose enqueue time approaches the
    # step time is host-bound) -- gathered so that the multi-rank line shows it per rank
    host_enqueue_per_rank = [host_enqueue_ms]
    if use_dist:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = host_enqueue_ms
        dist.all_reduce(t)
        host_enqueue_per_rank = [float(v) for v in t.tolist()]

    # ---- gradient all-reduce: the collective alone (whole flat buffer, RCCL), and what the step really pays for it
    # (same steps with the communication switched off; the bucketed schedule hides all but the last bucket)
    comm = None
    if use_dist:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        for _ in range(3):
            dist.all_reduce(grads.flat)
        dist.barrier()
        e0.record()
        for _ in range(reps):
            dist.all_reduce(grads.flat)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / reps
        job.comm = False
        grads.bucketed = False
        nocomm, _ = job.timed(2, max(4, args.steps // 2), args.warmup + args.steps, use_dist)
        job.comm = True
        grads.bucketed = job.bucketed_default
        broadcast = __import__("dcn_hip.distributed", fromlist=["broadcast_module"]).broadcast_module
        broadcast(dcn)   # the un-synchronised steps let the replicas drift: re-align before anything else is measured
        comm = {"allreduce_ms": ar_ms, "allreduce_bytes": grads.flat.numel() * 4,
                "allreduce_busbw_GBps": (grads.flat.numel() * 4 * 2 * (world - 1) / max(world, 1)) / (ar_ms * 1e-3) / 1e9 if world > 1 else None,
                "schedule": "monolithic: one all-reduce after backward" if args.monolithic_allreduce or graph is not None else
                "bucketed: fc+layer4 | layer3 | rest, each all-reduced on a communication stream as soon as the backbone's "
                "backward has produced it (dcn_plan_stream_wait_grad_bucket)",
                "ms_per_step_without_allreduce": 1e3 * nocomm / max(4, args.steps // 2),
                "allreduce_exposed_ms": 1e3 * elapsed / args.steps - 1e3 * nocomm / max(4, args.steps // 2),
                "bucketed_steps": grads.stats["bucketed_steps"], "monolithic_steps": grads.stats["monolithic_steps"],
                "bucket_collectives": grads.stats["bucket_collectives"]}
        comm.update(rccl)

    # ---- roofline of the dominant kernel: extra steps, every conv_gemm / conv_wgrad launch bracketed by HIP events
    def measure_roofline(job_, conv_mode, first_it):
        wl_, B_ = job_.wl, job_.B
        arch_ = "Resnet18_8s" if wl_["backbone"] == "DryRunNet" else wl_["backbone"]
        plan = bb.get_plan(arch_, wl_.get("base_width", 64), B_, wl_["H"], wl_["W"], wl_["D"]) if job_.separate else \
            bb.get_plan(arch_, wl_.get("base_width", 64), 2 * B_, wl_["H"], wl_["W"], wl_["D"], 2)
        plan.profile_begin()
        for it in range(args.profile_steps):
            job_.step(first_it + it, eager=True)   # the engine's per-launch events do not exist inside a graph
        prof = plan.profile_end()
        job_.last_profile = prof
        ms, n, fl = prof["conv_gemm"]
        wms, wn, wfl = prof["conv_wgrad"]
        hms, hn, hfl = prof["conv_gemm_hl"]
        if not (n > 0 and ms > 0):
            return None
        achieved = fl / (ms * 1e-3) / 1e12
        if conv_mode == "fp32":
            peak, kern = FP32_MFMA_PEAK_TFLOPS, "conv_gemm_kernel (fp32 v_mfma_f32_32x32x2_f32; forward + dgrad)"
            peak_note = "fp32 MFMA dense peak"
        else:
            # one fp32-accurate multiply-add = 3 fp16 MFMA products (hi*hi + hi*lo + lo*hi): the ceiling for
            # ALGORITHMIC flops is a third of the fp16 pipe's dense peak
            peak, kern = F16X3_PEAK_TFLOPS, ("gather-GEMM convolution, forward + dgrad (3 fp16 MFMAs per product: v_mfma_f32_16x16x32_f16 in the hl32 "
                                             "kernels, v_mfma_f32_32x32x16_f16 in the fp32-operand ones): conv_gemm_hl_kernel on the wide layers "
                                             "(pre-split hl32 operands by LDS-DMA, 320 x 256, 256 x 256 or 192 x 256 tiles by quantisation on the 256 CUs), "
                                             "conv_gemm_hlx_kernel (160 x 256 / 160 x 128 tiles, K split) where those leave CUs idle, "
                                             "conv_gemm_f16_kernel on the others")
            peak_note = "fp16 MFMA dense peak %.1f / 3 products per fp32-accurate MAC (fp32 MFMA peak: %.1f)" % (
                F16_MFMA_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS)
        # HBM-side traffic of the dominant kernel cannot be measured from inside the process (PMC counters need
        # rocprofv3): it is the COMMITTED measurement of this very command (profiles/, collected and corrected per
        # MI355X_MICROARCH.md's HBM section), attached only when workload / arithmetic / call pattern match -- the
        # field name says so: it was not measured in this run
        traffic, traffic_src, traffic_hl = None, None, None
        for name in ("r6_hbm_counters.json", "r5_hbm_counters.json", "r4_hbm_counters.json", "r3_hbm_counters.json", "r2m_hbm_counters.json", "r2_hbm_counters.json", "r1g_hbm_counters.json"):
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", name)))
                if (rec["workload"] == args.workload and rec["conv_mode"] == conv_mode and not args.batch and job_ is job and
                        (rec["forward_calls"] == "pair") == (not job_.separate)):
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_hl = (rec.get("hl_kernel") or {}).get("hbm_bytes_per_launch")
                    traffic_src = "committed rocprofv3 --pmc measurement of this command, profiles/%s: %s" % (name, rec["correction"])
                    break
            except (OSError, KeyError, ValueError):
                pass
        return {"bound": "mfma", "kernel": kern,
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "peak_note": peak_note,
                "frac": achieved / peak, "traffic": traffic, "traffic_from_committed_profile": traffic_src,
                "launches_per_step": n / args.profile_steps, "avg_launch_us": 1e3 * ms / n,
                "algorithmic_gflop_per_launch": fl / n / 1e9,
                "kernel_ms_per_step": ms / args.profile_steps,
                "hl_kernel": None if not hn else {
                    "kernel": "conv_gemm_hl_kernel (conv_hl_kernels.hip): the wide layers' share of the launches above",
                    "achieved": hfl / (hms * 1e-3) / 1e12, "frac": hfl / (hms * 1e-3) / 1e12 / peak,
                    "launches_per_step": hn / args.profile_steps, "avg_launch_us": 1e3 * hms / hn,
                    "algorithmic_gflop_per_launch": hfl / hn / 1e9, "kernel_ms_per_step": hms / args.profile_steps,
                    "traffic": traffic_hl},
                "conv_wgrad": {"kernel": "weight gradients: conv_wgrad_hl_kernel (256 x 256 tiles, wide layers), conv_wgrad_hlrp_kernel (row-window "
                                         "kernel, 64 / 128-channel 3 x 3 layers), conv_wgrad_f16_kernel (the others), slab reduce launches included",
                               "achieved": (wfl / (wms * 1e-3) / 1e12) if wms > 0 else None,
                               "frac": (wfl / (wms * 1e-3) / 1e12 / peak) if wms > 0 else None,
                               "launches_per_step": wn / args.profile_steps,
                               "avg_launch_us": (1e3 * wms / wn) if wn else None,
                               "kernel_ms_per_step": wms / args.profile_steps}}

    HBM_PEAK_GBPS = 8000.0

    def elementwise_roofline(prof):
        """K7 / K8: the streaming passes of the step against the HBM roofline -- algorithmic bytes (4 B x
        elements a pass reads and writes once; the launchers of elementwise_kernels.hip state them) / the pass's launch
        durations (per-launch HIP events of the profiled steps, serial schedule) / 8 TB/s."""
        if prof is None:
            return None
        rows = {}
        tot_b, tot_ms = 0.0, 0.0
        for k in ("bn_apply", "bn_bwd_reduce", "bn_bwd_apply", "resample"):
            ms, n, b = prof[k]
            if n == 0 or ms <= 0:
                continue
            rows[k] = {"GBps": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       "launches_per_step": n / args.profile_steps, "avg_launch_us": 1e3 * ms / n,
                       "kernel_ms_per_step": ms / args.profile_steps, "algorithmic_MB_per_step": b / args.profile_steps / 1e6}
            tot_b += b
            tot_ms += ms
        if tot_ms <= 0:
            return None
        fms, fn, _ = prof["bn_finalize"]
        return {"bound": "hbm", "kernel": "bn_apply_kernel, bn_bwd_reduce_kernel, bn_bwd_apply(_blocked)_kernel (batch norm forward / "
                                          "backward streaming passes, K7) + max pool / bilinear upsample / input layout passes (K8)",
                "achieved": tot_b / (tot_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel_ms_per_step": tot_ms / args.profile_steps,
                "algorithmic_MB_per_step": tot_b / args.profile_steps / 1e6, "passes": rows,
                "bn_finalize": {"launches_per_step": fn / args.profile_steps, "avg_launch_us": (1e3 * fms / fn) if fn else None,
                                "kernel_ms_per_step": fms / args.profile_steps, "note": "per-channel finalize kernels: latency-bound, no byte count"}}

    def step_breakdown(job_, prof, ms_per_step_):
        """Kernel-time sum of a step next to its wall time (is the step launch-bound?): every ENGINE launch from the profiled
        steps (serial schedule, per-launch events), the loss call and the optimizer step timed on their own with events."""
        if prof is None:
            return None
        eng_ms = sum(v[0] for k, v in prof.items() if k != "conv_gemm_hl") / args.profile_steps
        eng_n = sum(v[1] for k, v in prof.items() if k != "conv_gemm_hl") / args.profile_steps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        job_.opt.zero_grad()
        loss_ = job_.forward_backward()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            job_.opt.step()
            job_.opt.zero_grad()
        e1.record()
        torch.cuda.synchronize()
        opt_ms = e0.elapsed_time(e1) / reps
        with torch.no_grad():
            da_ = job_.dcn.process_network_output(job_.dcn.forward(job_.img_a), job_.B).detach().contiguous()
            db_ = job_.dcn.process_network_output(job_.dcn.forward(job_.img_b), job_.B).detach().contiguous()
        da_.requires_grad_(True)
        db_.requires_grad_(True)
        def one():
            l = loss_composer.get_loss_batched(job_.pcl, job_.match_type, da_, db_, job_.pair_lists)[0]
            return torch.autograd.grad(l, [da_, db_])
        for _ in range(3):
            one()
        e0.record()
        for _ in range(reps):
            one()
        e1.record()
        torch.cuda.synchronize()
        loss_ms = e0.elapsed_time(e1) / reps
        ksum = eng_ms + opt_ms + loss_ms
        return {"ms_per_step": ms_per_step_, "kernel_ms_sum": ksum, "engine_kernel_ms": eng_ms, "engine_launches_per_step": eng_n,
                "loss_call_ms": loss_ms, "optimizer_ms": opt_ms, "host_enqueue_ms_per_step": job_.host_enqueue_ms,
                "host_enqueue_idle_gpu_ms_per_step": getattr(job_, "host_enqueue_idle_ms", None),
                "engine_ms_by_category": {k: v[0] / args.profile_steps for k, v in prof.items()},
                "engine_launches_by_category": {k: v[1] / args.profile_steps for k, v in prof.items()},
                "note": "kernel_ms_sum = engine launches one by one (serial schedule, HIP events around each) + loss forward/backward "
                        "call + optimizer.step()/zero_grad(), each timed alone; ms_per_step is the timed region of the default "
                        "schedule (weight-gradient GEMMs overlapped on a side stream).  ms_per_step >> kernel_ms_sum would mean "
                        "launch / dependency gaps; host_enqueue_ms_per_step close to ms_per_step would mean a host-bound step"}

    it_next = args.warmup + args.steps + 64
    roofline = measure_roofline(job, args.conv_mode, it_next) if args.profile_steps > 0 else None
    headline_profile = getattr(job, "last_profile", None)
    roofline_elementwise = elementwise_roofline(headline_profile)
    headline_breakdown = step_breakdown(job, headline_profile, 1e3 * elapsed / args.steps) if headline_profile else None

    # ---- HBM roofline of the loss gather (K9): forward + backward of the fused contrastive loss alone, on the
    # descriptor maps of the last step, timed with events on the launch stream
    loss_roof = None
    if args.profile_steps > 0:
        with torch.no_grad():
            da = dcn.process_network_output(dcn.forward(img_a), B).detach().contiguous()
            db = dcn.process_network_output(dcn.forward(img_b), B).detach().contiguous()
        da.requires_grad_(True)
        db.requires_grad_(True)
        reps = 20
        def loss_fwd_bwd(a_, b_, lists_, pcl_):
            # (torch.autograd.grad: the kernel's two gradient maps are the result -- `.backward()` on leaf tensors would add an
            #  AccumulateGrad pass over both maps per call that the training step does not have)
            l = loss_composer.get_loss_batched(pcl_, match_type, a_, b_, lists_)[0]
            return torch.autograd.grad(l, [a_, b_])
        for _ in range(3):
            loss_fwd_bwd(da, db, pair_lists, pcl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            loss_fwd_bwd(da, db, pair_lists, pcl)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        npairs = pair_lists.total
        pair_bytes = (16 * D + 16) * npairs              # 2 descriptor reads + 2 int64 indices + 2 gradient accumulations
        fill_bytes = 2 * B * H * W * D * 4               # zero-fill of the two dense gradient maps
        loss_roof = {"bound": "hbm", "kernel": "loss_fwd_kernel + loss_finalize_kernel + loss_mean_kernel + order-independent backward (int64 fill, loss_bwd_vmax / _saved_exact / _exact_convert_kernel; fp32 atomics where the int64 maps would exceed 256 MB)",
                     "achieved": (pair_bytes + fill_bytes) / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": (pair_bytes + fill_bytes) / (ms * 1e-3) / 1e9 / 8000.0, "us_per_call": 1e3 * ms,
                     "pixel_pairs": npairs, "algorithmic_bytes": {"pairs": pair_bytes, "zero_fill": fill_bytes},
                     "note": "latency-bound at this size: %d random %d-byte gathers per call" % (2 * npairs, 4 * D)}

    # ---- the same loss measurement at BASELINE configs[2]'s list sizes (B = 32 pairs, D = 16, 10 000 + 50 000 + 50 000
    # pixel pairs each: the only config where the gather moves enough bytes to talk about HBM bandwidth), on
    # random descriptor maps resident in HBM
    if loss_roof is not None and world == 1 and not args.no_variants and args.workload != "config3" and not args.dry_run_cpu:
        from dcn_hip.loss import PairLists
        c3 = WORKLOADS["config3"]
        B3, D3, HW3 = c3["B"], c3["D"], c3["H"] * c3["W"]
        gen = torch.Generator().manual_seed(3)
        lists3 = []
        for _ in range(B3):
            t8 = []
            for n in (c3["Pm"], c3["Pk"], c3["Pg"]):
                t8 += [torch.randint(0, HW3, (n,), generator=gen, dtype=torch.int64) for _ in range(2)]
            t8 += [torch.tensor([-1], dtype=torch.int64)] * 2
            lists3.append(tuple(t8))
        pl3 = PairLists.from_lists(lists3, dev, hw=HW3)
        gg = torch.Generator(device=dev).manual_seed(4)
        da3 = ((torch.rand(B3, HW3, D3, device=dev, generator=gg) * 2 - 1) * 0.12).requires_grad_(True)
        db3 = ((torch.rand(B3, HW3, D3, device=dev, generator=gg) * 2 - 1) * 0.12).requires_grad_(True)
        pcl3 = type(pcl)(image_shape=[c3["H"], c3["W"]], config=LOSS_CONFIG)
        for _ in range(2):
            loss_fwd_bwd(da3, db3, pl3, pcl3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps3 = 5
        e0.record()
        for _ in range(reps3):
            loss_fwd_bwd(da3, db3, pl3, pcl3)
        e1.record()
        torch.cuda.synchronize()
        ms3 = e0.elapsed_time(e1) / reps3
        pb3, fb3 = (16 * D3 + 16) * pl3.total, 2 * B3 * HW3 * D3 * 4
        loss_roof["at_config3_list_sizes"] = {
            "pixel_pairs": pl3.total, "descriptor_dim": D3, "us_per_call": 1e3 * ms3,
            "algorithmic_bytes": {"pairs": pb3, "zero_fill": fb3},
            "achieved": (pb3 + fb3) / (ms3 * 1e-3) / 1e9, "frac": (pb3 + fb3) / (ms3 * 1e-3) / 1e9 / 8000.0,
            "achieved_pairs_only": pb3 / (ms3 * 1e-3) / 1e9, "frac_pairs_only": pb3 / (ms3 * 1e-3) / 1e9 / 8000.0}
        del da3, db3, pl3
        torch.cuda.empty_cache()

    # ---- measurements that ride on the same line (driver-timed): the same-precision fp32-MFMA arithmetic, BASELINE
    # configs[3]'s per-GPU share on ONE GPU (the 1-GPU point of the weak-scaling curve the multi-GPU runs trace), and for
    # N > 1 the fixed-global-batch-64 strong-scaling point
    variants = {}
    short = max(5, args.steps // 2) if not args.dry_run_cpu else 2

    def summarize(job_, sec, steps_, extra=None):
        d = {"value": 2 * job_.B * world * steps_ / sec, "unit": "images/s", "ms_per_step": 1e3 * sec / steps_, "steps": steps_,
             "pairs_per_gpu": job_.B, "global_pairs": job_.B * world, "host_enqueue_ms_per_step": getattr(job_, "host_enqueue_ms", None)}
        d.update(extra or {})
        return d

    if not args.no_variants and graph is None:
        if world == 1 and args.conv_mode == "f16x3":
            bb.set_conv_mode("fp32")
            sec, _ = job.timed(2, short, it_next + 16, use_dist)
            r32 = measure_roofline(job, "fp32", it_next + 64) if args.profile_steps > 0 else None
            variants["fp32_mfma"] = summarize(job, sec, short, {
                "arithmetic": "fp32 MFMA (v_mfma_f32_32x32x2_f32): the same-precision comparison point of the default",
                "roofline": None if r32 is None else {k: r32[k] for k in ("kernel", "achieved", "peak", "frac", "kernel_ms_per_step")}})
            bb.set_conv_mode(args.conv_mode)
        if world == 1 and args.workload == "config2" and not args.batch:
            # every other single-GPU BASELINE config on the same driver-timed line: configs[3]'s per-GPU share (the N = 1 point of
            # the weak-scaling curve the multi-GPU runs trace), configs[2] (B = 32, D = 16) and configs[4]'s per-GPU share
            # (ResNet50-8s 1280 x 960), each with the roofline of its gather-GEMM launches
            # (key, workload, warm-up, steps, two forward calls instead of forward_pair, step breakdown)
            for key, name, w_, k_, sep_, brk_ in (
                    ("separate_forwards", "config2", 4, short, True, False),      # the headline workload, LITERAL training.py:329-333 call pattern
                    ("config1_one_gpu", "config1", 8, 20, False, True),           # B = 1: the reference's own batch size (training.yaml:14)
                    ("config1_separate_forwards", "config1", 8, 20, True, True),
                    ("config4_one_gpu", "config4", 6, short, False, False), ("config3_one_gpu", "config3", 3, 5, False, False),
                    ("config5_one_gpu", "config5", 6, 8, False, False)):
                if sep_ and args.separate_forwards:
                    continue   # (the headline itself already runs the two-call pattern)
                wlv = dict(WORKLOADS[name])
                jobv = Job(args, wlv, wlv["B"], dev, rank, use_dist, separate_forwards=sep_ or args.separate_forwards)
                sec, _ = jobv.timed(w_, k_, 0, use_dist)
                rv = measure_roofline(jobv, args.conv_mode, w_ + k_ + 8) if args.profile_steps > 0 else None
                variants[key] = summarize(jobv, sec, k_, {
                    "workload": wlv["desc"],
                    "forward_calls": "forward(img_a), forward(img_b) -- training.py:329-333 as written" if jobv.separate else "forward_pair(img_a, img_b)",
                    "roofline": None if rv is None else {k: rv[k] for k in ("achieved", "peak", "frac", "kernel_ms_per_step", "hl_kernel", "conv_wgrad")}})
                if brk_ and rv is not None:
                    variants[key]["breakdown"] = step_breakdown(jobv, jobv.last_profile, 1e3 * sec / k_)
                    ew = elementwise_roofline(jobv.last_profile)
                    variants[key]["roofline_elementwise"] = None if ew is None else {k: ew[k] for k in ("achieved", "peak", "frac", "kernel_ms_per_step")}
                del jobv
                torch.cuda.empty_cache()
            variants["config4_one_gpu"]["note"] = ("the N = 1 point of the weak-scaling curve: multi-GPU lines run this workload per "
                                                   "rank (bench.py --gpus 1 --workload config4 prints it as the headline)")
        if use_dist and not args.batch:
            # the same per-rank workload on ONE GPU of this very box (rank 0 alone, the other ranks parked at the barrier, no
            # collective): the denominator of the weak-scaling efficiency, so that the curve does not depend on another run
            # (--force-dist with one rank: what the RCCL path costs a step when there is nobody to talk to)
            job.comm = False
            saved_mode = grads.bucketed
            grads.bucketed = False
            dist.barrier()
            if rank == 0:
                # (rank 0 steps alone: its parameters, BN buffers AND optimizer state are put back afterwards, so that the
                # replicas -- Adam moments and step counts included -- are identical again for whatever is measured next)
                import copy
                snap_model = {k: v.clone() for k, v in dcn.state_dict().items()}
                snap_opt = copy.deepcopy(job.opt.state_dict())
                sec1, _ = job.timed(2, short, it_next + 128, False)
                dcn.load_state_dict(snap_model)
                job.opt.load_state_dict(snap_opt)
                del snap_model, snap_opt
                one = 2 * B * short / sec1
                variants["weak_scaling"] = {"one_gpu_same_box": one, "one_gpu_ms_per_step": 1e3 * sec1 / short,
                                            "efficiency": (2 * B * world * args.steps / elapsed) / (world * one),
                                            "note": "rank 0 alone on the per-rank workload (%s), same process, no all-reduce" % args.workload}
            dist.barrier()
            job.comm = True
            grads.bucketed = saved_mode
            __import__("dcn_hip.distributed", fromlist=["broadcast_module"]).broadcast_module(dcn)
        if world > 1 and 64 % world == 0 and args.workload in ("config4", "dryrun") and not args.batch:
            Bs = (4 if args.dry_run_cpu else 64) // world
            if Bs == B:
                variants["strong_scaling_global64"] = summarize(job, elapsed, args.steps, {"note": "same as the headline at this N"})
            else:
                jobs = Job(args, wl, Bs, dev, rank, use_dist)
                sec, _ = jobs.timed(2, short, 0, use_dist)
                variants["strong_scaling_global64"] = summarize(jobs, sec, short, {
                    "workload": "BASELINE configs[3]: global batch 64 image pairs split over %d GPUs" % world})
                del jobs
        torch.cuda.empty_cache()

    if rank == 0:
        images_per_step = 2 * B * world
        ms_per_step = 1e3 * elapsed / args.steps
        out = {"metric": "training images/sec, 640x480 D=3 ResNet34-8s" if args.workload in ("config1", "config2", "config4")
               else ("training images/sec (%s)" % args.workload if not args.dry_run_cpu else "DRY RUN -- not a measurement"),
               "value": images_per_step * args.steps / elapsed, "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "host_enqueue_ms_per_step": host_enqueue_ms,
               "host_enqueue_idle_gpu_ms_per_step": getattr(job, "host_enqueue_idle_ms", None),
               "host_enqueue_ms_per_rank": host_enqueue_per_rank, "host_usable_cpus": usable_cpus(),
               "dtype": "f32 (f16x3 products)" if args.conv_mode == "f16x3" else "f32",
               "data": "synthetic" if not args.dry_run_cpu else "DRY RUN on CPU (host-emulated kernels, gloo): NOT a measurement",
               "arithmetic": ("fp32 tensors and accumulation; convolution products as 3 fp16 MFMAs on exact hi/lo operand "
                              "splits (~22 mantissa bits per operand; parity with the fp32 reference at 1e-4, tests/test_gpu_parity.py)"
                              if args.conv_mode == "f16x3" else "fp32 MFMA"),
               "config": {"workload": wl["desc"], "pairs_per_gpu": B, "images_per_step": images_per_step,
                          "image": "%dx%d" % (W, H), "descriptor_dim": D, "backbone": wl["backbone"],
                          "pixel_pairs_per_image_pair": [wl["Pm"], wl["Pk"], wl["Pg"]],
                          "conv_mode": args.conv_mode, "hip_graph": graph_note,
                          "backward_schedule": ("serial (DCN_BACKWARD_OVERLAP=0)" if os.environ.get("DCN_BACKWARD_OVERLAP") == "0" or args.conv_mode != "f16x3" or graph is not None
                                                else "weight-gradient GEMMs on the engine's side stream next to dgrad / BN backward; the roofline "
                                                     "steps time every launch alone (serial schedule), see profiles/r2_kernel_stats*.txt"),
                          "forward_calls": "forward(img_a), forward(img_b)" if args.separate_forwards else
                          "forward_pair(img_a, img_b): both network calls of the step as one grouped launch sequence, batch-norm "
                          "statistics / running statistics / gradients per image batch (identical results)", "optimizer": "Adam lr 1e-4 wd 1e-4 (%s)" % ("torch.optim.Adam" if args.torch_adam else "dcn_adam_step, one pass"), "parallelism": "dp%d" % world,
                          "library": info["version"], "final_loss": final_loss,
                          "weak_scaling_reference": ("this line IS the per-rank workload of the multi-GPU runs" if args.workload == "config4"
                                                     else "the multi-GPU lines run config4 (B = 8 pairs) per rank: their N = 1 point is "
                                                          "variants.config4_one_gpu, not `value`") if world == 1 else
                          "variants.weak_scaling.one_gpu_same_box (rank 0 alone on the same per-rank workload)",
                          "train_gflop_per_image": 3 * bb.get_plan("Resnet18_8s" if args.dry_run_cpu else wl["backbone"],
                                                                   wl.get("base_width", 64), B, H, W, D).forward_flops / B / 1e9},
               "roofline": roofline, "roofline_elementwise": roofline_elementwise, "breakdown": headline_breakdown,
               "roofline_loss_gather": loss_roof, "variants": variants,
               "allreduce_ms": comm["allreduce_ms"] if comm else None, "communication": comm}
        if world == 1 and args.cpu_baseline_steps > 0:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_baseline_steps, 1)
        else:
            out["cpu_baseline"] = None
        # ---- the comparisons that matter, flat and LAST on the line (a reader that keeps the contract keys and the tail of a
        # 14 KB line sees them; `variants` above has the detail): scalars only, short keys
        def pick(d, *path):
            for k in path:
                if not isinstance(d, dict) or d.get(k) is None:
                    return None
                d = d[k]
            return round(d, 4) if isinstance(d, float) else d
        summary = {"images_per_s": round(out["value"], 2), "ms_per_step": round(ms_per_step, 3),
                   "frac_gemm_fwd_dgrad": pick(roofline, "frac"), "frac_hl_kernel": pick(roofline, "hl_kernel", "frac"),
                   "frac_wgrad": pick(roofline, "conv_wgrad", "frac"), "wgrad_launches_per_step": pick(roofline, "conv_wgrad", "launches_per_step"),
                   "frac_elementwise_hbm": pick(roofline_elementwise, "frac"),
                   "elementwise_ms": pick(roofline_elementwise, "kernel_ms_per_step"),
                   "bn_finalize_launches": pick(roofline_elementwise, "bn_finalize", "launches_per_step"),
                   "bn_finalize_ms": pick(roofline_elementwise, "bn_finalize", "kernel_ms_per_step"),
                   "frac_loss_gather_hbm": pick(loss_roof, "frac"),
                   "frac_loss_gather_pairs_config3_sizes": pick(loss_roof, "at_config3_list_sizes", "frac_pairs_only"),
                   "engine_launches_per_step": pick(headline_breakdown, "engine_launches_per_step"),
                   "host_enqueue_idle_gpu_ms": pick(out, "host_enqueue_idle_gpu_ms_per_step")}
        for key, short_name in (("fp32_mfma", "exact_fp32"), ("separate_forwards", "separate_forwards"), ("config1_one_gpu", "config1_pair"),
                                ("config1_separate_forwards", "config1_separate_forwards"), ("config3_one_gpu", "config3"),
                                ("config4_one_gpu", "config4"), ("config5_one_gpu", "config5")):
            v = variants.get(key)
            if v:
                summary[short_name] = {"images_per_s": round(v["value"], 2), "ms_per_step": round(v["ms_per_step"], 3),
                                       "frac": pick(v, "roofline", "frac"),
                                       "host_idle_ms": pick(v, "breakdown", "host_enqueue_idle_gpu_ms_per_step")}
        if comm:
            summary["communication"] = {k: pick(comm, k) for k in ("allreduce_ms", "allreduce_exposed_ms", "bucketed_steps",
                                                                   "bucket_collectives", "monolithic_steps", "ranks")}
        # the driver's parser keeps the scalars of `roofline` and `config`: the two numbers asked for most often ride there too
        if roofline is not None:
            roofline["frac_wgrad"], roofline["frac_elementwise_hbm"] = summary["frac_wgrad"], summary["frac_elementwise_hbm"]
        out["config"]["exact_fp32_images_per_s"] = pick(summary, "exact_fp32", "images_per_s")
        out["config"]["separate_forwards_images_per_s"] = pick(summary, "separate_forwards", "images_per_s")
        out["summary"] = summary
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

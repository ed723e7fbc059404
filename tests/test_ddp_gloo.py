"""Multi-GPU path on CPU: gloo, world size 2 and 4.  Every rank runs the step on its own image pairs (kernels
host-emulated), FlatGradients averages -- bucket by bucket as the backbone's backward completes them (the default with
world size > 1), or with one all-reduce; the result must equal the oracle run per shard with averaged gradients
(BN statistics are per rank, no SyncBN), and the two schedules must agree."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, pair=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import use_emulation_library
    use_emulation_library()
    from dcn_hip.distributed import FlatGradients, broadcast_module
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import resnet_dilated_oracle as orc, step as ostep, synth
    from pytorch_segmentation_detection.models import resnet_dilated as prod
    H, W, D, B = (64, 64, 3, 1) if pair else (32, 32, 3, 1)   # (a group's rows must be tile-aligned for the grouped plan)
    torch.manual_seed(100 + rank)  # different init per rank: the broadcast must fix that
    m = prod.Resnet18_8s(num_classes=D, base_width=8)
    if rank == 0:
        m.load_state_dict(orc.build("Resnet18_8s", D, seed=0, base_width=8).state_dict())
    broadcast_module(m, src=0)
    grads = FlatGradients(m)
    m.train()
    img_a, img_b, lists = synth.make_batch(B, H, W, 40, 20, 20, seed=1 + rank)
    pcl = PixelwiseContrastiveLoss([H, W], synth.LOSS_CONFIG)
    grads.zero_()
    ya, yb = m.forward_pair(img_a, img_b) if pair else (m(img_a), m(img_b))   # bench.py's default is the grouped call
    if pair:
        from dcn_hip import backbone as _bb
        assert any(k[-1] == 2 for k in _bb._PLANS), "forward_pair fell back to two calls"
    pa = ya.permute(0, 2, 3, 1).reshape(B, H * W, D)
    pb = yb.permute(0, 2, 3, 1).reshape(B, H * W, D)
    L = lists[0]
    tup = [(L["matches_a"], L["matches_b"], L["masked_non_matches_a"], L["masked_non_matches_b"],
            L["background_non_matches_a"], L["background_non_matches_b"], L["blind_non_matches_a"],
            L["blind_non_matches_b"])]
    loss, _, _ = loss_composer.get_loss_batched(pcl, 0, pa, pb, tup)
    loss.backward()
    grads.all_reduce_mean()
    assert grads.stats["bucketed_steps"] == (1 if pair else 2) and grads.stats["monolithic_steps"] == 0   # world 2: bucketed
    # oracle: both shards on this rank, averaged
    ref = None
    for r in range(world):
        o = orc.build("Resnet18_8s", D, seed=0, base_width=8)
        o.train()
        ia, ib, ls = synth.make_batch(B, H, W, 40, 20, 20, seed=1 + r)
        lo = ostep.forward_loss(o, ia, ib, ls, synth.LOSS_CONFIG)[0]
        lo.backward()
        g = torch.cat([p.grad.reshape(-1) for p in o.parameters()])
        ref = g if ref is None else ref + g
    ref = ref / world
    mine = torch.cat([p.grad.contiguous().reshape(-1) for p in m.parameters()])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    q.put((rank, rel_err(mine, ref), float((gathered[0] - gathered[1]).abs().max()),
           all(p.grad.data_ptr() >= grads.flat.data_ptr() for p in m.parameters())))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("pair", [False, True], ids=["two_forward_calls", "forward_pair"])
def test_two_rank_gradient_average_matches_oracle(pair):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, pair)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=540) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, spread, views in res:
        assert err < 5e-4, (rank, err)
        assert spread == 0.0, "ranks disagree after the all-reduce"
        assert views


def _bucket_worker(rank, world, port, q):
    """Same step twice on every rank -- monolithic all-reduce, then the bucketed schedule -- and the exact-arithmetic check of
    the slicing: integer-valued "engine gradients" whose sums are exact in fp32 whatever order gloo adds them in."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="1")   # 1: deterministic atomics order
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import use_emulation_library
    use_emulation_library()
    from dcn_hip import backbone as bb
    from dcn_hip.distributed import FlatGradients, broadcast_module
    from oracle import resnet_dilated_oracle as orc, synth
    from pytorch_segmentation_detection.models import resnet_dilated as prod
    H, W, D = 64, 64, 3
    torch.manual_seed(7)
    m = prod.Resnet18_8s(num_classes=D, base_width=8)
    m.load_state_dict(orc.build("Resnet18_8s", D, seed=0, base_width=8).state_dict())
    broadcast_module(m, src=0)
    m.train()
    img_a, img_b, _ = synth.make_batch(1, H, W, 4, 4, 4, seed=1 + rank)
    g = torch.Generator().manual_seed(50 + rank)
    ga, gb = torch.randn(1, D, H, W, generator=g), torch.randn(1, D, H, W, generator=g)
    flats = {}
    for mode in (False, True):
        grads = FlatGradients(m, bucketed=mode)
        grads.zero_()
        ya, yb = m.forward_pair(img_a, img_b)
        torch.autograd.backward([ya, yb], [ga, gb])
        grads.all_reduce_mean()
        assert grads.stats["bucketed_steps"] == int(mode) and grads.stats["monolithic_steps"] == int(not mode)
        flats[mode] = grads.flat.clone()
    same_bits = bool(torch.equal(flats[False], flats[True]))
    close = float((flats[False] - flats[True]).abs().max() / flats[False].abs().max())
    # exact-arithmetic check of the bucket slicing / coverage
    plan = bb.get_plan("Resnet18_8s", 8, 2, H, W, D, 2)
    n = plan.grad_offsets[-1]
    eng = ((torch.arange(n) % 251) * world * (rank + 1)).float()        # multiples of `world`: the pre-division is exact
    grads = FlatGradients(m, bucketed=True)
    grads.zero_()
    grads.accumulate_and_reduce_buckets(plan, eng.clone())
    grads.all_reduce_mean()
    expect = (torch.arange(n) % 251).float() * sum(r + 1 for r in range(world))
    mono = FlatGradients(m, bucketed=False)
    mono.zero_()
    mono.flat.copy_(eng)
    mono.all_reduce_mean()
    exact = bool(torch.equal(grads.flat, expect)) and bool(torch.equal(mono.flat, expect))
    covered = sorted(plan.grad_buckets) == sorted(set(plan.grad_buckets)) and \
        sum(hi - lo for lo, hi in plan.grad_buckets) == n
    q.put((rank, same_bits, close, exact, covered))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_all_reduce_equals_monolithic(world):
    """world 2: bit for bit on real gradients (a + b is commutative, so gloo's chunking cannot matter); world 4: within fp32
    summation-order noise on real gradients, and bit for bit on integer-valued buffers whose sums are exact."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=540) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same_bits, close, exact, covered in res:
        assert covered and exact, rank
        assert close < 1e-6, (rank, close)
        if world == 2:
            assert same_bits, rank

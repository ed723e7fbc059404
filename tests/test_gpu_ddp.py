"""Data-parallel schedule on REAL streams: two and four ranks sharing the one MI355X of the test box (backend gloo on device tensors --
RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's), so that the part of the bucketed all-reduce that
the CPU / gloo tests cannot see is exercised on hardware: the engine's grad-ready events, the communication stream, the
cross-stream lifetime of the engine's gradient buffer, the join before the optimizer step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from helpers import use_gfx950_library
    use_gfx950_library()
    import parity_common as pc
    from dcn_hip import backbone as bb
    from dcn_hip.distributed import FlatGradients, broadcast_module
    from dcn_hip.optim import Adam
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import synth
    bb.set_conv_mode("f16x3")
    H, W, D, B = 192, 256, 3, 2
    torch.manual_seed(100 + rank)                      # different initial weights per rank: the broadcast must fix that
    dcn, _ = pc.build_dcn("Resnet34_8s", D, H, W)
    if rank != 0:
        with torch.no_grad():
            for p in dcn.parameters():
                p.add_(0.01)
    broadcast_module(dcn, src=0)
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    img_a, img_b, lists = synth.make_batch(B, H, W, 400, 200, 200, seed=1 + rank)
    img_a, img_b = img_a.cuda(), img_b.cuda()
    tup = [tuple(Ld[k].cuda() for k in pc.KEYS) for Ld in lists]
    gen = torch.Generator(device="cuda").manual_seed(7 + rank)
    ga = torch.randn(B, D, H, W, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    gb = torch.randn(B, D, H, W, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)

    def grads_of(mode, fixed, pair):
        grads = FlatGradients(dcn, bucketed=mode)
        grads.zero_()
        ya, yb = dcn.forward_pair(img_a, img_b) if pair else (dcn.forward(img_a), dcn.forward(img_b))
        if fixed:                                      # (no loss atomics: bit-reproducible)
            torch.autograd.backward([ya, yb], [ga, gb])
        else:
            loss_composer.get_loss_batched(pcl, 0, dcn.process_network_output(ya, B), dcn.process_network_output(yb, B),
                                           tup)[0].backward()
        # keep the compute stream busy behind backward: the join in all_reduce_mean has to order the optimizer after the
        # communication stream, not merely after the host
        grads.all_reduce_mean()
        out = grads.flat.clone()
        torch.cuda.synchronize()
        return out, dict(grads.stats)

    res = {}
    mono, st0 = grads_of(False, True, True)
    buck, st1 = grads_of(None, True, True)             # None: decided per step -> bucketed, because world size is 2
    res["bitwise_pair"] = bool(torch.equal(mono, buck))
    res["stats"] = (st0["bucketed_steps"], st0["monolithic_steps"], st1["bucketed_steps"], st1["monolithic_steps"])
    mono2, _ = grads_of(False, True, False)            # two forward calls -> two backward calls -> buckets reduced twice
    buck2, st2 = grads_of(None, True, False)
    # (two backward calls: (ga + gb)/2 summed over the ranks vs gb/2 summed + ga/2 summed -- another association, not the same bits)
    res["two_calls_rel"] = float((mono2 - buck2).abs().max() / mono2.abs().max())
    res["two_calls_buckets"] = st2["bucketed_steps"]
    # ranks agree, and the average really is the average of the two ranks' own gradients
    gathered = [torch.zeros_like(buck) for _ in range(world)]
    dist.all_gather(gathered, buck)
    res["ranks_agree"] = all(bool(torch.equal(gathered[0], t)) for t in gathered[1:])
    res["bucket_vs_mono_rel"] = float((mono - buck).abs().max() / mono.abs().max())
    dist.barrier()
    # a full step with the real loss + Adam on both schedules from identical parameters: parameters must agree to round-off
    state = {k: v.clone() for k, v in dcn.state_dict().items()}
    outs = []
    for mode in (False, None):
        dcn.load_state_dict(state)
        grads = FlatGradients(dcn, bucketed=mode)
        opt = grads.attach(Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4))
        for _ in range(2):
            opt.zero_grad()
            ya, yb = dcn.forward_pair(img_a, img_b)
            loss_composer.get_loss_batched(pcl, 0, dcn.process_network_output(ya, B), dcn.process_network_output(yb, B),
                                           tup)[0].backward()
            grads.all_reduce_mean()
            opt.step()
        torch.cuda.synchronize()
        outs.append(torch.cat([p.detach().reshape(-1) for p in dcn.parameters()]).clone())
    res["step_max_diff"] = float((outs[0] - outs[1]).abs().max())
    allp = [torch.zeros_like(outs[1]) for _ in range(world)]
    dist.all_gather(allp, outs[1])
    res["replicas_in_sync"] = all(bool(torch.equal(allp[0], t)) for t in allp[1:])
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_bucketed_overlap_on_device(world):
    """world 2 and 4 processes on the one MI355X of the box: the bucketed schedule on real streams (grad-ready events,
    communication stream, buffer lifetime, join) at the world sizes the driver's scaling runs use."""
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1000) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, r in res:
        assert r["stats"] == (0, 1, 1, 0), (rank, r)
        if world == 2:
            assert r["bitwise_pair"], (rank, r)        # world 2: a + b is commutative, so the slicing cannot change a bit
        else:                                          # world 4: gloo may associate the four summands differently per message size
            assert r["bucket_vs_mono_rel"] < 1e-6, (rank, r)
        assert r["two_calls_rel"] < 1e-6 and r["two_calls_buckets"] == 2, (rank, r)
        assert r["ranks_agree"] and r["replicas_in_sync"], (rank, r)
        # two Adam steps on real (atomics-ordered) gradients: first steps are lr * sign(g), so a flipped near-zero gradient moves
        # a parameter by up to 2 * lr per step
        assert r["step_max_diff"] <= 4.5e-4, (rank, r)

#!/usr/bin/env python3
"""Generate loss golden vectors by executing the REFERENCE'S OWN source text.

Runs only in the authoring container (needs /root/reference); the GPU box never runs it.  The
reference files are Python 2 / torch 1.1, so their text is read from /root/reference, patched IN
MEMORY with the minimal py2->py3 edits listed in ``PATCHES`` below (each one preserves the py2 /
torch-1.1 meaning), exec'd, and driven on seeded inputs.  No reference source is copied into this
repository -- only the numeric outputs are stored (tests/golden/loss_ref_*.npz).

    python tests/golden/make_loss_goldens_from_reference.py
"""
import os
import re
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

PCL_FILE = os.path.join(REF, "dense_correspondence/loss_functions/pixelwise_contrastive_loss.py")
COMPOSER_FILE = os.path.join(REF, "dense_correspondence/loss_functions/loss_composer.py")

# (file tag, regex, replacement, why)
PATCHES = [
    ("pcl", r"long\(", "int(", "py2 builtin long"),
    ("pcl", r"num_non_matches / num_matches", "num_non_matches // num_matches", "py2 int division (pcl.py:113)"),
    ("pcl", r"len\(non_matches_b\)/len\(matches_b\)", "len(non_matches_b)//len(matches_b)", "py2 int division (pcl.py:321)"),
    ("pcl", r"u_v_pixel_locations\[:,1\]/self\.image_width", "u_v_pixel_locations[:,1]//self.image_width",
     "torch-1.1 LongTensor '/' is integer division (pcl.py:351)"),
    ("composer", r'print "([^"]*)"', r'print("\1")', "py2 print statement"),
    ("composer", r"\.cuda\(\)", "", "no GPU in the authoring container (loss_composer.py:215)"),
]


def load_reference_modules():
    # stub for the dataset module the composer imports (loss_composer.py:1); the two names it uses are
    # restated from spartan_dataset_masked.py:31-36 and dense_correspondence_dataset_masked.py:218-223
    class SpartanDatasetDataType:
        SINGLE_OBJECT_WITHIN_SCENE = 0
        SINGLE_OBJECT_ACROSS_SCENE = 1
        DIFFERENT_OBJECT = 2
        MULTI_OBJECT = 3
        SYNTHETIC_MULTI_OBJECT = 4

    class SpartanDataset:
        @staticmethod
        def is_empty(tensor):
            return (len(tensor) == 1) and (tensor[0] == -1)

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    mod("dense_correspondence")
    mod("dense_correspondence.dataset")
    ds = mod("dense_correspondence.dataset.spartan_dataset_masked")
    ds.SpartanDataset = SpartanDataset
    ds.SpartanDatasetDataType = SpartanDatasetDataType
    mod("dense_correspondence.loss_functions")

    def patched(path, tag):
        src = open(path).read().expandtabs(8)
        for t, pat, rep, _why in PATCHES:
            if t == tag:
                src, n = re.subn(pat, rep, src)
                assert n > 0, (pat, "did not match -- reference changed?")
        return src

    pcl_mod = mod("dense_correspondence.loss_functions.pixelwise_contrastive_loss")
    exec(compile(patched(PCL_FILE, "pcl"), PCL_FILE, "exec"), pcl_mod.__dict__)
    comp_mod = mod("dense_correspondence.loss_functions.loss_composer")
    exec(compile(patched(COMPOSER_FILE, "composer"), COMPOSER_FILE, "exec"), comp_mod.__dict__)
    return pcl_mod, comp_mod, SpartanDatasetDataType


BASE_CFG = {  # config/dense_correspondence/training/training.yaml:51-61
    "M_masked": 0.5, "M_background": 0.5, "M_pixel": 50, "match_loss_weight": 1.0, "non_match_loss_weight": 1.0,
    "use_l2_pixel_loss_on_masked_non_matches": False, "use_l2_pixel_loss_on_background_non_matches": False,
    "scale_by_hard_negatives": True, "scale_by_hard_negatives_DIFFERENT_OBJECT": True, "alpha_triplet": 0.1,
}

CASES = [
    # name, H, W, D, Pm, Pk, Pg, Pblind, descriptor scale, cfg overrides, match_type
    dict(name="within_d3", H=48, W=64, D=3, Pm=200, Pk=100, Pg=100, Pb=0, scale=0.35, cfg={}, mt=0),
    dict(name="within_d16", H=48, W=64, D=16, Pm=150, Pk=300, Pg=300, Pb=0, scale=0.12, cfg={}, mt=0),
    dict(name="within_blind", H=48, W=64, D=3, Pm=64, Pk=64, Pg=64, Pb=50, scale=0.35, cfg={}, mt=0),
    dict(name="within_single", H=48, W=64, D=3, Pm=1, Pk=1, Pg=1, Pb=0, scale=0.2, cfg={}, mt=0),
    dict(name="within_noscale", H=48, W=64, D=3, Pm=100, Pk=80, Pg=120, Pb=0, scale=0.35,
         cfg={"scale_by_hard_negatives": False}, mt=0),
    dict(name="within_margins", H=48, W=64, D=4, Pm=100, Pk=128, Pg=64, Pb=0, scale=0.5,
         cfg={"M_masked": 0.7, "M_background": 0.3, "match_loss_weight": 0.5, "non_match_loss_weight": 2.0}, mt=0),
    dict(name="within_nohard", H=48, W=64, D=3, Pm=50, Pk=40, Pg=40, Pb=0, scale=30.0, cfg={}, mt=0),
    dict(name="within_pixel_l2", H=48, W=64, D=3, Pm=40, Pk=120, Pg=80, Pb=0, scale=0.35,
         cfg={"use_l2_pixel_loss_on_masked_non_matches": True, "use_l2_pixel_loss_on_background_non_matches": True,
              "M_pixel": 25}, mt=0),
    dict(name="multi_object", H=48, W=64, D=3, Pm=60, Pk=60, Pg=60, Pb=0, scale=0.35, cfg={}, mt=3),
    dict(name="different_object", H=48, W=64, D=3, Pm=0, Pk=0, Pg=0, Pb=300, scale=0.35, cfg={}, mt=2),
    dict(name="different_object_noscale", H=48, W=64, D=3, Pm=0, Pk=0, Pg=0, Pb=300, scale=0.35,
         cfg={"scale_by_hard_negatives_DIFFERENT_OBJECT": False}, mt=2),
]


def make_lists(HW, n, gen):
    if n == 0:
        return torch.tensor([-1]), torch.tensor([-1])
    return (torch.randint(0, HW, (n,), generator=gen, dtype=torch.int64),
            torch.randint(0, HW, (n,), generator=gen, dtype=torch.int64))


def main():
    pcl_mod, comp_mod, T = load_reference_modules()
    for ci, case in enumerate(CASES):
        gen = torch.Generator().manual_seed(100 + ci)
        HW = case["H"] * case["W"]
        cfg = dict(BASE_CFG)
        cfg.update(case["cfg"])
        A = ((torch.rand(1, HW, case["D"], generator=gen) * 2 - 1) * case["scale"]).requires_grad_(True)
        B = ((torch.rand(1, HW, case["D"], generator=gen) * 2 - 1) * case["scale"]).requires_grad_(True)
        ma, mb = make_lists(HW, case["Pm"], gen)
        ka, kb = make_lists(HW, case["Pk"], gen)
        ga, gb = make_lists(HW, case["Pg"], gen)
        ba, bb = make_lists(HW, case["Pb"], gen)
        if cfg["use_l2_pixel_loss_on_masked_non_matches"]:
            # layout the reference requires (pcl.py:323): non-matches grouped per match
            ka = ma.repeat_interleave(case["Pk"] // case["Pm"])
            ga = ma.repeat_interleave(case["Pg"] // case["Pm"])
        pcl = pcl_mod.PixelwiseContrastiveLoss(image_shape=[case["H"], case["W"]], config=cfg)
        out = comp_mod.get_loss(pcl, torch.tensor([case["mt"]]), A, B, ma, mb, ka, kb, ga, gb, ba, bb)
        loss = out[0]
        if loss.requires_grad:
            loss.sum().backward()
        gradA = A.grad if A.grad is not None else torch.zeros_like(A)
        gradB = B.grad if B.grad is not None else torch.zeros_like(B)
        extra = {}
        if case["mt"] == 0 and case["Pm"] > 1 and not cfg["use_l2_pixel_loss_on_masked_non_matches"]:
            # the building blocks, called the way the composer calls them (F6-F8)
            with torch.no_grad():
                ml, _, _ = pcl_mod.PixelwiseContrastiveLoss.match_loss(A, B, ma, mb)
                vec, hn, _, _ = pcl_mod.PixelwiseContrastiveLoss.non_match_descriptor_loss(A, B, ka, kb, M=cfg["M_masked"])
                vec_inv, hn_inv, _, _ = pcl_mod.PixelwiseContrastiveLoss.non_match_descriptor_loss(
                    A, B, ka, kb, M=cfg["M_masked"], invert=True)
                orig = pcl.get_loss_original(A, B, ma, mb, ka, kb)
            extra = dict(f6_match_loss=ml.numpy(), f7_vec=vec.numpy(), f7_hard=np.int64(hn),
                         f7_vec_invert=vec_inv.numpy(), f7_hard_invert=np.int64(hn_inv),
                         original_loss=np.array([o.item() for o in orig], np.float32))
            if case["Pk"] % case["Pm"] == 0:
                with torch.no_grad():
                    ka_t = ma.repeat_interleave(case["Pk"] // case["Pm"])
                    trip = pcl_mod.PixelwiseContrastiveLoss.get_triplet_loss(A, B, ma, mb, ka_t, kb, cfg["alpha_triplet"])
                extra["triplet"] = trip.numpy()
                extra["triplet_non_matches_a"] = ka_t.numpy()
        path = os.path.join(HERE, "loss_ref_%s.npz" % case["name"])
        np.savez_compressed(
            path, A=A.detach().numpy(), B=B.detach().numpy(),
            matches_a=ma.numpy(), matches_b=mb.numpy(), masked_a=ka.numpy(), masked_b=kb.numpy(),
            background_a=ga.numpy(), background_b=gb.numpy(), blind_a=ba.numpy(), blind_b=bb.numpy(),
            H=case["H"], W=case["W"], match_type=case["mt"],
            cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([float(v) for v in cfg.values()]),
            out=np.array([float(o.sum().item()) for o in out], np.float64),
            gradA=gradA.numpy(), gradB=gradB.numpy(), **extra)
        print(case["name"], [float(o.sum().item()) for o in out], os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Test infrastructure: run the REFERENCE'S OWN training driver -- ``DenseCorrespondenceTraining.run()`` and
``run_from_pretrained()`` of /root/reference/dense_correspondence/training/training.py:46-601, imported from where it lies
(tests/reference_py3.py converts the Python-2 text in memory) -- for a few iterations on a tiny in-memory dataset, in one of
two configurations, and write what it logged and saved to an ``.npz``:

  --mode product     this repository's source root in FRONT of the reference's on sys.path (INTEGRATION.md's recipe):
                     training.py, the real SpartanDataset / evaluation modules come from the reference, the network wrapper,
                     backbone and loss from pytorch-dense-correspondence_amd/ (kernels: --library hostemu on a CPU box,
                     gfx950 on a GPU box);
  --mode reference   the reference alone: its own dense_correspondence_network.py / pixelwise_contrastive_loss.py /
                     loss_composer.py, with the un-vendored ``resnet_dilated`` module (network.py:16) supplied by the oracle
                     backbone.  This is what tests/golden/training_loop_ref.npz is made from.

Runs in its own process (the two configurations need different ``sys.modules``).  Needs /root/reference: the GPU box never
runs it -- tests/test_reference_training_loop.py replays the recorded batches there against the fixture."""
import argparse
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "pytorch-dense-correspondence_amd")

H, W, D = 32, 48, 3
BASE_WIDTH = 8
ARCH = "Resnet18_8s"
FC_SCALE = 4.0          # untrained descriptors are tiny: scale the scoring layer so that both sides of the hinge occur
NUM_SAMPLES = 4         # distinct image pairs; sample 2 of every round is the loader's "empty" sentinel (training.py:304-306)
PAIRS = (60, 30, 30)
FIRST_RUN_ITERATIONS = 6    # training.py:453 stops at `> num_iterations`: 7 iterations, the empty one included
RESUMED_ITERATIONS = 2


def make_samples():
    """The tiny dataset's content: NUM_SAMPLES seeded image pairs + index lists (oracle.synth) as the 12-tuples of
    spartan_dataset_masked.py:839; one more entry is the empty sentinel of dense_correspondence_dataset_masked.py:202-207."""
    import torch
    from oracle import synth
    img_a, img_b, lists = synth.make_batch(NUM_SAMPLES, H, W, *PAIRS, seed=1)
    keys = ("matches_a", "matches_b", "masked_non_matches_a", "masked_non_matches_b", "background_non_matches_a",
            "background_non_matches_b", "blind_non_matches_a", "blind_non_matches_b")
    out = []
    for i in range(NUM_SAMPLES):
        out.append((0, img_a[i].clone(), img_b[i].clone()) + tuple(lists[i][k].clone() for k in keys) + ({"type": 0},))
    empty = torch.LongTensor([-1])
    out.insert(2, (-1, img_a[0].clone(), img_b[0].clone()) + (empty,) * 8 + ({"type": -1},))
    return out


def initial_state_dict():
    import torch
    from oracle import resnet_dilated_oracle as orc
    o = orc.build(ARCH, D, seed=0, base_width=BASE_WIDTH)
    with torch.no_grad():
        getattr(o, ARCH.lower()).fc.weight.mul_(FC_SCALE)
    return o, o.state_dict()


def setup_imports(mode, library):
    sys.path.insert(0, HERE)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)            # `oracle` (this script is test infrastructure)
    import reference_py3 as rp
    assert rp.available(), "the reference tree is not mounted"
    rp.install()
    rp.install_third_party_stubs()
    os.environ["DC_SOURCE_DIR"] = rp.REF
    os.environ.setdefault("DCN_QUIET_SHIMS", "1")
    if mode == "product":
        sys.path.insert(0, PKG)             # this root in front of the reference's
        sys.path.extend([rp.REF, os.path.join(rp.REF, "modules")])
        from dcn_hip import _lib
        if library == "hostemu":
            sys.path.insert(0, os.path.join(HERE, "hostemu"))
            import build_emu
            _lib.load(build_emu.build())
            assert _lib.is_hostemu()
        else:
            _lib.load(_lib.DEFAULT_PATH)
            assert not _lib.is_hostemu()
        import pytorch_segmentation_detection.models.resnet_dilated as rd
        _, sd = initial_state_dict()

        class NarrowTestNet(getattr(rd, ARCH)):          # from_config passes only num_classes (network.py:375)
            def __init__(self, num_classes):
                super(NarrowTestNet, self).__init__(num_classes=num_classes, base_width=BASE_WIDTH)
                self.load_state_dict(sd)
        NarrowTestNet.arch, NarrowTestNet.attr = getattr(rd, ARCH).arch, getattr(rd, ARCH).attr
        rd.NarrowTestNet = NarrowTestNet
    else:
        import types
        sys.path.extend([rp.REF, os.path.join(rp.REF, "modules")])
        # the un-vendored submodule (.gitmodules:1-3): resnet_dilated = the oracle's restatement; fcn / transforms: names only
        psd = types.ModuleType("pytorch_segmentation_detection")
        psd.__path__ = []
        models = types.ModuleType("pytorch_segmentation_detection.models")
        models.__path__ = []
        rd = types.ModuleType("pytorch_segmentation_detection.models.resnet_dilated")
        fcn = types.ModuleType("pytorch_segmentation_detection.models.fcn")
        tr = types.ModuleType("pytorch_segmentation_detection.transforms")
        for n in ("ComposeJoint", "RandomHorizontalFlipJoint", "RandomScaleJoint", "CropOrPad", "ResizeAspectRatioPreserve",
                  "RandomCropJoint", "Split2D"):
            setattr(tr, n, type(n, (object,), {}))

        def NarrowTestNet(num_classes):
            assert num_classes == D
            return initial_state_dict()[0]
        rd.NarrowTestNet = NarrowTestNet
        psd.models, psd.transforms, models.resnet_dilated, models.fcn = models, tr, rd, fcn
        for m in (psd, models, rd, fcn, tr):
            sys.modules[m.__name__] = m
        # tensorboard_logger (training.py:20) is not installed: the same scalars.tsv recorder the product ships
        import importlib.util
        spec = importlib.util.spec_from_file_location("tensorboard_logger", os.path.join(PKG, "tensorboard_logger.py"))
        tbl = importlib.util.module_from_spec(spec)
        sys.modules["tensorboard_logger"] = tbl
        spec.loader.exec_module(tbl)


def neutralise_cuda_calls():
    """training.py:311-323, network.py:435 hard-code ``.cuda()``; on a box without a GPU they become the identity (the
    reference-mode run and the host-emulated product run are CPU runs)."""
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def run(mode, library, out_path, workdir):
    import numpy as np
    import torch
    setup_imports(mode, library)
    on_gpu = mode == "product" and library != "hostemu"
    if not on_gpu:
        neutralise_cuda_calls()
    import dense_correspondence.training.training as training_module
    import dense_correspondence_manipulation.utils.utils as utils
    from dense_correspondence.dataset.spartan_dataset_masked import SpartanDataset
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    assert training_module.__file__.startswith(os.environ["DC_SOURCE_DIR"]), training_module.__file__
    served = []

    class TinyDataset(SpartanDataset):
        """The real SpartanDataset class (so that its ``set_parameters_from_training_config`` runs on the reference's
        training.yaml) with the disk-reading parts replaced: samples come from memory, in call order -- like the real one,
        ``__getitem__`` ignores its index (dense_correspondence_dataset_masked.py:55-66)."""

        def __init__(self):                    # (no scene files)
            self.debug, self.mode, self._verbose = False, "train", False
            self._config = {"tiny_in_memory_dataset": True, "num_samples": NUM_SAMPLES + 1}
            self._samples = make_samples()

        def __len__(self):
            return len(self._samples)

        def load_all_pose_data(self):
            pass

        def __getitem__(self, index):
            k = len(served) % len(self._samples)
            served.append(k)
            return self._samples[k]

    os.environ["DC_DATA_DIR"] = workdir
    config = utils.getDictFromYamlFilename(os.path.join(os.environ["DC_SOURCE_DIR"], "config", "dense_correspondence",
                                                        "training", "training.yaml"))
    config["training"].update(num_iterations=FIRST_RUN_ITERATIONS, num_workers=0, save_rate=3, logging_rate=2,
                              steps_between_learning_rate_decay=2, logging_dir_name="first",
                              logging_dir=os.path.join(workdir, "trained_models"))
    # (py2's `/=` on the int array the yaml's `1` / `0` make is classic division; numpy under py3 refuses it: floats)
    config["training"]["data_type_probabilities"] = {k: float(v) for k, v in config["training"]["data_type_probabilities"].items()}
    config["dense_correspondence_network"].update(image_width=W, image_height=H, descriptor_dimension=D)
    config["dense_correspondence_network"]["backbone"]["resnet_name"] = "NarrowTestNet"

    torch.manual_seed(1)
    train = training_module.DenseCorrespondenceTraining(config=config, dataset=TinyDataset())
    train.run()
    first_dir = train.logging_dir
    log1 = {k: list(v) for k, v in train._logging_dict["train"].items()}
    files1 = sorted(os.listdir(first_dir))

    # resume (training.py:169-226): latest checkpoint + optimizer state of the first run, two more iterations
    import copy
    config2 = copy.deepcopy(config)
    config2["training"].update(num_iterations=RESUMED_ITERATIONS, logging_dir_name="resumed")
    train2 = training_module.DenseCorrespondenceTraining(config=config2, dataset=TinyDataset())
    train2.run_from_pretrained(first_dir)
    log2 = {k: list(v) for k, v in train2._logging_dict["train"].items()}
    files2 = sorted(os.listdir(train2.logging_dir))

    def scalars(d):
        rows = {}
        for line in open(os.path.join(d, "tensorboard", "scalars.tsv")):
            step, name, value = line.rstrip("\n").split("\t")
            rows.setdefault(name, []).append((int(step), float(value)))
        return rows
    s1, s2 = scalars(first_dir), scalars(train2.logging_dir)
    key = "train loss SINGLE_OBJECT_WITHIN_SCENE"

    def checkpoint_summary(path):
        sd = torch.load(path, map_location="cpu")
        names = sorted(sd.keys())
        norms = np.array([float(sd[k].double().norm()) for k in names])
        heads = np.array([sd[k].double().reshape(-1)[:4].tolist() + [0.0] * max(0, 4 - sd[k].numel()) for k in names])
        return names, norms, heads
    last1 = [f for f in files1 if f.endswith(".pth")][-1]
    last2 = [f for f in files2 if f.endswith(".pth")][-1]
    names, norms0, heads0 = checkpoint_summary(os.path.join(first_dir, "000000.pth"))
    _, norms1, heads1 = checkpoint_summary(os.path.join(first_dir, last1))
    _, norms2, heads2 = checkpoint_summary(os.path.join(train2.logging_dir, last2))

    # the model folder the reference's loop wrote loads back through the wrapper (network.py:441-485)
    reloaded = DenseCorrespondenceNetwork.from_model_folder(train2.logging_dir)
    sd_saved = torch.load(os.path.join(train2.logging_dir, last2), map_location="cpu")
    sd_now = reloaded.state_dict()
    assert list(sd_now.keys()) == list(sd_saved.keys())
    assert all(torch.equal(sd_now[k].cpu(), sd_saved[k]) for k in sd_saved)
    opt_sd = torch.load(os.path.join(train2.logging_dir, last2 + ".opt"), map_location="cpu")
    assert set(opt_sd.keys()) == {"state", "param_groups"} and len(opt_sd["state"]) == len(list(reloaded.parameters()))

    np.savez_compressed(
        out_path,
        mode=mode, library=str(library), training_file=training_module.__file__,
        network_file=sys.modules["dense_correspondence.network.dense_correspondence_network"].__file__,
        loss_file=sys.modules["dense_correspondence.loss_functions.loss_composer"].__file__,
        dataset_file=sys.modules["dense_correspondence.dataset.spartan_dataset_masked"].__file__,
        served=np.array(served),
        loss_steps=np.array([s for s, _ in s1[key]] + [s for s, _ in s2[key]]),
        loss=np.array([v for _, v in s1[key]] + [v for _, v in s2[key]]),
        match_loss=np.array(log1["match_loss"] + log2["match_loss"]),
        masked_non_match_loss=np.array(log1["masked_non_match_loss"] + log2["masked_non_match_loss"]),
        background_non_match_loss=np.array(log1["background_non_match_loss"] + log2["background_non_match_loss"]),
        learning_rate=np.array(log1["learning_rate"] + log2["learning_rate"]),
        files_first=np.array(files1), files_resumed=np.array(files2),
        param_names=np.array(names), norms_initial=norms0, norms_first=norms1, norms_resumed=norms2,
        heads_initial=heads0, heads_first=heads1, heads_resumed=heads2,
        config_hwd=np.array([H, W, D]), base_width=BASE_WIDTH, arch=ARCH, fc_scale=FC_SCALE,
        pairs=np.array(PAIRS), num_samples=NUM_SAMPLES)
    print("%s: %d logged iterations, loss %s" % (mode, len(s1[key]) + len(s2[key]),
                                                 " ".join("%.6f" % v for _, v in s1[key] + s2[key])))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("product", "reference"), required=True)
    ap.add_argument("--library", choices=("hostemu", "gfx950"), default="hostemu")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory(prefix="dcn_refloop_") as wd:
        run(a.mode, a.library, a.out, wd)

"""The drop-in claim, executed.

* With this repository's source root in front of the reference's on ``sys.path`` the packages MERGE (dcn_hip/_dropin.py):
  ``dense_correspondence.training.training``, the real ``SpartanDataset``, ``evaluation.evaluation``, ``utils.transformations``
  resolve to the reference, network / loss / ``resnet_dilated`` to the MI355X path.
* The reference's OWN driver -- ``DenseCorrespondenceTraining.run()`` + ``run_from_pretrained()``, training.py:46-601, imported
  from /root/reference (Python-2 text converted in memory, nothing copied) -- runs against the product (kernels host-emulated
  here) and logs the loss sequence the reference's own network / loss modules log (``tests/golden/training_loop_ref.npz``, made by
  ``tests/golden/make_training_loop_golden_from_reference.py``), writes the same files, and its checkpoints load back.
* Where /root/reference does not exist (the GPU box) the recorded batches are replayed through the same call sequence against
  the fixture: on the MI355X under ``-m gpu``, host-emulated otherwise.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

import reference_loop_runner as runner
import reference_py3

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "pytorch-dense-correspondence_amd")
FIXTURE = os.path.join(HERE, "golden", "training_loop_ref.npz")

needs_reference = pytest.mark.skipif(not reference_py3.available(), reason="the reference tree is not mounted on this box")


def _python(code, *paths, env=None):
    e = dict(os.environ, DCN_QUIET_SHIMS="0", **(env or {}))
    e.pop("PYTHONPATH", None)
    boot = "import sys\nsys.path[:0] = %r\n" % (list(paths),)
    return subprocess.run([sys.executable, "-c", boot + textwrap.dedent(code)], capture_output=True, text=True, env=e, timeout=600)


# ---- package merging and placeholders (no reference needed: a synthetic second source root) ------------------------------

def _fake_reference_tree(tmp_path):
    files = {
        "dense_correspondence/__init__.py": "",
        "dense_correspondence/training/__init__.py": "",
        "dense_correspondence/training/training.py": "MARK = 'their training'\n",
        "dense_correspondence/dataset/__init__.py": "",
        "dense_correspondence/dataset/spartan_dataset_masked.py":
            "MARK = 'their dataset'\nclass SpartanDatasetDataType:\n    SINGLE_OBJECT_WITHIN_SCENE = 0\n"
            "class SpartanDataset(object):\n    full = True\n    @staticmethod\n    def is_empty(t):\n        return len(t) == 1 and t[0] == -1\n",
        "dense_correspondence/network/__init__.py": "",
        "dense_correspondence/network/dense_correspondence_network.py": "MARK = 'their network'\n",
        "dense_correspondence/correspondence_tools/__init__.py": "",
        "dense_correspondence/correspondence_tools/correspondence_finder.py":
            "def batch_find_pixel_correspondences(*a, **k):\n    return ('their cpu sampler', k.get('device'))\n"
            "def random_sample_from_masked_image(m, n):\n    return 'theirs'\n",
        "dense_correspondence/evaluation/__init__.py": "",
        "dense_correspondence/evaluation/utils.py": "class PandaDataFrameWrapper(object):\n    pass\n",
        "modules/dense_correspondence_manipulation/__init__.py": "",
        "modules/dense_correspondence_manipulation/utils/__init__.py": "",
        "modules/dense_correspondence_manipulation/utils/transformations.py": "MARK = 'their transformations'\n",
        # Python-2 text: must NOT take the placeholder's place
        "modules/dense_correspondence_manipulation/utils/utils.py": "print 'py2'\n",
    }
    for rel, text in files.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    return str(tmp_path), str(tmp_path / "modules")


def test_packages_merge_with_a_second_source_root(tmp_path):
    ref, ref_modules = _fake_reference_tree(tmp_path)
    r = _python("""
        import dense_correspondence.network.dense_correspondence_network as n
        import dense_correspondence.training.training as t
        import dense_correspondence.dataset.spartan_dataset_masked as d
        import dense_correspondence_manipulation.utils.transformations as tf
        import dense_correspondence_manipulation.utils.utils as u
        import dense_correspondence.correspondence_tools.correspondence_finder as cf
        import dense_correspondence.evaluation.utils as eu
        import dense_correspondence.loss_functions.loss_composer as lc
        assert n.__file__.startswith(%r) and hasattr(n, "DenseCorrespondenceNetwork"), n.__file__
        assert t.MARK == "their training" and tf.MARK == "their transformations"
        assert d.MARK == "their dataset" and d.SpartanDataset.full          # the placeholder stepped aside ...
        assert lc.SpartanDataset is d.SpartanDataset                        # ... and the loss composer sees the real class
        assert u.getPaddedString(7) == "000007" and u.__file__.startswith(%r)   # py2 text there: the placeholder stays
        assert cf.batch_find_pixel_correspondences(0, 0, 0, 0, device="CPU") == ("their cpu sampler", "CPU")
        assert cf.random_sample_from_masked_image(0, 1) == "theirs" and cf.create_non_correspondences.__module__ == cf.__name__
        assert eu.PandaDataFrameWrapper.__module__.endswith("__reference") and hasattr(eu, "extract_descriptor_images_for_scene")
        try:
            eu.no_such_name
        except AttributeError as e:
            assert "no_such_name" in str(e)
        else:
            raise AssertionError
        print("OK")
        """ % (PKG, PKG), PKG, ref, ref_modules)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert "dense_correspondence_manipulation.utils.utils" in r.stderr and "SyntaxError" in r.stderr   # said so, once


def test_source_root_added_after_first_import_is_merged_too(tmp_path):
    """utils.add_dense_correspondence_to_python_path() (training.py:27) extends sys.path at run time."""
    ref, ref_modules = _fake_reference_tree(tmp_path)
    r = _python("""
        import dense_correspondence, dense_correspondence.network
        try:
            import dense_correspondence.training.training
        except ImportError:
            pass
        else:
            raise AssertionError("nothing provides it yet")
        sys.path += [%r, %r]
        import dense_correspondence.training.training as t
        assert t.MARK == "their training"
        print("OK")
        """ % (ref, ref_modules), PKG)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_placeholders_alone():
    r = _python("""
        import dense_correspondence.dataset.spartan_dataset_masked as d
        import dense_correspondence_manipulation.utils.constants as c
        import dense_correspondence.correspondence_tools.correspondence_finder as cf
        import pytorch_segmentation_detection.transforms as tr, pytorch_segmentation_detection.models.fcn as fcn
        assert d.SpartanDataset.is_empty(d.SpartanDataset.empty_tensor()) and len(c.DEFAULT_IMAGE_MEAN) == 3
        try:
            cf.batch_find_pixel_correspondences(None, None, None, None, device="CPU")
        except ValueError as e:
            assert "reference" in str(e)
        else:
            raise AssertionError
        assert not hasattr(cf, "random_sample_from_masked_image") and not hasattr(fcn, "FCN_32s")
        try:
            tr.ComposeJoint([])
        except NotImplementedError:
            print("OK")
        """, PKG)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


# ---- the real reference tree ---------------------------------------------------------------------------------------------

@needs_reference
def test_import_probe_against_the_real_reference():
    """The five modules VERDICT r5 probed (ModuleNotFoundError then) + where each name of training.py:20-43 comes from."""
    r = _python("""
        import os
        sys.path.insert(0, %r)
        import reference_py3 as rp
        rp.install(); rp.install_third_party_stubs()
        os.environ["DC_SOURCE_DIR"] = rp.REF
        sys.path += [rp.REF, os.path.join(rp.REF, "modules")]
        import importlib
        theirs = ["dense_correspondence.training.training", "dense_correspondence.dataset.dense_correspondence_dataset_masked",
                  "dense_correspondence.dataset.spartan_dataset_masked", "dense_correspondence.evaluation.evaluation",
                  "dense_correspondence_manipulation.utils.transformations", "dense_correspondence_manipulation.utils.utils",
                  "dense_correspondence_manipulation.utils.constants", "dense_correspondence.dataset.scene_structure",
                  "dense_correspondence.correspondence_tools.correspondence_augmentation"]
        ours = ["dense_correspondence.network.dense_correspondence_network", "dense_correspondence.loss_functions.loss_composer",
                "dense_correspondence.loss_functions.pixelwise_contrastive_loss",
                "pytorch_segmentation_detection.models.resnet_dilated", "dense_correspondence.correspondence_tools.correspondence_finder",
                "dense_correspondence.evaluation.utils", "tensorboard_logger"]
        for m in theirs:
            f = importlib.import_module(m).__file__
            assert f.startswith(rp.REF), (m, f)
        for m in ours:
            f = importlib.import_module(m).__file__
            assert f.startswith(%r), (m, f)
        import dense_correspondence.training.training as t
        assert t.SpartanDataset.__module__ == "dense_correspondence.dataset.spartan_dataset_masked" and hasattr(t.SpartanDataset, "get_within_scene_data")
        assert t.DenseCorrespondenceNetwork.__module__ == "dense_correspondence.network.dense_correspondence_network"
        assert hasattr(t.DenseCorrespondenceNetwork, "forward_pair")                         # (the product's class)
        assert t.DenseCorrespondenceEvaluation.__module__ == "dense_correspondence.evaluation.evaluation"
        from dense_correspondence.evaluation.utils import PandaDataFrameWrapper               # evaluation.py:34, handed on
        import dense_correspondence.correspondence_tools.correspondence_finder as cf
        assert callable(cf.pinhole_projection_image_to_world) and cf.batch_find_pixel_correspondences.__module__ == cf.__name__
        print("OK")
        """ % (HERE, PKG), PKG)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def _assert_matches_fixture(z, g, first_tol, later_tol):
    assert list(z["served"]) == list(g["served"]) and list(z["loss_steps"]) == list(g["loss_steps"])
    assert np.array_equal(z["learning_rate"], g["learning_rate"])
    for k in ("loss", "match_loss", "masked_non_match_loss", "background_non_match_loss"):
        rel = np.abs(z[k] - g[k]) / np.abs(g[k])
        assert rel[0] <= first_tol and rel.max() <= later_tol, (k, rel)
    if "files_first" in z.files:
        assert list(z["files_first"]) == list(g["files_first"]) and list(z["files_resumed"]) == list(g["files_resumed"])
        assert list(z["param_names"]) == list(g["param_names"])
    for k in ("norms_initial", "norms_first", "norms_resumed"):
        # Adam's first steps are lr * sign(g) for every weight: a round-off-sized gradient may go the other way, 2 lr apart
        d = np.abs(z[k] - g[k])
        assert (d <= 1e-3 * np.abs(g[k]) + 2e-3).all(), (k, d.max())
    assert np.array_equal(z["heads_initial"], g["heads_initial"])


@needs_reference
def test_fixture_regenerates_from_the_reference(tmp_path):
    out = str(tmp_path / "ref.npz")
    subprocess.check_call([sys.executable, os.path.join(HERE, "reference_loop_runner.py"), "--mode", "reference", "--out", out],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    z, g = np.load(out), np.load(FIXTURE)
    for k in g.files:
        assert np.array_equal(z[k], g[k]), k
    assert str(z["network_file"]).startswith(reference_py3.REF) and str(z["loss_file"]).startswith(reference_py3.REF)


@needs_reference
def test_reference_training_driver_runs_on_the_product(tmp_path):
    """training.py unchanged (2to3 in memory), real SpartanDataset subclass, reference's training.yaml; network + loss = product."""
    out = str(tmp_path / "prod.npz")
    p = subprocess.run([sys.executable, os.path.join(HERE, "reference_loop_runner.py"), "--mode", "product", "--library", "hostemu",
                        "--out", out], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    z, g = np.load(out), np.load(FIXTURE)
    assert str(z["training_file"]).startswith(reference_py3.REF) and str(z["dataset_file"]).startswith(reference_py3.REF)
    assert str(z["network_file"]).startswith(PKG) and str(z["loss_file"]).startswith(PKG)
    assert "empty data, continuing" in p.stdout                        # the loader's sentinel branch (training.py:304-306)
    _assert_matches_fixture(z, g, 1e-4, 1e-3)


# ---- replay of the recorded run (no reference needed; on the MI355X under -m gpu) ------------------------------------------

def _replay(device, tmp_path):
    """The call sequence of training.py:228-456 + 169-226 restated (the GPU box has no reference tree to import it from):
    same samples in the recorded order, stock ``optim.Adam``, the learning-rate decay of :544-558, ``save_network`` /
    ``load_pretrained`` through files."""
    import pytorch_segmentation_detection.models.resnet_dilated as rd
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    from oracle import synth
    g = np.load(FIXTURE)
    _, sd0 = runner.initial_state_dict()

    class NarrowTestNet(getattr(rd, runner.ARCH)):
        def __init__(self, num_classes):
            super(NarrowTestNet, self).__init__(num_classes=num_classes, base_width=runner.BASE_WIDTH)
            self.load_state_dict(sd0)
    NarrowTestNet.arch, NarrowTestNet.attr = getattr(rd, runner.ARCH).arch, getattr(rd, runner.ARCH).attr
    rd.NarrowTestNet = NarrowTestNet
    cfg = {"descriptor_dimension": runner.D, "image_width": runner.W, "image_height": runner.H, "normalize": False,
           "backbone": {"model_class": "Resnet", "resnet_name": "NarrowTestNet"}}
    samples = runner.make_samples()
    served = list(g["served"])
    out = {k: [] for k in ("loss", "match_loss", "masked_non_match_loss", "background_non_match_loss", "learning_rate", "loss_steps")}

    def build():
        dcn = DenseCorrespondenceNetwork.from_config(dict(cfg), load_stored_params=False).to(device)
        dcn.train()
        return dcn, torch.optim.Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4)

    def iterations(dcn, opt, it, stop_after):
        pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
        while True:
            it += 1
            s = samples[served.pop(0)]
            if s[0] == -1:
                continue
            img_a, img_b = s[1].unsqueeze(0).to(device), s[2].unsqueeze(0).to(device)
            lists = [t.to(device) for t in s[3:11]]
            opt.zero_grad()
            if it % 2 == 0:
                for grp in opt.param_groups:
                    grp["lr"] = grp["lr"] * 0.9
            pa = dcn.process_network_output(dcn.forward(img_a), 1)
            pb = dcn.process_network_output(dcn.forward(img_b), 1)
            loss, ml, mk, bg, _ = loss_composer.get_loss(pcl, torch.tensor([0]), pa, pb, *lists)
            loss.backward()
            opt.step()
            out["learning_rate"].append(opt.param_groups[0]["lr"])
            for k, v in (("loss", loss), ("match_loss", ml), ("masked_non_match_loss", mk), ("background_non_match_loss", bg)):
                out[k].append(v.item())
            out["loss_steps"].append(it)
            if it > stop_after:
                return it

    def summary(sd):
        names = sorted(sd.keys())
        return (np.array([float(sd[k].double().norm()) for k in names]),
                np.array([sd[k].double().reshape(-1)[:4].tolist() + [0.0] * max(0, 4 - sd[k].numel()) for k in names]))
    dcn, opt = build()
    out["norms_initial"], out["heads_initial"] = summary({k: v.cpu() for k, v in dcn.state_dict().items()})
    it = iterations(dcn, opt, 0, runner.FIRST_RUN_ITERATIONS)
    torch.save(dcn.state_dict(), str(tmp_path / "a.pth"))
    torch.save(opt.state_dict(), str(tmp_path / "a.pth.opt"))
    out["norms_first"], _ = summary(torch.load(str(tmp_path / "a.pth"), map_location="cpu"))
    dcn, opt = build()
    dcn.load_state_dict(torch.load(str(tmp_path / "a.pth")))
    opt.load_state_dict(torch.load(str(tmp_path / "a.pth.opt")))
    iterations(dcn, opt, it, it + runner.RESUMED_ITERATIONS)
    out["norms_resumed"], _ = summary({k: v.cpu() for k, v in dcn.state_dict().items()})
    out["served"] = g["served"]
    assert not served

    class Z(dict):
        files = list(out.keys())
    return Z({k: np.asarray(v) for k, v in out.items()}), g


def test_recorded_training_loop_replayed_host_emulated(tmp_path):
    from helpers import use_emulation_library
    use_emulation_library()
    z, g = _replay(torch.device("cpu"), tmp_path)
    _assert_matches_fixture(z, g, 1e-4, 1e-3)


@pytest.mark.gpu
def test_recorded_training_loop_replayed_on_the_gpu(tmp_path):
    from helpers import use_gfx950_library
    use_gfx950_library()
    z, g = _replay(torch.device("cuda:0"), tmp_path)
    _assert_matches_fixture(z, g, 1e-4, 1e-3)

"""CPU-only tests of the host side: cfg parser, module construction / state_dict / .weights I/O (no compute), the C-ABI
library's exports, the no-CPU-fallback rule, the oracle-isolation rule."""
import ctypes
import os
import re
import subprocess
import sys
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
PKG = os.path.join(ROOT, "mit-driverless-cv-traininginfra_amd")


def test_library_exports_every_declared_symbol():
    from mdcv import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (mdcv_\w+)", out))
    assert exported == set(protos), exported ^ set(protos)       # header and library agree exactly


def test_argument_errors_are_reported_without_a_gpu():
    from mdcv import _lib
    L = _lib.lib()
    assert L.conv2d(1, 0, None, 8, None, None, 8, None, None, 0, None, 1, 4, 4, 8, 4, 4, 8, 3, 3, 1, 1, 1, None) == -1
    assert L.conv2d_stats_rows(129) == 2
    assert L.conv2d_wgrad_splits(1, 100000, 256, 1152) >= 1
    assert L.yolo_head_workspace_bytes(2, 3, 13, 13) > 0


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mdcv import _lib
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    with pytest.raises(_lib.MdcvError):
        KeypointNet()(torch.zeros(1, 3, 80, 80))
    with pytest.raises(_lib.MdcvError):
        CrossRatioLoss("l1_softargmax", True, 0.0, 0.0)(None, torch.zeros(2, 7, 2), None, torch.zeros(2, 7, 2))
    cwd = os.getcwd()
    os.chdir(os.path.join(G, "mini"))
    try:
        from mdcv.yolo.models import Darknet
        net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
    finally:
        os.chdir(cwd)
    with pytest.raises(_lib.MdcvError):
        net(torch.zeros(1, 3, 64, 64), torch.zeros(1, 1, 5))


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|/oracle/|oracle\.", re.M)
    for dp, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                code = "\n".join(l for l in txt.split("\n") if not l.strip().startswith(("#", "//", "*", '"""')))
                assert not pat.search(code), os.path.join(dp, f)


def test_parse_model_config_contract(tmp_path):
    from mdcv.yolo.utils.parse_config import parse_model_config
    p = tmp_path / "a.cfg"
    p.write_text("# comment\n[net]\n width = 64 \nheight=64\n\n[convolutional]\nfilters=16\nsize=3\nstride=1\n  [route]\nlayers = -1, 4\n")
    d = parse_model_config(str(p))
    assert d[0] == {"type": "net", "width": "64", "height": "64"}
    assert d[1] == {"type": "convolutional", "batch_normalize": 0, "filters": "16", "size": "3", "stride": "1"}
    assert d[2] == {"type": "route", "layers": "-1, 4"}
    from oracle.yolo_oracle import parse_cfg
    assert parse_cfg(os.path.join(G, "mini", "mini.cfg")) == parse_model_config(os.path.join(G, "mini", "mini.cfg"))


def test_darknet_structure_state_dict_and_weights_io(tmp_path):
    z = np.load(os.path.join(G, "mini_darknet.npz"))
    cwd = os.getcwd()
    os.chdir(os.path.join(G, "mini"))
    try:
        from mdcv.yolo.models import Darknet, YOLOLayer
        net = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
        net.load_weights("mini.weights", net.get_start_weight_dim())
    finally:
        os.chdir(cwd)
    assert list(net.state_dict().keys()) == [str(k) for k in z["param_names"]]
    assert net.get_anchors() == [[4.0, 6.0], [6.0, 10.0], [10.0, 8.0], [12.0, 20.0], [20.0, 16.0], [24.0, 36.0]]
    assert net.get_loss_constant() == [2.0, 1.6, 25.0, 0.1] and net.img_size() == (64, 64) and net.get_threshs() == (0.8, 0.25, 0.5)
    assert net.get_onnx_name() == "mini_6464.onnx" and net.get_num_classes() == 1 and net.get_bw() is False
    yl = [m[0] for m in net.module_list if isinstance(m[0], YOLOLayer)]
    assert yl[0].anchors == [[12.0, 20.0], [20.0, 16.0], [24.0, 36.0]] and yl[1].anchors == [[4.0, 6.0], [6.0, 10.0], [10.0, 8.0]]
    p = tmp_path / "rt.weights"
    net.save_weights(str(p))
    assert open(p, "rb").read() == open(os.path.join(G, "mini", "mini.weights"), "rb").read()
    # an 80-class file initialises a narrower head through start_weight_dim (Q13): emulate with 18 -> keep first 12 filters
    from oracle.yolo_oracle import DarknetOracle, read_anchor_row
    os.chdir(os.path.join(G, "mini"))
    try:
        orc = DarknetOracle("mini.cfg", anchors=read_anchor_row("dataset/train.csv"))
        orc.load_weights("mini.weights", [18, 18])
    finally:
        os.chdir(cwd)
    sd = net.state_dict()
    assert torch.equal(sd["module_list.11.conv_11.weight"], orc.params["conv11.weight"])
    assert torch.equal(sd["module_list.3.batch_norm_3.running_var"], orc.params["bn3.running_var"])


def test_keypointnet_state_dict_matches_reference_names():
    from mdcv.rektnet.keypoint_net import KeypointNet
    z = np.load(os.path.join(G, "rektnet_net.npz"))
    torch.manual_seed(0)
    net = KeypointNet()
    ref_keys = [k[4:] for k in z.files if k.startswith("sd::")]
    assert list(net.state_dict().keys()) == ref_keys
    assert sum(p.numel() for p in net.parameters()) == 311383
    # kaiming-normal fan_out init: std = sqrt(2 / (Cout*k*k)), zero biases, BN 1/0
    w = net.res3.conv2.weight
    assert abs(float(w.std()) - (2.0 / (64 * 9)) ** 0.5) < 0.003 and float(net.conv.bias.abs().max()) == 0.0
    assert float(net.bn.weight.min()) == 1.0


def test_cross_ratio_prints_and_unknown_type():
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    crit = CrossRatioLoss("bogus", True, 0.1, 0.2)
    with pytest.raises(NameError):
        crit(None, torch.zeros(1, 7, 2), None, torch.zeros(1, 7, 2))


def test_models_with_cached_plans_deepcopy_and_pickle():
    """RektNet/train_eval.py:99 keeps `best_model = copy.deepcopy(model)` after a forward; the launch plans a forward leaves behind
    hold ctypes function pointers and closures, which must not travel with the copy (it rebuilds them lazily)."""
    import copy
    import pickle
    from mdcv.rektnet.keypoint_net import KeypointNet
    from mdcv.yolo.models import Darknet

    class FakePlan:                        # what a cached plan looks like to pickle: ctypes pointers + closures
        def __init__(self):
            self.fn = ctypes.CFUNCTYPE(ctypes.c_int)(lambda: 0)
            self.closure = lambda s: 0

    cwd = os.getcwd()
    os.chdir(os.path.join(G, "mini"))
    try:
        yolo = Darknet("mini.cfg", 2.0, 1.6, 25.0, 0.1, False)
    finally:
        os.chdir(cwd)
    for net in (KeypointNet(), yolo):
        net._flatten()                     # parameters become views of one flat buffer, as after the first forward
        net._plans[("k",)] = FakePlan()
        net._last_train_plan = FakePlan()
        before = {k: v.clone() for k, v in net.state_dict().items()}
        for dup in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
            assert dup._plans == {} and not hasattr(dup, "_last_train_plan") and not dup._flat_ok()
            for k, v in dup.state_dict().items():
                assert torch.equal(v, before[k]), k
            p0 = next(dup.parameters())
            with torch.no_grad():
                p0.add_(1.0)                # the copy owns its storage
            assert torch.equal(net.state_dict()[next(iter(before))], before[next(iter(before))])
        assert ("k",) in net._plans        # the original keeps its plans


def test_cross_ratio_rejects_other_keypoint_counts():
    from mdcv.rektnet.cross_ratio_loss import CrossRatioLoss
    crit = CrossRatioLoss("l1_softargmax", True, 0.1, 0.2)
    with pytest.raises(ValueError):
        crit(None, torch.zeros(2, 5, 2), None, torch.zeros(2, 5, 2))


def test_bench_cli_contract_without_gpu():
    """bench.py refuses to run without a GPU (no CPU fallback) -- and says so instead of crashing."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


def test_kernel_fingerprint_tracks_sources(tmp_path):
    from mdcv import _fingerprint as F
    a = F.kernel_fingerprint()
    assert a == F.kernel_fingerprint() and len(a) == 16
    assert any(f.endswith("conv_igemm.hip") for f in F.fingerprint_files()) and any(f.endswith("engine.py") for f in F.fingerprint_files())

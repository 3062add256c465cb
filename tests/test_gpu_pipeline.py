"""GPU parity for the detect -> crop -> keypoints glue."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pipeline_oracle as PL          # noqa: E402
from oracle import postprocess_oracle as PO       # noqa: E402
from oracle import rektnet_oracle as RO           # noqa: E402

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _boxes(B, K, H, W, rng):
    c = rng.random((B, K, 2)) * [W, H]
    s = rng.random((B, K, 2)) * [W * 0.3, H * 0.4] + 1.5
    b = np.concatenate([c - s / 2, c + s / 2], -1).astype(np.float32)
    b[0, 0] = [-9.5, -3.25, 7.75, 11.5]                  # sticks out of the frame
    b[0, 1] = [W - 3.5, H - 2.5, W + 20, H + 20]
    b[0, 2] = [5.0, 5.0, 5.0, 5.0]                       # empty -> one pixel
    b[0, 3] = [W + 50, H + 50, W + 60, H + 60]           # fully outside -> edge pixel
    return b


@pytest.mark.parametrize("B,K,H,W,out,scale,offset", [(3, 7, 97, 131, (80, 80), (1.0, 1.0), (0.0, 0.0)),
                                                      (2, 5, 608, 608, (80, 80), (1.0, 1.0), (0.0, 0.0)),
                                                      (2, 6, 300, 480, (80, 80), (480 / 608.0, 480 / 608.0), (0.0, -90.0)),
                                                      (1, 4, 64, 64, (16, 40), (1.0, 1.0), (0.0, 0.0)),
                                                      (2, 4, 50, 70, (256, 256), (1.0, 1.0), (0.0, 0.0))])
def test_crop_resize_bit_exact(B, K, H, W, out, scale, offset):
    from mdcv.pipeline import crop_resize
    rng = np.random.default_rng(B * 100 + K)
    frames = rng.random((B, 3, H, W), dtype=np.float32)
    boxes = _boxes(B, K, H / scale[1] if scale[1] != 1 else H, W / scale[0] if scale[0] != 1 else W, rng)
    count = rng.integers(0, K + 1, B).astype(np.int32)
    count[0] = K
    crops, owner, M = crop_resize(_dev(frames), _dev(boxes), _dev(count), out, scale, offset, pad_rows_to=8)
    ref, ref_owner = PL.crop_resize(frames, boxes, count, out[0], out[1], scale, offset)
    assert M == ref.shape[0] and crops.shape[0] % 8 == 0
    np.testing.assert_array_equal(crops[:M].cpu().numpy(), ref)
    np.testing.assert_array_equal(owner[:M].cpu().numpy(), ref_owner)
    assert float(crops[M:].abs().sum()) == 0.0


def test_crop_resize_identity_and_errors():
    from mdcv.pipeline import crop_resize
    rng = np.random.default_rng(1)
    frames = rng.random((1, 3, 120, 120), dtype=np.float32)
    boxes = np.array([[[20.0, 30.0, 100.0, 110.0]]], np.float32)
    crops, _, M = crop_resize(_dev(frames), _dev(boxes), _dev(np.array([1], np.int32)), (80, 80))
    assert M == 1
    np.testing.assert_array_equal(crops[0].cpu().numpy(), frames[0, :, 30:110, 20:100])    # 80x80 box: pure copy
    with pytest.raises(ValueError):
        crop_resize(_dev(frames), _dev(boxes), _dev(np.array([1], np.int32)), (257, 80))
    with pytest.raises(Exception):
        crop_resize(torch.from_numpy(frames), torch.from_numpy(boxes), torch.tensor([1], dtype=torch.int32))


def test_joint_pipeline_vs_oracle():
    """Recorded detector outputs -> boxes -> crops -> KeypointNet (fp32 kernels) against the three oracles chained."""
    from mdcv.pipeline import JointPipeline
    from mdcv.rektnet.keypoint_net import KeypointNet
    rng = np.random.default_rng(5)
    B, N, C, S = 3, 2028, 1, 208
    out = np.zeros((B, N, 5 + C), np.float32)
    out[:, :, 0:2] = rng.random((B, N, 2)) * S
    out[:, :, 2:4] = rng.random((B, N, 2)) * 50 + 6
    out[:, :, 4] = rng.random((B, N)) * 0.7
    out[:, :, 5:] = rng.random((B, N, C))
    for b in range(B):
        hot = rng.permutation(N)[: 3 + 4 * b]
        out[b, hot, 4] = 0.8 + 0.2 * rng.random(hot.size)
    out[2, :, 4] *= 0.5                                   # image 2: nothing above the threshold
    imgs = rng.random((B, 3, S, S), dtype=np.float32)

    class Replay(torch.nn.Module):
        def get_threshs(self):
            return 0.8, 0.25, 0.5

        def img_size(self):
            return S, S

        def forward(self, x):
            return _dev(out)

    torch.manual_seed(3)
    kp = KeypointNet(7, (80, 80), precision="fp32").cuda().eval()
    with torch.no_grad():                                 # non-trivial running statistics
        for m in kp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    pipe = JointPipeline(Replay(), kp, max_cones=16, bucket=8)
    res = pipe(_dev(imgs))
    # oracle chain
    boxes = np.zeros((B, 16, 4), np.float32); count = np.zeros(B, np.int32)
    for b in range(B):
        r = PO.postprocess_image(out[b], np.zeros((1, 5), np.float32), 0.8, 0.25, 0.5, S, S)
        n = min(r["count"], 16)
        boxes[b, :n] = r["boxes"][:n]; count[b] = n
    crops, owner = PL.crop_resize(imgs, boxes, count, 80, 80)
    assert res["num"] == crops.shape[0] > 0 and count[2] == 0
    np.testing.assert_array_equal(res["crops"].cpu().numpy(), crops)
    np.testing.assert_array_equal(res["owner"].cpu().numpy(), owner)
    sd = {k: v.detach().cpu() for k, v in kp.state_dict().items()}
    _, pts = RO.keypoint_forward(torch.from_numpy(crops), sd, train=False)
    np.testing.assert_allclose(res["keypoints"].cpu().numpy(), pts.numpy(), atol=2e-4)
    kf = res["keypoints_frame"].cpu().numpy()
    assert kf.shape == (crops.shape[0], 7, 2) and kf.min() >= 0 and kf.max() <= S


@pytest.mark.parametrize("B,K,H,W,out,scale,offset", [(3, 7, 97, 131, (80, 80), (1.0, 1.0), (0.0, 0.0)),
                                                      (2, 5, 608, 608, (80, 80), (1.0, 1.0), (0.0, 0.0)),
                                                      (2, 6, 300, 480, (80, 80), (480 / 608.0, 480 / 608.0), (0.0, -90.0)),
                                                      (1, 4, 64, 64, (16, 40), (1.0, 1.0), (0.0, 0.0)),
                                                      (2, 4, 50, 70, (256, 256), (1.0, 1.0), (0.0, 0.0))])
def test_crop_resize_u8_bit_exact(B, K, H, W, out, scale, offset):
    """The reference's rule (8-bit image -> cv2 fixed-point INTER_LINEAR -> /255, RektNet/dataset.py:35-38,52): every output value
    equal to the oracle's, from uint8 frames and from [0,1] float frames (quantised on device)."""
    from mdcv.pipeline import crop_resize
    rng = np.random.default_rng(B * 100 + K + 7)
    frames8 = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    boxes = _boxes(B, K, H / scale[1] if scale[1] != 1 else H, W / scale[0] if scale[0] != 1 else W, rng)
    count = rng.integers(0, K + 1, B).astype(np.int32)
    count[0] = K
    ref, ref_owner = PL.crop_resize(frames8, boxes, count, out[0], out[1], scale, offset, u8=True)
    crops, owner, M = crop_resize(_dev(frames8), _dev(boxes), _dev(count), out, scale, offset, pad_rows_to=8)
    assert M == ref.shape[0]
    np.testing.assert_array_equal(crops[:M].cpu().numpy(), ref)
    np.testing.assert_array_equal(owner[:M].cpu().numpy(), ref_owner)
    framesf = (frames8.astype(np.float64) / 255.0).astype(np.float32)        # what the detector is fed
    crops2, _, M2 = crop_resize(_dev(framesf), _dev(boxes), _dev(count), out, scale, offset, u8=True)
    np.testing.assert_array_equal(crops2[:M2].cpu().numpy(), ref)
    assert float(np.abs(np.rint(ref * 255) / 255 - ref).max()) < 1e-7             # on the 8-bit grid


def _write_cfg(tmp_path, size, classes):
    import bench
    return bench.write_yolo_cfg(str(tmp_path), size=size, classes=classes)


def test_joint_608_end_to_end_vs_chained_oracles(tmp_path):
    """BASELINE.json configs[4] at its real size, one GPU's share scaled to B = 4: eval-mode yolo_baseline 608x608 (fp32 kernels) ->
    conf filter / NMS -> <= 16 crops per frame from the uint8 frames -> batched KeypointNet eval, every stage against its oracle on the
    previous stage's HIP output (detector rows vs the CPU oracle forward at B = 2: the [B, 22743, 6] eval tensor)."""
    from oracle import yolo_oracle as yo
    from mdcv.pipeline import JointPipeline
    from mdcv.yolo.models import Darknet
    from mdcv.rektnet.keypoint_net import KeypointNet
    cfg = _write_cfg(tmp_path, 608, 1)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        torch.manual_seed(11)
        net = Darknet(cfg, 2.0, 1.6, 25.0, 0.1, True, precision="fp32")
        with torch.no_grad():                                 # non-trivial running statistics (eval mode uses them)
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    g = torch.Generator().manual_seed(int(m.num_features))
                    m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                    m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
        wpath = str(tmp_path / "init.weights")
        net.save_weights(wpath)
        orc = yo.DarknetOracle(cfg, anchors=yo.VANILLA_ANCHORS)
        orc.load_weights(wpath, [18, 18, 18])
    finally:
        os.chdir(cwd)
    net = net.cuda().eval()
    rng = np.random.default_rng(8)
    B = 4
    frames8 = rng.integers(0, 256, (B, 3, 608, 608), dtype=np.uint8)
    imgs = (frames8.astype(np.float64) / 255.0).astype(np.float32)
    with torch.no_grad():
        rows = net(_dev(imgs))
    assert tuple(rows.shape) == (B, 22743, 6)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref_rows = orc.forward(torch.from_numpy(imgs[:2]), None, bn_train=False).numpy()
    got = rows[:2].cpu().numpy()
    np.testing.assert_allclose(got[..., :4], ref_rows[..., :4], rtol=2e-3, atol=2e-2)      # boxes in pixels of a 608 frame
    np.testing.assert_allclose(got[..., 4:], ref_rows[..., 4:], rtol=0, atol=2e-3)        # confidences / class scores in [0, 1]
    torch.manual_seed(3)
    kp = KeypointNet(7, (80, 80), precision="fp32").cuda().eval()
    with torch.no_grad():
        for m in kp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    conf = float(np.quantile(got[..., 4], 0.995))            # a random-init detector has no confident rows: keep its top 0.5 %
    pipe = JointPipeline(net, kp, conf_thres=conf, nms_thres=0.25, max_cones=16, bucket=64)
    res = pipe(_dev(imgs), frames=_dev(frames8))
    out = rows.cpu().numpy()
    boxes = np.zeros((B, 16, 4), np.float32); count = np.zeros(B, np.int32)
    for b in range(B):
        r = PO.postprocess_image(out[b], np.zeros((1, 5), np.float32), conf, 0.25, 0.5, 608, 608)
        n = min(r["count"], 16)
        boxes[b, :n] = r["boxes"][:n]; count[b] = n
        assert int(res["det"].count[b]) == r["count"]
        np.testing.assert_array_equal(res["det"].image(b)["index"].cpu().numpy(), r["index"])
    crops, owner = PL.crop_resize(frames8, boxes, count, 80, 80, u8=True)
    assert res["num"] == crops.shape[0] and res["num"] >= B                       # every frame contributes
    np.testing.assert_array_equal(res["crops"].cpu().numpy(), crops)
    np.testing.assert_array_equal(res["owner"].cpu().numpy(), owner)
    sdk = {k: v.detach().cpu() for k, v in kp.state_dict().items()}
    _, pts = RO.keypoint_forward(torch.from_numpy(crops), sdk, train=False)
    np.testing.assert_allclose(res["keypoints"].cpu().numpy(), pts.numpy(), atol=2e-4)
    kf = res["keypoints_frame"].cpu().numpy()
    assert kf.shape == (crops.shape[0], 7, 2) and kf.min() >= 0 and kf.max() <= 608

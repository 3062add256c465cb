"""Drop-in for the reference's RektNet hot-path modules (keypoint_net.py, resnet.py, cross_ratio_loss.py)."""

"""Residual block of RektNet (reference: RektNet/resnet.py:8-27) — parameter container.

    out = relu( shortcut_bn(shortcut_conv(x)) + bn2(conv2( relu(bn1(conv1(x))) )) )
    conv1: 3x3 dilation 2 padding 2 ; conv2: 3x3 padding 1 ; shortcut_conv: 1x1 ; every conv has a bias.

The module keeps the reference's attribute names (state_dict compatibility).  Its math runs inside KeypointNet's fused
HIP plan (keypoint_net.py); calling the block on its own is not part of the reference's call surface.
"""
import torch.nn as nn


class ResNet(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=2, dilation=2)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.relu2 = nn.ReLU()
        self.shortcut_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1)
        self.shortcut_bn = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        raise RuntimeError("ResNet blocks execute inside KeypointNet's fused HIP plan; call KeypointNet.forward instead")

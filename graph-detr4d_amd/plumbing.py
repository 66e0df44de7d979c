"""Host-side plumbing in front of the decoder: images -> per-level multi-camera feature maps.

Mirror of `Detr3D.extract_img_feat` / `extract_feat` (projects/mmdet3d_plugin/models/detectors/detr3d.py:39-72): the
(B, N, 3, H, W) image batch is folded to (B*N, 3, H, W), pushed through the image backbone and neck, and every level is
unfolded to (B, N, C, H_l, W_l) fp32 - the `value` list the attention modules take.  The backbone and neck are the
caller's (mmdet's ResNet / VoVNet + FPN in the reference; PyTorch-ROCm modules here) and stay plain PyTorch: they are
outside the hot path.  `ResNet18FPN` is a small stand-in for BASELINE.json configs[0] ("DETR3D 1-layer
decoder, 100 queries, 6 x (3 x 256 x 256) synthetic images, ResNet18 backbone on CPU"): a BasicBlock ResNet-18 with
an FPN configured like the reference's necks (config detr3d_res50.py: start_level=1, add_extra_convs='on_output',
num_outs=4, relu_before_extra_convs=True), giving strides 8 / 16 / 32 / 64.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(out)) + identity)


class ResNet18FPN(nn.Module):
    """ResNet-18 (stages of 2 BasicBlocks, 64/128/256/512 channels, strides 4/8/16/32) + FPN over C3..C5 with one extra
    stride-2 level on the last output: 4 maps of `out_channels` at strides 8/16/32/64."""

    def __init__(self, out_channels=256):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(3, 2, 1))
        chans, stages, cin = (64, 128, 256, 512), [], 64
        for i, c in enumerate(chans):
            stages.append(nn.Sequential(_BasicBlock(cin, c, 1 if i == 0 else 2), _BasicBlock(c, c, 1)))
            cin = c
        self.stages = nn.ModuleList(stages)
        self.lateral_convs = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in chans[1:])
        self.fpn_convs = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in chans[1:])
        self.extra_conv = nn.Conv2d(out_channels, out_channels, 3, stride=2, padding=1)

    def forward(self, img):
        x = self.stem(img)
        feats = []
        for stage in self.stages:
            x = stage(x)
            feats.append(x)
        lat = [conv(f) for conv, f in zip(self.lateral_convs, feats[1:])]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[-2:], mode='nearest')
        outs = [conv(f) for conv, f in zip(self.fpn_convs, lat)]
        outs.append(self.extra_conv(outs[-1]))               # add_extra_convs='on_output' (first extra level: no ReLU)
        return outs


class ImageFeatureExtractor(nn.Module):
    """`Detr3D.extract_feat` (detectors/detr3d.py:39-72): fold cameras into the batch, backbone (+ neck), unfold.

    forward(img (B, N, 3, H, W) or (N, 3, H, W), img_metas) -> list of L tensors (B, N, C, H_l, W_l) fp32 on
    `out_device` (the reference's @auto_fp16(out_fp32=True): whatever precision the backbone ran in, the decoder
    receives fp32).  Every meta gets `input_shape` = the image size fed to the backbone (:43-46).

    channels_last=True: the same logical (B, N, C, H, W) tensors, STORED (B, N, H, W, C) - torch.channels_last on the
    folded (B*N, C, H, W) maps, which is what a backbone run in channels_last memory format returns anyway.  The decoder's
    cross-attention then gathers them in place (ops.PyramidView.channels_last_levels): the reference's per-layer
    flatten / transpose / cat (deform3d_cross_attn.py:264-276) and this build's once-per-sample copy both disappear."""

    def __init__(self, backbone, neck=None, out_device=None, channels_last=False):
        super().__init__()
        self.img_backbone, self.img_neck = backbone, neck
        self.out_device = out_device
        self.channels_last = bool(channels_last)

    def forward(self, img, img_metas):
        if img is None:
            return None
        input_shape = tuple(img.shape[-2:])
        for meta in img_metas:
            meta.update(input_shape=input_shape)
        if img.dim() == 4:                                   # a single sample's cameras
            img = img.unsqueeze(0)
        b, n = img.shape[:2]
        x = img.reshape(b * n, *img.shape[2:])
        feats = self.img_backbone(x)
        if isinstance(feats, dict):
            feats = list(feats.values())
        if self.img_neck is not None:
            feats = self.img_neck(feats)
        out = []
        for f in feats:
            bn, c, h, w = f.shape
            f = f.float()
            if self.out_device is not None:
                f = f.to(self.out_device)
            if self.channels_last:
                f = f.contiguous(memory_format=torch.channels_last)       # (B*N, C, H, W) with strides (HWC, 1, WC, C): no-op if it already is
                out.append(f.view(b, bn // b, c, h, w))
            else:
                out.append(f.contiguous().view(b, bn // b, c, h, w))
        return out

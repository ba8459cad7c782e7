"""Image branch: VGG16-BN convolutional trunk + global average pool -> ``[B, 512]``.

Mirrors reference ``src/models/image_net.py:6-24`` (``ImageEncoderWarpper``, spelling
kept so that the reference's entry points import it unchanged).  The reference takes the
trunk from ``torchvision.models.vgg16_bn(pretrained=True).features``; torchvision and the
network are absent here, so the trunk is defined below with the SAME ``nn.Sequential``
numbering, hence the same ``img_feature_extractor.<n>.{weight,bias,running_*}`` state-dict
keys -- a reference checkpoint, or a torchvision ``vgg16_bn`` state
dict given by local path, loads key for key.  The dense 3x3 convolutions are plain GEMM
work and run on MFMA through MIOpen; no hand kernel.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import winograd
from .fused_bn import (_conv_without_bias, bn_act, bn_act_of, bn_act_pool_of, conv_bn_act, conv_bn_act_pool,
                       foldable_into, fusable_conv, wants_conv_stats)

# VGG-16 ("configuration D"): channel widths, 'P' = 2x2 max-pool
_VGG16_PLAN = (64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512, 512, "P", 512, 512,
               512, "P")


def vgg16_bn_trunk() -> nn.Sequential:
    """conv3x3 -> BN -> ReLU triples and max-pools, numbered like torchvision's
    ``vgg16_bn().features`` (44 entries)."""
    layers = []
    c_in = 3
    for item in _VGG16_PLAN:
        if item == "P":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        conv = nn.Conv2d(c_in, item, kernel_size=3, padding=1)
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
        nn.init.zeros_(conv.bias)
        layers += [conv, nn.BatchNorm2d(item), nn.ReLU(inplace=True)]
        c_in = item
    return nn.Sequential(*layers)


class ImageEncoderWarpper(nn.Module):
    """``forward(x[B,3,H,W]) -> [B,512]``.

    ``weights`` (extension): path of a local state dict holding either this module's keys
    or torchvision ``vgg16_bn`` keys (``features.<n>.*``); without it the trunk is randomly
    initialised (the reference downloads ImageNet weights, which cannot be done offline).
    As in the reference every trunk parameter is trainable: ``_set_finetune`` exists but is
    never called by any entry point.
    """

    def __init__(self, core: str = "vgg_16", finetune_layer: int = 0, weights: str | None = None):
        super().__init__()
        if core != "vgg_16":
            raise NotImplementedError(f"Unsupported Image Encoder Core Compoenent: {core}")
        self.finetune_layer = finetune_layer
        self.img_feature_extractor = vgg16_bn_trunk()
        self.img_feature_pool = nn.AdaptiveAvgPool2d(output_size=(1, 1))
        if weights:
            self.load_trunk(weights)

    def load_trunk(self, path: str) -> None:
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if any(k.startswith("features.") for k in sd):
            sd = {k[len("features."):]: v for k, v in sd.items() if k.startswith("features.")}
        elif any(k.startswith("img_feature_extractor.") for k in sd):
            sd = {k[len("img_feature_extractor."):]: v for k, v in sd.items()
                  if k.startswith("img_feature_extractor.")}
        self.img_feature_extractor.load_state_dict(sd)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        layers = list(self.img_feature_extractor)
        i = 0
        pending = None      # (y, conv, bn, parts): bias-free output of `conv` whose BatchNorm + ReLU the next conv applies
        while i < len(layers):
            layer = layers[i]
            triple = (isinstance(layer, nn.Conv2d) and i + 2 < len(layers) and isinstance(layers[i + 1], nn.BatchNorm2d)
                      and isinstance(layers[i + 2], nn.ReLU))
            if triple and (pending is not None or fusable_conv(layer, layers[i + 1], x)):
                bn = layers[i + 1]
                # conv (bias-free) -> K5: bias + BatchNorm + ReLU in one pass, bias gradient from its dx
                # pass; a stage's closing 2x2 max-pool joins the same pass; between two Winograd
                # convolutions of a stage the BatchNorm + ReLU apply is folded into the second one's load
                # the convolution's epilogue also sums y + bias and its square per channel: the BatchNorm that
                # follows takes its statistics from those partial sums instead of reading y again
                want = wants_conv_stats(layer, bn, x)
                parts = None
                nxt3 = layers[i + 3] if i + 5 < len(layers) else None
                if (pending is None and i == 0 and isinstance(nxt3, nn.Conv2d) and isinstance(layers[i + 4], nn.BatchNorm2d)
                        and isinstance(layers[i + 5], nn.ReLU) and fusable_conv(nxt3, layers[i + 4], x)
                        and winograd.stem_applies(x, layer, bn, nxt3)):
                    # the trunk's first two convolutions as one autograd function: conv1_1's BatchNorm backward stops
                    # after its sums, K8 forms dy itself (winograd._StemConvBNReluConv); from here on the loop is at
                    # conv1_2's output
                    i += 3
                    layer, bn = nxt3, layers[i + 1]
                    want = bn.training and winograd.stats_enabled()      # (what wants_conv_stats answers for conv1_2 here)
                    y = winograd.stem_conv_bn_relu_conv(x, layers[0], layers[1], layer, want_parts=want)
                elif pending is not None:
                    py, pconv, pbn, pparts = pending
                    y = winograd.bn_relu_conv3x3(py, pconv.bias, pbn, layer.weight, parts=pparts,
                                                 stats_bias=layer.bias, want_parts=want)
                    pending = None
                else:
                    y = _conv_without_bias(layer, x, want)
                if want:
                    y, parts = y
                nxt = layers[i + 3] if i + 3 < len(layers) else None
                next_is_triple = (isinstance(nxt, nn.Conv2d) and i + 5 < len(layers)
                                  and isinstance(layers[i + 4], nn.BatchNorm2d) and isinstance(layers[i + 5], nn.ReLU))
                if isinstance(nxt, nn.MaxPool2d):
                    x = bn_act_pool_of(y, layer, bn, nxt, "relu", parts)
                    i += 4
                elif next_is_triple and fusable_conv(nxt, layers[i + 4], y) and foldable_into(y, bn, nxt):
                    pending = (y, layer, bn, parts)
                    x = y                     # placeholder: the next iteration consumes `pending`
                    i += 3
                else:
                    x = bn_act_of(y, layer, bn, "relu")
                    i += 3
            elif triple:
                if i + 3 < len(layers) and isinstance(layers[i + 3], nn.MaxPool2d):
                    x = conv_bn_act_pool(layer, layers[i + 1], layers[i + 3], x, "relu")
                    i += 4
                else:
                    x = conv_bn_act(layer, layers[i + 1], x, "relu")
                    i += 3
            elif isinstance(layer, nn.BatchNorm2d) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU):
                x = bn_act(layer, x, "relu")     # BatchNorm + ReLU as one pass (K5) on the GPU
                i += 2
            else:
                x = layer(x)
                i += 1
        return self.img_feature_pool(x).flatten(1)

    def _set_finetune(self, new_layer: int | None = None) -> None:
        """Un-freezes the last ``finetune_layer`` convolutions, freezes the others
        (reference ``image_net.py:26-39``)."""
        if new_layer is not None:
            self.finetune_layer = new_layer
        budget = self.finetune_layer
        for layer in reversed(self.img_feature_extractor):
            if isinstance(layer, nn.Conv2d):
                layer.requires_grad_(budget > 0)
                budget -= 1 if budget > 0 else 0

"""CIFAR-style ResNets (3x3 stem, no max-pool, 4x4 average pool).

Same family as ``fllib/models/cifar10/resnet_cifar.py``: BasicBlock / Bottleneck
stacks, **every BatchNorm has ``track_running_stats=False``** — there are no
buffers, ``state_dict == parameters`` and evaluation also normalises with batch
statistics.  In the batched engine the "batch" of a BN layer is one client's
batch, i.e. statistics are per (client, channel).

Parameter names (conv1, bn1, layer{1..4}.{i}.conv1/bn1/conv2/bn2/shortcut.0/1,
linear) and their order match the reference, so D and the flattening agree:
ResNet10 D = 4,903,242, ResNet18 D = 11,173,962.
"""
from typing import Sequence

import torch.nn.functional as F
from torch import nn


def _bn(c):
    return nn.BatchNorm2d(c, track_running_stats=False)


def _shortcut(cin, cout, stride):
    if stride == 1 and cin == cout:
        return nn.Sequential()
    return nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), _bn(cout))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = _bn(planes)
        self.shortcut = _shortcut(in_planes, planes * self.expansion, stride)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + self.shortcut(x))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = _bn(planes * self.expansion)
        self.shortcut = _shortcut(in_planes, planes * self.expansion, stride)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + self.shortcut(x))


class ResNet(nn.Module):
    def __init__(self, block, num_blocks: Sequence[int], num_classes: int = 10, in_channels: int = 3):
        super().__init__()
        self.in_planes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 3, stride=1, padding=1, bias=False)
        self.bn1 = _bn(64)
        self.layer1 = self._stage(block, 64, num_blocks[0], 1)
        self.layer2 = self._stage(block, 128, num_blocks[1], 2)
        self.layer3 = self._stage(block, 256, num_blocks[2], 2)
        self.layer4 = self._stage(block, 512, num_blocks[3], 2)
        self.linear = nn.Linear(512 * block.expansion, num_classes)

    def _stage(self, block, planes, n, stride):
        blocks = []
        for s in [stride] + [1] * (n - 1):
            blocks.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*blocks)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        out = F.avg_pool2d(out, 4)
        return self.linear(out.reshape(out.size(0), -1))


def ResNet10(**kw):
    return ResNet(BasicBlock, [1, 1, 1, 1], **kw)


def ResNet18(**kw):
    return ResNet(BasicBlock, [2, 2, 2, 2], **kw)


def ResNet34(**kw):
    return ResNet(BasicBlock, [3, 4, 6, 3], **kw)


def ResNet50(**kw):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kw)


def ResNet101(**kw):
    return ResNet(Bottleneck, [3, 4, 23, 3], **kw)


def ResNet152(**kw):
    return ResNet(Bottleneck, [3, 8, 36, 3], **kw)

"""HRNet backbone with the reference's module tree (models/hrnet.py:10-261 of tomguluson92/SCAT):
``HRNet(c, nof_joints, bn_momentum)``, stem (two stride-2 3x3 convs), ``layer1`` (4 Bottlenecks),
``transition1..3``, ``stage2..4`` of ``StageModule`` (4 BasicBlocks per branch + fuse layers with
1x1+BN+nearest-upsample / strided 3x3 chains), ``final_layer`` 1x1 with bias — identical
``state_dict`` keys.  Every conv / BN / ReLU / upsample / add runs on libscat_hip kernels (per-layer
autograd nodes; the fused single-node executor used for ResNet-50 is the planned next step for this
many-small-conv graph).

Quirk kept: ``BasicBlock.conv2`` is declared ``inplanes -> planes`` (hrnet.py:56), harmless because
inplanes == planes wherever it is used.
"""
from __future__ import annotations

import torch.nn as nn

from .. import nn as snn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_momentum=0.1):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.conv2 = snn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.conv3 = snn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = snn.BatchNorm2d(planes * self.expansion, momentum=bn_momentum)
        self.relu = snn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(snn.add(out, residual))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_momentum=0.1):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.relu = snn.ReLU(inplace=True)
        self.conv2 = snn.Conv2d(inplanes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(snn.add(out, residual))


class StageModule(nn.Module):
    def __init__(self, stage, output_branches, c, bn_momentum):
        super().__init__()
        self.stage = stage
        self.output_branches = output_branches
        self.branches = nn.ModuleList()
        for i in range(stage):
            w = c * (2 ** i)
            self.branches.append(nn.Sequential(*[BasicBlock(w, w, bn_momentum=bn_momentum) for _ in range(4)]))
        self.fuse_layers = nn.ModuleList()
        for i in range(output_branches):
            self.fuse_layers.append(nn.ModuleList())
            for j in range(stage):
                if i == j:
                    self.fuse_layers[-1].append(nn.Sequential())
                elif i < j:
                    self.fuse_layers[-1].append(nn.Sequential(
                        snn.Conv2d(c * (2 ** j), c * (2 ** i), kernel_size=(1, 1), stride=(1, 1), bias=False),
                        snn.BatchNorm2d(c * (2 ** i), eps=1e-05, momentum=0.1),
                        snn.Upsample(scale_factor=(2.0 ** (j - i)), mode="nearest")))
                else:
                    chain = []
                    for _ in range(i - j - 1):
                        chain.append(nn.Sequential(
                            snn.Conv2d(c * (2 ** j), c * (2 ** j), kernel_size=(3, 3), stride=(2, 2), padding=(1, 1),
                                       bias=False),
                            snn.BatchNorm2d(c * (2 ** j), eps=1e-05, momentum=0.1),
                            snn.ReLU(inplace=True)))
                    chain.append(nn.Sequential(
                        snn.Conv2d(c * (2 ** j), c * (2 ** i), kernel_size=(3, 3), stride=(2, 2), padding=(1, 1),
                                   bias=False),
                        snn.BatchNorm2d(c * (2 ** i), eps=1e-05, momentum=0.1)))
                    self.fuse_layers[-1].append(nn.Sequential(*chain))
        self.relu = snn.ReLU(inplace=True)

    def forward(self, x):
        assert len(self.branches) == len(x)
        x = [branch(b) for branch, b in zip(self.branches, x)]
        fused = []
        for i in range(len(self.fuse_layers)):
            acc = self.fuse_layers[i][0](x[0])
            for j in range(1, len(self.branches)):
                acc = snn.add(acc, self.fuse_layers[i][j](x[j]))
            fused.append(self.relu(acc))
        return fused


class HRNet(nn.Module):
    def __init__(self, c=48, nof_joints=17, bn_momentum=0.1):
        super().__init__()
        m = bn_momentum
        self.conv1 = snn.Conv2d(3, 64, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1), bias=False)
        self.bn1 = snn.BatchNorm2d(64, eps=1e-05, momentum=m)
        self.conv2 = snn.Conv2d(64, 64, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1), bias=False)
        self.bn2 = snn.BatchNorm2d(64, eps=1e-05, momentum=m)
        self.relu = snn.ReLU(inplace=True)
        downsample = nn.Sequential(snn.Conv2d(64, 256, kernel_size=(1, 1), stride=(1, 1), bias=False),
                                   snn.BatchNorm2d(256, eps=1e-05, momentum=m))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, downsample=downsample), Bottleneck(256, 64),
                                    Bottleneck(256, 64), Bottleneck(256, 64))

        def cbr(cin, cout, stride):
            return nn.Sequential(snn.Conv2d(cin, cout, kernel_size=(3, 3), stride=(stride, stride), padding=(1, 1),
                                            bias=False),
                                 snn.BatchNorm2d(cout, eps=1e-05, momentum=m), snn.ReLU(inplace=True))

        self.transition1 = nn.ModuleList([cbr(256, c, 1), nn.Sequential(cbr(256, c * 2, 2))])
        self.stage2 = nn.Sequential(StageModule(2, 2, c, m))
        self.transition2 = nn.ModuleList([nn.Sequential(), nn.Sequential(), nn.Sequential(cbr(c * 2, c * 4, 2))])
        self.stage3 = nn.Sequential(*[StageModule(3, 3, c, m) for _ in range(4)])
        self.transition3 = nn.ModuleList([nn.Sequential(), nn.Sequential(), nn.Sequential(),
                                          nn.Sequential(cbr(c * 4, c * 8, 2))])
        self.stage4 = nn.Sequential(StageModule(4, 4, c, m), StageModule(4, 4, c, m), StageModule(4, 1, c, m))
        self.final_layer = snn.Conv2d(c, nof_joints, kernel_size=(1, 1), stride=(1, 1))

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        x = [trans(x) for trans in self.transition1]
        x = self.stage2(x)
        x = [self.transition2[0](x[0]), self.transition2[1](x[1]), self.transition2[2](x[-1])]
        x = self.stage3(x)
        x = [self.transition3[0](x[0]), self.transition3[1](x[1]), self.transition3[2](x[2]),
             self.transition3[3](x[-1])]
        x = self.stage4(x)
        return self.final_layer(x[0])

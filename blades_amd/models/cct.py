"""CIFAR-10 CCT wrapper (reference ``fllib/models/cifar10/cct.py`` — unused there)."""
from torch import nn

from .cctnets import cct_2_3x2_32


class CCTNet10(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = cct_2_3x2_32()

    def forward(self, x):
        return self.net(x)

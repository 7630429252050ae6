"""Fashion-MNIST CNN (D = 1,475,146; fc1 holds 94 % of it).

Structure and parameter names follow ``fllib/models/fashionmnist/cnn.py``:
conv(1->32,k3,p1)+ReLU+pool2, conv(32->64,k3)+ReLU+pool2, 2304->600,
Dropout2d(0.25), 600->120, 120->10 with **no activation between the FCs**.
"""
import torch
from torch import nn


class FashionCNN(nn.Module):
    def __init__(self, num_classes: int = 10):
        super().__init__()
        self.layer1 = nn.Sequential(nn.Conv2d(1, 32, kernel_size=3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2))
        self.layer2 = nn.Sequential(nn.Conv2d(32, 64, kernel_size=3), nn.ReLU(), nn.MaxPool2d(2))
        self.fc1 = nn.Linear(64 * 6 * 6, 600)
        self.drop = nn.Dropout2d(0.25)
        self.fc2 = nn.Linear(600, 120)
        self.fc3 = nn.Linear(120, num_classes)

    def forward(self, x):
        if x.dim() == 3:
            x = x.unsqueeze(1)
        out = self.layer2(self.layer1(x))
        out = out.reshape(out.size(0), -1)
        out = self.drop(self.fc1(out))
        return self.fc3(self.fc2(out))

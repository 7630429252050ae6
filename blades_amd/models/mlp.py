"""MNIST MLP: 784-128-256-10, ReLU, Dropout(0.2) (D = 136,074).

Parameter names/order match ``fllib/models/mnist/mlp.py`` (fc1, fc2, fc3) so
flattened update vectors line up with the reference's.
"""
from torch import nn

from blades_amd.ops.activations import relu


class MLP(nn.Module):
    def __init__(self, in_features: int = 28 * 28, hidden=(128, 256), num_classes: int = 10, p_drop: float = 0.2):
        super().__init__()
        self.in_features = in_features
        self.fc1 = nn.Linear(in_features, hidden[0])
        self.fc2 = nn.Linear(hidden[0], hidden[1])
        self.fc3 = nn.Linear(hidden[1], num_classes)
        self.dropout = nn.Dropout(p_drop)

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)  # one row per sample (also lets a grouped executor fold clients into the row)
        x = self.dropout(relu(self.fc1(x)))
        x = self.dropout(relu(self.fc2(x)))
        return self.fc3(x)

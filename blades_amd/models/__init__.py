from .catalog import ModelCatalog
from .mlp import MLP
from .cnn import FashionCNN
from .resnet import (ResNet, BasicBlock, Bottleneck, ResNet10, ResNet18, ResNet34, ResNet50, ResNet101,
                     ResNet152)

__all__ = ["ModelCatalog", "MLP", "FashionCNN", "ResNet", "BasicBlock", "Bottleneck", "ResNet10", "ResNet18",
           "ResNet34", "ResNet50", "ResNet101", "ResNet152"]

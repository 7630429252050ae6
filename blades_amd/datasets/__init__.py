from .fldataset import FLDataset, DeviceStore, CustomTensorDataset
from .catalog import DatasetCatalog, make_dataset, CIFAR10, MNIST, FASHIONMNIST

__all__ = ["FLDataset", "DeviceStore", "CustomTensorDataset", "DatasetCatalog", "make_dataset", "CIFAR10",
           "MNIST", "FASHIONMNIST"]

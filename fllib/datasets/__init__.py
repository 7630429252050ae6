from blades_amd.datasets import CIFAR10, FASHIONMNIST, MNIST, DatasetCatalog, FLDataset, make_dataset  # noqa: F401

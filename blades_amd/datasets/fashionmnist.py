"""Fashion-MNIST (reference ``fllib/datasets/fashionmnist.py``)."""
from .mnist import MNIST


class FASHIONMNIST(MNIST):
    _name = "fashionmnist"

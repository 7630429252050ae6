from .task import Task, TaskSpec, ce_sum_and_topk
from .mnist import MNIST, Classification

__all__ = ["Task", "TaskSpec", "MNIST", "Classification", "ce_sum_and_topk"]

from blades_amd.tasks import MNIST, Task, TaskSpec  # noqa: F401

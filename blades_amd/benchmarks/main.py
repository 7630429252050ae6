"""Centralised CIFAR-10 training of the catalogue ResNet — the sanity baseline the reference ships as
``blades/benchmarks/main.py:1-95`` (SGD lr 0.1, momentum 0.9, wd 5e-4, StepLR 50/0.1, batch 128,
random-crop(pad 4) + flip + normalise), restated for one MI355X:

* the whole dataset is resident in HBM as uint8 (150 MB); there is no DataLoader and no host worker —
  a batch is a device gather, and crop/flip/normalise run on the GPU on the batch that was just gathered;
* bf16 autocast + channels-last, fp32 master weights;
* ``WORLD_SIZE > 1`` (``python -m blades_amd.parallel.launcher --nproc 8 -- -m blades_amd.benchmarks.main``)
  trains data-parallel: gradients are flattened into one fp32 bucket and all-reduced over RCCL once per step.

It reports epoch loss, samples/s and the final test accuracy.  Without the CIFAR-10 binary files under
``--data-root`` it needs ``--synthetic`` (or ``BLADES_SYNTHETIC_DATA=1``).
"""
import argparse
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

MEAN = (0.4914, 0.4822, 0.4465)
STD = (0.2023, 0.1994, 0.2010)


def random_crop_flip(x_u8: torch.Tensor, gen: torch.Generator, pad: int = 4) -> torch.Tensor:
    """uint8 [N,3,32,32] -> float [N,3,32,32]: per-sample random crop of the 4-pixel zero-padded image and
    horizontal flip, as one gather (no per-sample loop)."""
    n, _, h, w = x_u8.shape
    dev = x_u8.device
    xp = F.pad(x_u8, (pad, pad, pad, pad))
    oy = torch.randint(0, 2 * pad + 1, (n,), generator=gen, device=dev)
    ox = torch.randint(0, 2 * pad + 1, (n,), generator=gen, device=dev)
    flip = torch.rand(n, generator=gen, device=dev) < 0.5
    ar = torch.arange(w, device=dev)
    cols = torch.where(flip[:, None], (w - 1 - ar)[None, :], ar[None, :]) + ox[:, None]  # [N, W]
    rows = torch.arange(h, device=dev)[None, :] + oy[:, None]  # [N, H]
    idx = (rows[:, :, None] * (w + 2 * pad) + cols[:, None, :]).reshape(n, 1, h * w).expand(n, 3, h * w)
    out = xp.reshape(n, 3, -1).gather(2, idx).reshape(n, 3, h, w)
    return out.float().div_(255.0)


def normalise(x: torch.Tensor) -> torch.Tensor:
    m = torch.tensor(MEAN, device=x.device).view(1, 3, 1, 1)
    s = torch.tensor(STD, device=x.device).view(1, 3, 1, 1)
    return (x - m) / s


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", default="resnet")
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--weight-decay", type=float, default=5e-4)
    ap.add_argument("--step-size", type=int, default=50)
    ap.add_argument("--gamma", type=float, default=0.1)
    ap.add_argument("--data-root", default="./data")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--n-train", type=int, default=None, help="synthetic only")
    ap.add_argument("--n-test", type=int, default=None, help="synthetic only")
    ap.add_argument("--amp", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args(argv)

    from blades_amd.datasets import raw
    from blades_amd.models.catalog import ModelCatalog
    from blades_amd.parallel.runtime import init_runtime, shutdown_runtime

    rt = init_runtime(use_gpu=not a.cpu)
    dev, world, rank = rt.device, rt.world_size, rt.rank
    torch.manual_seed(a.seed)
    (xtr, ytr), (xte, yte), synth = raw.load("cifar10", a.data_root, True if a.synthetic else None, a.seed,
                                              a.n_train, a.n_test)
    if xtr.dtype != np.uint8:  # the synthetic stand-in is float in [0, 1]
        xtr, xte = (np.clip(xtr, 0, 1) * 255).astype(np.uint8), (np.clip(xte, 0, 1) * 255).astype(np.uint8)
    Xtr, Ytr = torch.from_numpy(xtr).to(dev), torch.from_numpy(ytr).to(dev)
    Xte, Yte = torch.from_numpy(xte).to(dev), torch.from_numpy(yte).to(dev)
    model = ModelCatalog.get_model(a.model).to(dev)
    if dev.type == "cuda":
        model = model.to(memory_format=torch.channels_last)
        torch.backends.cudnn.benchmark = True
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=a.lr, momentum=a.momentum, weight_decay=a.weight_decay)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=a.step_size, gamma=a.gamma)
    amp = torch.bfloat16 if a.amp == "bf16" else None
    gen = torch.Generator(device=dev)
    gen.manual_seed(a.seed * 1000 + rank)
    perm_gen = torch.Generator(device=dev)
    n = Xtr.shape[0]
    per_rank_bs = max(1, a.batch_size // world)
    steps = n // (per_rank_bs * world)
    bucket = torch.zeros(sum(p.numel() for p in params), device=dev)

    def say(msg):
        if rt.is_main:
            print(msg, flush=True)

    say(f"centralised {a.model} on {'synthetic' if synth else 'real'} CIFAR-10: n={n}, {steps} steps/epoch, "
        f"world={world}, dtype={a.amp}")
    for epoch in range(a.epochs):
        model.train()
        perm_gen.manual_seed(a.seed * 7919 + epoch)  # same permutation on every rank; each takes its stride
        perm = torch.randperm(n, generator=perm_gen, device=dev)
        loss_acc = torch.zeros((), device=dev)
        t0 = time.perf_counter()
        for s in range(steps):
            idx = perm[(s * world + rank) * per_rank_bs: (s * world + rank + 1) * per_rank_bs]
            x = normalise(random_crop_flip(Xtr[idx], gen))
            if dev.type == "cuda":
                x = x.contiguous(memory_format=torch.channels_last)
            with torch.autocast(dev.type, dtype=amp, enabled=amp is not None):
                loss = F.cross_entropy(model(x).float(), Ytr[idx])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if world > 1:
                torch.cat([p.grad.reshape(-1) for p in params], out=bucket)
                dist.all_reduce(bucket)
                bucket.div_(world)
                off = 0
                for p in params:
                    p.grad.copy_(bucket[off: off + p.numel()].view_as(p))
                    off += p.numel()
            opt.step()
            loss_acc += loss.detach()
        sched.step()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        say(f"Epoch [{epoch + 1}/{a.epochs}] Loss: {float(loss_acc) / max(steps, 1):.4f}  "
            f"({steps * per_rank_bs * world / dt:.0f} samples/s)")
    model.eval()
    correct = 0
    with torch.no_grad():
        for i in range(0, Xte.shape[0], 1024):
            x = normalise(Xte[i: i + 1024].float() / 255.0)
            if dev.type == "cuda":
                x = x.contiguous(memory_format=torch.channels_last)
            with torch.autocast(dev.type, dtype=amp, enabled=amp is not None):
                correct += int((model(x).argmax(1) == Yte[i: i + 1024]).sum())
    acc = 100.0 * correct / Xte.shape[0]
    say(f"Accuracy on test set: {acc:.2f}%")
    shutdown_runtime()
    return acc


if __name__ == "__main__":
    main()

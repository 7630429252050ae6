"""The three execution paths (fused shared-weight, generic vectorised,
sequential reference semantics) must produce the same update matrix."""
import numpy as np
import pytest
import torch

from blades_amd.algorithms.fedavg import FedavgConfig


def build(data_root, model, mode, dataset="cifar10", nb=1, momentum=0.0, adversary=None, f=0, bs=8,
          callbacks=None, lr=0.5, K=6, seed=3):
    cc = {"lr": lr, "momentum": momentum, "num_batch_per_round": nb}
    if callbacks:
        cc["callbacks_config"] = callbacks
    cfg = (FedavgConfig()
           .data(dataset_config={"type": dataset, "num_clients": K, "num_classes": 10, "train_batch_size": bs,
                                 "n_train": K * bs * 2, "n_test": K * 4, "data_root": data_root, "use_cache": False,
                                 "synthetic": True})
           .training(global_model=model, server_config={"aggregator": {"type": "Mean"}, "optimizer": {"lr": 1.0}})
           .client(client_config=cc)
           .adversary(num_malicious_clients=f, adversary_config=adversary or {})
           .evaluation(evaluation_interval=0)
           .engine(execution_mode=mode, clients_per_chunk=4, amp_dtype=None, device="cpu"))
    cfg.seed = seed
    return cfg.build()


def one_round_updates(algo):
    # identical batches for every path: freeze augmentation and use the same sampler seed
    algo._dataset.train_transform = None
    algo.train()
    return algo.update_matrix[:, : algo.engine.D].clone(), algo.server.theta[: algo.engine.D].clone()


def _no_random_drops(net):
    from blades_amd.models.cctnets.layers import DropPath

    for m in net.modules():
        if isinstance(m, (torch.nn.modules.dropout._DropoutNd, DropPath)):
            m.p = 0.0


@pytest.mark.parametrize("model", ["resnet", "cnn", "mlp", "cct"])
def test_fused_equals_vectorised(data_root, model):
    ds = "cifar10" if model in ("resnet", "cct") else "fashionmnist"
    a = build(data_root, model, "fused", ds)
    b = build(data_root, model, "vmap", ds)
    assert a.engine.fused is not None and b.engine.fused is None
    for algo in (a, b):
        algo.engine.task.model.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.modules.dropout._DropoutNd) else None)
        algo.engine.model.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.modules.dropout._DropoutNd) else None)
        if algo.engine.fused is not None:
            algo.engine.fused.template.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.modules.dropout._DropoutNd) else None)
        for net in (algo.engine.task.model, algo.engine.model,
                    algo.engine.fused.template if algo.engine.fused is not None else None):
            if net is not None:
                _no_random_drops(net)
    Ua, ta = one_round_updates(a)
    Ub, tb = one_round_updates(b)
    torch.testing.assert_close(Ua, Ub, rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(ta, tb, rtol=1e-3, atol=2e-5)


def test_fused_signflip_labelflip_clip_momentum(data_root):
    for adv in ("SignFlipAdversary", "LabelFlipAdversary"):
        a = build(data_root, "resnet", "fused", f=2, adversary={"type": f"blades.adversaries.{adv}"})
        b = build(data_root, "resnet", "vmap", f=2, adversary={"type": f"blades.adversaries.{adv}"})
        Ua, _ = one_round_updates(a)
        Ub, _ = one_round_updates(b)
        torch.testing.assert_close(Ua, Ub, rtol=1e-3, atol=2e-5)
    cb = {"type": "blades.clients.ClippingCallback", "clip_threshold": 0.05}
    a = build(data_root, "resnet", "fused", callbacks=cb, momentum=0.9)
    b = build(data_root, "resnet", "vmap", callbacks=cb, momentum=0.9)
    for _ in range(2):
        Ua, _ = one_round_updates(a)
        Ub, _ = one_round_updates(b)
        torch.testing.assert_close(Ua, Ub, rtol=2e-3, atol=2e-5)
    assert float(Ua.norm(dim=1).max()) <= 0.5 * 0.05 * 1.9 + 1e-4  # lr * clip * (1 + momentum)


@pytest.mark.parametrize("model,expect_grouped", [("resnet", True), ("mlp", True), ("cnn", True)])
def test_grouped_local_steps_equal_vectorised(data_root, model, expect_grouped):
    """Several local steps: the grouped wide-network executor (client-innermost layout, grouped conv + plain
    BatchNorm) must reproduce the torch.vmap path for every catalogue CNN/MLP/ResNet."""
    ds = "cifar10" if model == "resnet" else "fashionmnist"
    a = build(data_root, model, "fused", ds, nb=3, momentum=0.9, lr=0.02)
    b = build(data_root, model, "vmap", ds, nb=3, momentum=0.9, lr=0.02)
    assert (a.engine.grouped is not None) == expect_grouped, getattr(a.engine, 'grouped_check', None)
    assert b.engine.grouped is None
    if not expect_grouped:
        return
    for algo in (a, b):
        for net in (algo.engine.task.model, algo.engine.model, algo.engine.grouped.template if algo.engine.grouped else None):
            if net is not None:
                net.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.modules.dropout._DropoutNd) else None)
    # The two paths round differently (one wide grouped kernel vs per-client kernels) and a ReLU sitting on zero
    # can flip: per-step gradients agree to ~1e-6 with occasional 1e-3 outliers, which training at random init
    # amplifies round over round.  So: direction and size per client, tight in round 1, loose in round 2.
    for rnd, (tol_rel, tol_cos) in enumerate([(5e-3, 0.9999), (1e-1, 0.99)]):
        Ua, ta = one_round_updates(a)
        Ub, tb = one_round_updates(b)
        cos = torch.nn.functional.cosine_similarity(Ua, Ub, dim=1)
        rel = (Ua - Ub).norm(dim=1) / Ub.norm(dim=1)
        assert float(cos.min()) > tol_cos, (rnd, cos)
        assert float(rel.max()) < tol_rel, (rnd, rel)


def test_dgrad_as_forward_conv_matches_autograd():
    """engine/fused.py computes the data gradient of stride-1 "same" convolutions as a forward convolution with
    the flipped, transposed filter: must equal autograd's."""
    import torch
    from blades_amd.engine.fused import dgrad_same_conv

    g = torch.Generator().manual_seed(0)
    for cin, cout, k in ((3, 5, 3), (4, 4, 1), (2, 6, 5)):
        x = torch.randn(2, cin, 9, 7, generator=g, requires_grad=True)
        w = torch.randn(cout, cin, k, k, generator=g)
        y = torch.nn.functional.conv2d(x, w, None, 1, k // 2)
        gy = torch.randn(y.shape, generator=g)
        (ref,) = torch.autograd.grad(y, x, gy)
        torch.testing.assert_close(dgrad_same_conv(gy, w, (k // 2, k // 2)), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode", ["fused", "vmap"])
def test_local_steps_do_not_depend_on_chunking(data_root, mode):
    """Updates of the multi-step path must not depend on how clients are cut into chunks — in particular a chunk
    holding a single client (regression: its weight replica used to alias the global weights)."""
    def run(chunk):
        a = build(data_root, "mlp", mode, "fashionmnist", nb=3, K=5)
        a.engine.chunk = chunk
        for net in (a.engine.task.model, a.engine.model, a.engine.grouped.template if a.engine.grouped else None):
            if net is not None:
                _no_random_drops(net)
        theta0 = a.server.theta.clone()
        U, _ = one_round_updates(a)
        return U, theta0

    U3, t3 = run(3)
    U2, t2 = run(2)   # chunks of 2, 2 and 1 clients
    U1, _ = run(1)
    torch.testing.assert_close(t3, t2)
    torch.testing.assert_close(U2, U3, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(U1, U3, rtol=1e-5, atol=1e-6)
    assert float(U1.abs().sum(1).min()) > 0


@pytest.mark.parametrize("model,ds", [("mlp", "fashionmnist"), ("cnn", "fashionmnist")])
def test_client_keyed_dropout_is_active_and_path_independent(data_root, model, ds):
    """Dropout stays on in the engine templates (ClientDropout, keyed by seed / client / round / layer): the fused,
    vmap and — for several local steps — grouped executors see the same masks, and the masks matter."""
    from blades_amd.engine.layers import ClientDropout

    a = build(data_root, model, "fused", ds)
    b = build(data_root, model, "vmap", ds)
    drops = [m for m in a.engine.fused.template.modules() if isinstance(m, ClientDropout)]
    assert drops and all(m.p > 0 for m in drops)
    Ua, _ = one_round_updates(a)
    Ub, _ = one_round_updates(b)
    torch.testing.assert_close(Ua, Ub, rtol=1e-3, atol=2e-5)
    c = build(data_root, model, "fused", ds)
    c.engine.drop_ctx.enabled = False
    Uc, _ = one_round_updates(c)
    assert float((Ua - Uc).abs().max()) > 1e-4  # dropping really changed the gradients
    # several local steps: grouped wide-network executor vs vmap, first round (later rounds amplify rounding)
    g1 = build(data_root, model, "fused", ds, nb=2, lr=0.02)
    g2 = build(data_root, model, "vmap", ds, nb=2, lr=0.02)
    assert g1.engine.grouped is not None and g2.engine.grouped is None
    U1, _ = one_round_updates(g1)
    U2, _ = one_round_updates(g2)
    rel = (U1 - U2).norm(dim=1) / U2.norm(dim=1)
    assert float(rel.max()) < 5e-3, rel


def _sine_cct():
    from blades_amd.models import cctnets

    return cctnets.cct_2_3x2_32(positional_embedding="sine")


@pytest.mark.parametrize("nb", [1, 2])
def test_frozen_parameters_take_no_update_columns(data_root, nb):
    """A frozen parameter (the sinusoidal positional embedding) is a constant of the model: server, engine (fused and
    vmap), and the sequential client path all flatten the trainable parameters only, so widths and offsets agree
    (the reference's client emits columns for frozen parameters that its server then skips, fllib/tasks/task.py:162
    vs fllib/algorithms/server.py:122)."""
    algos = {}
    for mode in ("fused", "vmap", "sequential"):
        torch.manual_seed(0)
        a = build(data_root, _sine_cct(), mode, nb=nb, lr=0.05)
        for net in (a.engine.task.model, a.engine.model, a.engine.fused.template if a.engine.fused else None):
            if net is not None:
                _no_random_drops(net)
                net.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.modules.dropout._DropoutNd) else None)
        algos[mode] = a
    a = algos["fused"]
    frozen = [n for n, p in a.engine.task.model.named_parameters() if not p.requires_grad]
    assert frozen == ["classifier.positional_emb"]
    n_train = sum(p.numel() for p in a.engine.task.model.parameters() if p.requires_grad)
    assert a.engine.D == a.server.D == n_train
    assert a.engine.fused is not None
    pe0 = a.engine.task.model.classifier.positional_emb.clone()
    res = {m: one_round_updates(al) for m, al in algos.items()}
    torch.testing.assert_close(res["fused"][0], res["vmap"][0], rtol=2e-3, atol=2e-5)
    if nb == 1:  # sequential clients draw their batches from a DataLoader-like reader with the same sampler
        assert res["sequential"][0].shape == res["vmap"][0].shape
    assert torch.equal(a.engine.task.model.classifier.positional_emb, pe0)


def test_stem_im2col_torch_path_matches_convolution():
    """The engine's im2col of a first-layer input (torch branch; the GPU branch is ops/csrc/im2col.hip): a 1x1
    convolution over it with ``weight.reshape(Co, Ci*9)`` padded to 32 channels is the 3x3 convolution."""
    import torch.nn.functional as F
    from blades_amd.engine.fused import _stem_im2col

    torch.manual_seed(0)
    for Ci in (1, 3):
        x = torch.randn(5, Ci, 9, 7)
        w = torch.randn(6, Ci, 3, 3)
        col = _stem_im2col(x, 3, 1)
        assert col.shape == (5, 32, 9, 7)
        w2 = F.pad(w.reshape(6, Ci * 9), (0, 32 - Ci * 9)).reshape(6, 32, 1, 1)
        torch.testing.assert_close(F.conv2d(col, w2), F.conv2d(x, w, None, 1, 1), rtol=1e-5, atol=1e-5)
        assert float(col[:, Ci * 9:].abs().max()) == 0

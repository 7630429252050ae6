"""Models, datasets, catalogs, config objects."""
import numpy as np
import pytest
import torch

from blades_amd.algorithms.fedavg import FedavgConfig
from blades_amd.datasets import DatasetCatalog, FLDataset, make_dataset
from blades_amd.models import ModelCatalog


def nparams(m):
    return sum(p.numel() for p in m.parameters())


def test_model_catalog_dispatch_and_sizes():
    from blades_amd.models import FashionCNN, MLP, ResNet

    assert nparams(ModelCatalog.get_model("mlp")) == 136074
    assert nparams(ModelCatalog.get_model("cnn")) == 1475146
    r = ModelCatalog.get_model("resnet")  # the trap: "resnet" is ResNet-10
    assert isinstance(r, ResNet) and nparams(r) == 4903242 and not list(r.buffers())
    assert nparams(ModelCatalog.get_model("resnet18")) == 11173962
    assert nparams(ModelCatalog.get_model("cct")) == 283723
    inst = torch.nn.Linear(3, 2)
    assert ModelCatalog.get_model(inst) is inst

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(2, 2)

    ModelCatalog.register_custom_model("mynet", Net)
    assert isinstance(ModelCatalog.get_model({"custom_model": "mynet"}), Net)
    with pytest.raises(NotImplementedError):
        ModelCatalog.get_model("nope")


@pytest.mark.parametrize("name,shape", [("resnet", (3, 32, 32)), ("cnn", (28, 28)), ("mlp", (28, 28)),
                                        ("cct", (3, 32, 32)), ("cvt_2_4_32", (3, 32, 32)), ("resnet50", (3, 32, 32))])
def test_models_forward_backward(name, shape):
    m = ModelCatalog.get_model(name)
    out = m(torch.randn(4, *shape))
    assert out.shape == (4, 10)
    out.sum().backward()
    assert all(p.grad is not None for p in m.parameters() if p.requires_grad)


def test_dataset_partition_split_subset_pickle(data_root):
    import pickle

    ds = make_dataset("mnist", 10, train_batch_size=16, seed=3, n_train=1000, n_test=200, data_root=data_root,
                      synthetic=True)
    assert ds.client_ids == [str(i) for i in range(10)]
    assert sum(ds.train_size(i) for i in ds.client_ids) == 1000
    sub = ds.subset(["1", "3"])
    assert sub.client_ids == ["1", "3"]
    for n in (2, 5):
        parts = ds.split(n)
        assert sorted(i for p in parts for i in p.client_ids) == sorted(ds.client_ids)
        assert max(len(p.client_ids) for p in parts) - min(len(p.client_ids) for p in parts) <= 1
    x, y = next(ds.get_train_loader("0"))
    assert x.shape == (16, 28, 28) and y.dtype == torch.long and float(x.max()) <= 1.0
    back = pickle.loads(pickle.dumps(sub))
    assert isinstance(back, FLDataset) and back.client_ids == ["1", "3"]
    # IID = unshuffled array_split; test shards even and unshuffled
    tot = [ds.test_size(i) for i in ds.client_ids]
    assert sum(tot) == 200 and max(tot) - min(tot) <= 1


def test_dataset_cache_and_noniid(data_root):
    kw = dict(train_batch_size=8, seed=5, n_train=1500, n_test=100, data_root=data_root, synthetic=True)
    a = make_dataset("fashionmnist", 8, iid=False, alpha=0.3, **kw)
    sizes = [a.train_size(i) for i in a.client_ids]
    assert min(sizes) >= 10 and sum(sizes) == 1500 and len(set(sizes)) > 2  # Dirichlet => unequal
    b = make_dataset("fashionmnist", 8, iid=False, alpha=0.3, **kw)  # served from the cache
    assert [b.train_size(i) for i in b.client_ids] == sizes
    hist = np.stack([np.bincount(np.asarray(a.train_data[i]["y"]), minlength=10) for i in a.client_ids])
    assert (hist.max(1) / hist.sum(1)).mean() > 0.25  # label-skewed


def test_cifar_transform_is_batch_level(data_root):
    ds = make_dataset("cifar10", 4, train_batch_size=8, seed=1, n_train=64, n_test=16, data_root=data_root,
                      synthetic=True)
    x, y = next(ds.get_train_loader("0"))
    assert x.shape == (8, 3, 32, 32)
    assert abs(float(x.mean())) < 3.0  # normalised
    tl = list(ds.get_test_loader("1"))
    assert tl[0][0].shape[1:] == (3, 32, 32)


def test_missing_real_data_raises_clearly(tmp_path, monkeypatch):
    monkeypatch.setenv("BLADES_SYNTHETIC_DATA", "0")
    with pytest.raises(FileNotFoundError):
        make_dataset("cifar10", 4, data_root=str(tmp_path), synthetic=None)


def test_custom_dataset_registration():
    from .simple_fixtures import SimpleDataset  # registers "simple"

    ds = DatasetCatalog.get_dataset({"custom_dataset": "simple", "num_clients": 4})
    assert isinstance(ds, SimpleDataset) and len(ds.client_ids) == 4
    ds2 = DatasetCatalog.get_dataset({"custom_dataset": SimpleDataset, "num_clients": 2})
    assert len(ds2.client_ids) == 2


def test_config_validation_and_injection():
    cfg = FedavgConfig().update_from_dict({
        "dataset_config": {"type": "mnist", "num_clients": 7}, "num_malicious_clients": 2,
        "server_config": {"aggregator": {"type": "Multikrum"}, "optimizer": {"lr": 0.3}}})
    cfg.validate()
    assert cfg.num_clients == 7 and cfg.dataset_config["seed"] == cfg.seed
    assert cfg.server_config["aggregator"]["num_byzantine"] == 2  # injected for DnC/Trimmedmean/Multikrum
    cfg.freeze()
    assert cfg.get_server_config().optimizer["lr"] == 0.3
    assert cfg["num_malicious_clients"] == 2 and cfg.get("nope", 5) == 5 and "seed" in cfg.keys()
    bad = FedavgConfig().update_from_dict({"dataset_config": {"type": "mnist", "num_clients": 3},
                                           "num_malicious_clients": 4})
    with pytest.raises(ValueError):
        bad.validate()
    with pytest.raises(ValueError):
        FedavgConfig().validate()  # no num_clients anywhere
    with pytest.raises(ValueError):
        FedavgConfig().get_server_config()  # not frozen


def test_lr_schedule_piecewise():
    from blades_amd.algorithms import PiecewiseSchedule

    s = PiecewiseSchedule([[0, 0.1], [1500, 0.1], [1501, 0.01], [2000, 0.01]])
    assert s.value(0) == 0.1 and s.value(1500) == pytest.approx(0.1) and s.value(1501) == pytest.approx(0.01)
    assert s.value(5000) == 0.01
    assert PiecewiseSchedule([[0, 1.0], [10, 0.0]]).value(5) == pytest.approx(0.5)


def test_client_proxy_and_callbacks():
    from blades_amd.clients import Client, ClientConfig, ClientProxy, ClippingCallback, LabelFlipCallback

    cfg = ClientConfig("blades_amd.clients.Client").update_from_dict(
        {"lr": 0.3, "callbacks_config": {"type": "blades.clients.ClippingCallback", "clip_threshold": 2.0}})
    cfg.uid = "0"
    c = Client(cfg)
    assert not c.is_malicious and c.callbacks.batched_spec() == {"grad_clip": 2.0}
    c.to_malicious(callbacks_cls=LabelFlipCallback(10), local_training=True)
    assert isinstance(c, ClientProxy) and c.is_malicious and c.client_id == "0" and c.local_training
    assert c.active_callbacks().batched_spec() == {"grad_clip": 2.0, "label_flip": 10}
    c2 = Client(cfg)
    c2.to_malicious(local_training=False)
    assert c2.train_one_round(iter([])) == {"id": "0"}

    class Py(ClippingCallback):
        def on_train_round_end(self):
            pass

    class Custom(Py):
        def batched_spec(self):
            return None

    c3 = Client(cfg)
    c3.add_callback(Custom(1.0))
    assert c3.callbacks.batched_spec() is None  # forces the sequential path for this client


def test_cct_through_generic_engine(data_root):
    from tests.test_engine_paths import build

    a = build(data_root, "cct", "auto", K=3, bs=4)
    assert a.engine.fused is not None  # shared-weight executor: per-client LayerNorm / embedding gradients
    r = a.train()
    assert np.isfinite(r["train_loss"]) and float(a.update_matrix.abs().sum()) > 0
    b = build(data_root, "cct", "vmap", K=3, bs=4)  # the generic vectorised path still takes it
    assert b.engine.fused is None
    r = b.train()
    assert np.isfinite(r["train_loss"]) and float(b.update_matrix.abs().sum()) > 0


def test_text_transformer_variants_and_checkpoint_helpers():
    """Masked (padding-aware) CCT pieces + positional-table resampling (reference: cctnets/utils/*)."""
    import torch
    from blades_amd.models import cctnets as C

    torch.manual_seed(0)
    m = C.TextCCT(seq_len=12, word_embedding_dim=16, embedding_dim=32, vocab_size=50, num_classes=3).eval()
    ids = torch.randint(2, 50, (4, 12))
    mask = torch.ones(4, 12, dtype=torch.bool)
    mask[:, 8:] = False
    out = m(ids, mask)
    assert out.shape == (4, 3) and torch.isfinite(out).all()
    # padded positions must not influence the result
    ids2 = ids.clone()
    ids2[:, 8:] = torch.randint(2, 50, (4, 4))
    torch.testing.assert_close(m(ids2, mask), out)
    out.sum().backward()
    assert all(p.grad is not None for p in m.parameters() if p.requires_grad)
    # class-token variant + learnable table
    clf = C.MaskedTransformerClassifier(seq_pool=False, embedding_dim=32, num_layers=1, num_heads=2, mlp_ratio=1.0,
                                        num_classes=5, positional_embedding="learnable", seq_len=6).eval()
    assert clf(torch.randn(2, 6, 32), torch.ones(2, 6, dtype=torch.bool)).shape == (2, 5)
    assert clf(torch.randn(2, 4, 32)).shape == (2, 5)  # short input is padded to the table
    # positional table resampling 4x4 -> 8x8 grid, fc reset on class-count change
    small, big = C.cct_2_3x2_32(img_size=16), C.cct_2_3x2_32(img_size=32, num_classes=7)
    sd = {k: v.clone() for k, v in small.state_dict().items()}
    sd = C.fc_check(big, C.pe_check(big, sd))
    assert sd["classifier.positional_emb"].shape == big.state_dict()["classifier.positional_emb"].shape
    assert sd["classifier.fc.weight"].shape[0] == 7
    big.load_state_dict(sd)

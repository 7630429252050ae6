"""CCT (conv tokenizer) and CVT (patch tokenizer) model families.

Factory names follow Compact-Transformers / the reference (``fllib/models/backbones/cctnets``):
``cct_<L>_<k>x<n>_<img>[_sine][_c100]`` — L encoder layers, n conv layers of kernel k, image size —
and ``cvt_<L>_<patch>_<img>[_sine]``.  Only ``cct_2_3x2_32`` is reachable through the reference's
model catalog (``global_model: cct``); all the others are addressable here by their full name.
No pretrained weights are downloadable offline: ``pretrained=True`` raises.
"""
from functools import partial

from torch import nn

from .layers import Attention, DropPath, Tokenizer, TransformerClassifier, TransformerEncoderLayer  # noqa: F401

# layers, heads, mlp_ratio, embedding dim
_SIZES = {2: (2, 2, 1, 128), 4: (4, 2, 1, 128), 6: (6, 4, 2, 256), 7: (7, 4, 2, 256), 8: (8, 4, 2, 256),
          14: (14, 6, 3, 384)}


class CCT(nn.Module):
    def __init__(self, img_size=224, embedding_dim=768, n_input_channels=3, n_conv_layers=1, kernel_size=7, stride=2,
                 padding=3, pooling_kernel_size=3, pooling_stride=2, pooling_padding=1, dropout=0.0,
                 attention_dropout=0.1, stochastic_depth=0.1, num_layers=14, num_heads=6, mlp_ratio=4.0,
                 num_classes=1000, positional_embedding="learnable", **_):
        super().__init__()
        self.tokenizer = Tokenizer(kernel_size, stride, padding, pooling_kernel_size, pooling_stride, pooling_padding,
                                   n_conv_layers, n_input_channels, embedding_dim, activation=nn.ReLU, max_pool=True,
                                   conv_bias=False)
        self.classifier = TransformerClassifier(
            True, embedding_dim, num_layers, num_heads, mlp_ratio, num_classes, dropout, attention_dropout,
            stochastic_depth, positional_embedding,
            self.tokenizer.sequence_length(n_input_channels, img_size, img_size))

    def forward(self, x):
        return self.classifier(self.tokenizer(x))


class CVT(nn.Module):
    def __init__(self, img_size=224, embedding_dim=768, n_input_channels=3, kernel_size=16, dropout=0.0,
                 attention_dropout=0.1, stochastic_depth=0.1, num_layers=14, num_heads=6, mlp_ratio=4.0,
                 num_classes=1000, positional_embedding="learnable", **_):
        super().__init__()
        assert img_size % kernel_size == 0, f"image size {img_size} must be divisible by patch size {kernel_size}"
        self.tokenizer = Tokenizer(kernel_size, kernel_size, 0, n_conv_layers=1, n_input_channels=n_input_channels,
                                   n_output_channels=embedding_dim, activation=None, max_pool=False, conv_bias=True)
        self.classifier = TransformerClassifier(
            True, embedding_dim, num_layers, num_heads, mlp_ratio, num_classes, dropout, attention_dropout,
            stochastic_depth, positional_embedding,
            self.tokenizer.sequence_length(n_input_channels, img_size, img_size))

    def forward(self, x):
        return self.classifier(self.tokenizer(x))


def _cct(size, kernel_size, n_conv_layers, img_size, positional_embedding, num_classes, pretrained=False,
         progress=False, **kw):
    if pretrained:
        raise RuntimeError("pretrained CCT weights cannot be downloaded in this (offline) framework")
    L, H, R, E = _SIZES[size]
    kw.setdefault("img_size", img_size)
    kw.setdefault("positional_embedding", positional_embedding)
    kw.setdefault("num_classes", num_classes)
    stride = kw.pop("stride", None)
    padding = kw.pop("padding", None)
    stride = stride if stride is not None else max(1, kernel_size // 2 - 1)
    padding = padding if padding is not None else max(1, kernel_size // 2)
    return CCT(num_layers=L, num_heads=H, mlp_ratio=R, embedding_dim=E, kernel_size=kernel_size, stride=stride,
               padding=padding, n_conv_layers=n_conv_layers, **kw)


def _cvt(size, patch, img_size, positional_embedding, num_classes, pretrained=False, progress=False, **kw):
    if pretrained:
        raise RuntimeError("pretrained CVT weights cannot be downloaded in this (offline) framework")
    L, H, R, E = _SIZES[size]
    kw.setdefault("img_size", img_size)
    kw.setdefault("positional_embedding", positional_embedding)
    kw.setdefault("num_classes", num_classes)
    return CVT(num_layers=L, num_heads=H, mlp_ratio=R, embedding_dim=E, kernel_size=patch, **kw)


def _cct_factory(size, kernel, n_conv, img, pe, classes):
    def make(pretrained=False, progress=False, img_size=img, positional_embedding=pe, num_classes=classes, **kw):
        return _cct(size, kernel, n_conv, img_size, positional_embedding, num_classes, pretrained, progress, **kw)

    return make


def _cvt_factory(size, patch, img, pe, classes):
    def make(pretrained=False, progress=False, img_size=img, positional_embedding=pe, num_classes=classes, **kw):
        return _cvt(size, patch, img_size, positional_embedding, num_classes, pretrained, progress, **kw)

    return make


_CCT_VARIANTS = [  # (size, kernel, conv layers, img, classes, suffixes)
    (2, 3, 2, 32, 10, ("", "_sine")), (4, 3, 2, 32, 10, ("", "_sine")), (6, 3, 1, 32, 10, ("", "_sine")),
    (6, 3, 2, 32, 10, ("", "_sine")), (7, 3, 1, 32, 10, ("", "_sine")), (7, 3, 2, 32, 10, ("", "_sine")),
    (7, 7, 2, 224, 102, ("", "_sine")), (14, 7, 2, 224, 1000, ("",)), (14, 7, 2, 384, 1000, ("",)),
]
__all__ = ["CCT", "CVT", "Tokenizer", "TransformerClassifier", "TransformerEncoderLayer", "Attention", "DropPath"]
for _s, _k, _n, _img, _nc, _sufs in _CCT_VARIANTS:
    for _suf in _sufs:
        _name = f"cct_{_s}_{_k}x{_n}_{_img}{_suf}"
        globals()[_name] = _cct_factory(_s, _k, _n, _img, "sine" if _suf else "learnable", _nc)
        globals()[_name].__name__ = _name
        __all__.append(_name)
cct_7_3x1_32_c100 = _cct_factory(7, 3, 1, 32, "learnable", 100)
cct_7_3x1_32_sine_c100 = _cct_factory(7, 3, 1, 32, "sine", 100)
cct_14_7x2_384_fl = _cct_factory(14, 7, 2, 384, "learnable", 102)
__all__ += ["cct_7_3x1_32_c100", "cct_7_3x1_32_sine_c100", "cct_14_7x2_384_fl"]
for _s in (2, 4, 6, 7):
    for _suf in ("", "_sine"):
        _name = f"cvt_{_s}_4_32{_suf}"
        globals()[_name] = _cvt_factory(_s, 4, 32, "sine" if _suf else "learnable", 10)
        globals()[_name].__name__ = _name
        __all__.append(_name)

from .helpers import fc_check, pe_check, resize_pos_embed  # noqa: E402,F401
from .text import (Embedder, MaskedAttention, MaskedTransformerClassifier,  # noqa: E402,F401
                   MaskedTransformerEncoderLayer, TextCCT, TextTokenizer)

__all__ += ["Embedder", "TextTokenizer", "MaskedAttention", "MaskedTransformerEncoderLayer",
            "MaskedTransformerClassifier", "TextCCT", "resize_pos_embed", "pe_check", "fc_check"]

"""Compact-Transformer building blocks (Hassani et al., "Escaping the Big Data Paradigm with
Compact Transformers").

Own implementation of the architecture the reference vendors under
``fllib/models/backbones/cctnets`` — conv tokenizer, pre-norm encoder with GELU MLP and
stochastic depth, sequence pooling head.  Sub-module / parameter names follow the upstream
layout (``tokenizer.conv_layers.N.0``, ``classifier.blocks.N.{pre_norm,self_attn.qkv,
self_attn.proj,linear1,norm1,linear2}``, ``classifier.{positional_emb,attention_pool,norm,fc}``)
so the flattened parameter order — and therefore update vectors — line up with the reference's.
"""

import torch
import torch.nn.functional as F
from torch import nn

from blades_amd.ops.activations import gelu, seq_pool


class DropPath(nn.Module):
    """Stochastic depth: drops the whole residual branch per sample."""

    def __init__(self, p: float = 0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p <= 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        mask = torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device) < keep
        return x * mask / keep


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, attention_dropout=0.1, projection_dropout=0.1):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.attn_drop = nn.Dropout(attention_dropout)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(projection_dropout)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        attn = self.attn_drop(((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1))
        return self.proj_drop(self.proj((attn @ v).transpose(1, 2).reshape(B, N, C)))


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, attention_dropout=0.1, drop_path_rate=0.1):
        super().__init__()
        self.pre_norm = nn.LayerNorm(d_model)
        self.self_attn = Attention(d_model, nhead, attention_dropout, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout2 = nn.Dropout(dropout)
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0 else nn.Identity()

    def forward(self, src):
        src = src + self.drop_path(self.self_attn(self.pre_norm(src)))
        src = self.norm1(src)
        src2 = self.linear2(self.dropout1(gelu(self.linear1(src))))
        return src + self.drop_path(self.dropout2(src2))


def sinusoidal_embedding(n_tokens: int, dim: int) -> torch.Tensor:
    pos = torch.arange(n_tokens, dtype=torch.float32).unsqueeze(1)
    i = torch.arange(dim, dtype=torch.float32).unsqueeze(0)
    angle = pos / torch.pow(10000.0, 2 * torch.div(i, 2, rounding_mode="floor") / dim)
    pe = torch.where((torch.arange(dim) % 2 == 0).unsqueeze(0), torch.sin(angle), torch.cos(angle))
    return pe.unsqueeze(0)


class Tokenizer(nn.Module):
    """Conv (+activation) (+max-pool) stack -> token sequence ``[B, H'*W', C]``."""

    def __init__(self, kernel_size, stride, padding, pooling_kernel_size=3, pooling_stride=2, pooling_padding=1,
                 n_conv_layers=1, n_input_channels=3, n_output_channels=64, in_planes=64, activation=None,
                 max_pool=True, conv_bias=False):
        super().__init__()
        chans = [n_input_channels] + [in_planes] * (n_conv_layers - 1) + [n_output_channels]
        stages = []
        for a, b in zip(chans[:-1], chans[1:]):
            stages.append(nn.Sequential(
                nn.Conv2d(a, b, (kernel_size, kernel_size), (stride, stride), (padding, padding), bias=conv_bias),
                nn.Identity() if activation is None else activation(),
                nn.MaxPool2d(pooling_kernel_size, pooling_stride, pooling_padding) if max_pool else nn.Identity()))
        self.conv_layers = nn.Sequential(*stages)
        self.flattener = nn.Flatten(2, 3)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def sequence_length(self, n_channels=3, height=224, width=224) -> int:
        with torch.no_grad():
            return self.forward(torch.zeros(1, n_channels, height, width)).shape[1]

    def forward(self, x):
        return self.flattener(self.conv_layers(x)).transpose(-2, -1)


class TransformerClassifier(nn.Module):
    def __init__(self, seq_pool=True, embedding_dim=768, num_layers=12, num_heads=12, mlp_ratio=4.0,
                 num_classes=1000, dropout=0.1, attention_dropout=0.1, stochastic_depth=0.1,
                 positional_embedding="learnable", sequence_length=None):
        super().__init__()
        if positional_embedding not in ("sine", "learnable", "none"):
            positional_embedding = "sine"
        self.embedding_dim, self.sequence_length, self.seq_pool = embedding_dim, sequence_length, seq_pool
        assert sequence_length is not None or positional_embedding == "none"
        if not seq_pool:
            sequence_length += 1
            self.class_emb = nn.Parameter(torch.zeros(1, 1, embedding_dim))
        else:
            self.attention_pool = nn.Linear(embedding_dim, 1)
        if positional_embedding == "learnable":
            self.positional_emb = nn.Parameter(torch.zeros(1, sequence_length, embedding_dim))
            nn.init.trunc_normal_(self.positional_emb, std=0.2)
        elif positional_embedding == "sine":
            self.positional_emb = nn.Parameter(sinusoidal_embedding(sequence_length, embedding_dim),
                                               requires_grad=False)
        else:
            self.positional_emb = None
        self.dropout = nn.Dropout(dropout)
        rates = torch.linspace(0, stochastic_depth, num_layers).tolist()
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(embedding_dim, num_heads, int(embedding_dim * mlp_ratio), dropout,
                                    attention_dropout, rates[i]) for i in range(num_layers)])
        self.norm = nn.LayerNorm(embedding_dim)
        self.fc = nn.Linear(embedding_dim, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        if not self.seq_pool:
            x = torch.cat((self.class_emb.expand(x.shape[0], -1, -1), x), dim=1)
        if self.positional_emb is not None:
            x = x + self.positional_emb
        x = self.dropout(x)
        for blk in self.blocks:
            x = blk(x)
        x = self.norm(x)
        if self.seq_pool:
            x = seq_pool(x, self.attention_pool(x))
        else:
            x = x[:, 0]
        return self.fc(x)

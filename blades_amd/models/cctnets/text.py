"""Sequence (text) variants of the Compact-Transformer blocks: embedding front end, 1-D conv tokenizer and
padding-mask aware attention / encoder / classifier.

The reference vendors these under ``fllib/models/backbones/cctnets/utils/{embedder,tokenizer,transformers}.py``
(``Embedder``, ``TextTokenizer``, ``Masked*``); nothing in its catalog reaches them and two of them cannot run as
shipped (``mask._device``, an undefined ``self.n_channels``).  These are working equivalents with the same
constructor arguments and parameter names, so a text model written against the reference's classes loads here.
A mask is a bool tensor ``[B, N]``, True = real token.
"""
import torch
import torch.nn.functional as F
from torch import nn

from blades_amd.ops.activations import gelu, seq_pool

from .layers import DropPath, sinusoidal_embedding


class Embedder(nn.Module):
    def __init__(self, word_embedding_dim=300, vocab_size=100000, padding_idx=1, pretrained_weight=None,
                 embed_freeze=False, *args, **kwargs):
        super().__init__()
        if pretrained_weight is not None:
            self.embeddings = nn.Embedding.from_pretrained(pretrained_weight, freeze=embed_freeze)
        else:
            self.embeddings = nn.Embedding(vocab_size, word_embedding_dim, padding_idx=padding_idx)
            nn.init.normal_(self.embeddings.weight)
        self.embeddings.weight.requires_grad = not embed_freeze

    def forward(self, x, mask=None):
        e = self.embeddings(x)
        if mask is not None:
            e = e * (mask > 0).unsqueeze(-1).to(e.dtype)
        return e, mask


class TextTokenizer(nn.Module):
    """``[B, N, E]`` embeddings -> conv over (kernel_size tokens x full embedding) -> optional max-pool over
    tokens -> ``[B, N', C]``; the padding mask is carried through the same windows."""

    def __init__(self, kernel_size, stride, padding, pooling_kernel_size=3, pooling_stride=2, pooling_padding=1,
                 embedding_dim=300, n_output_channels=128, activation=None, max_pool=True, *args, **kwargs):
        super().__init__()
        self.max_pool = max_pool
        self.conv_layers = nn.Sequential(
            nn.Conv2d(1, n_output_channels, (kernel_size, embedding_dim), (stride, 1), (padding, 0), bias=False),
            nn.Identity() if activation is None else activation(),
            nn.MaxPool2d((pooling_kernel_size, 1), (pooling_stride, 1), (pooling_padding, 0)) if max_pool
            else nn.Identity())
        nn.init.kaiming_normal_(self.conv_layers[0].weight)

    def seq_len(self, seq_len=32, embed_dim=300) -> int:
        with torch.no_grad():
            dev = self.conv_layers[0].weight.device
            return self.forward(torch.zeros(1, seq_len, embed_dim, device=dev))[0].shape[1]

    def forward_mask(self, mask):
        conv, pool = self.conv_layers[0], self.conv_layers[2]
        m = mask.unsqueeze(1).float()
        m = F.conv1d(m, torch.ones(1, 1, conv.kernel_size[0], device=m.device), None, conv.stride[0], conv.padding[0])
        if self.max_pool:
            m = F.max_pool1d(m, pool.kernel_size[0], pool.stride[0], pool.padding[0])
        return m.squeeze(1) > 0

    def forward(self, x, mask=None):
        x = self.conv_layers(x.unsqueeze(1)).squeeze(-1).transpose(1, 2)  # [B, N', C]
        if mask is not None:
            mask = self.forward_mask(mask)
            x = x * mask.unsqueeze(-1).to(x.dtype)
        return x, mask


class MaskedAttention(nn.Module):
    def __init__(self, dim, num_heads=8, attention_dropout=0.1, projection_dropout=0.1):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.attn_drop = nn.Dropout(attention_dropout)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(projection_dropout)

    def forward(self, x, mask=None):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        attn = (q @ k.transpose(-2, -1)) * self.scale
        if mask is not None:
            if mask.shape[-1] != N:
                raise ValueError("mask has incorrect dimensions")
            pair = (mask[:, None, :] & mask[:, :, None]).unsqueeze(1)  # [B, 1, N, N]
            attn = attn.masked_fill(~pair, -torch.finfo(attn.dtype).max)
        attn = self.attn_drop(attn.softmax(dim=-1))
        return self.proj_drop(self.proj((attn @ v).transpose(1, 2).reshape(B, N, C)))


class MaskedTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, attention_dropout=0.1, drop_path_rate=0.1):
        super().__init__()
        self.pre_norm = nn.LayerNorm(d_model)
        self.self_attn = MaskedAttention(d_model, nhead, attention_dropout, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout2 = nn.Dropout(dropout)
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0 else nn.Identity()

    def forward(self, src, mask=None, *args, **kwargs):
        src = src + self.drop_path(self.self_attn(self.pre_norm(src), mask))
        src = self.norm1(src)
        src2 = self.linear2(self.dropout1(gelu(self.linear1(src))))
        return src + self.drop_path(self.dropout2(src2))


class MaskedTransformerClassifier(nn.Module):
    def __init__(self, seq_pool=True, embedding_dim=768, num_layers=12, num_heads=12, mlp_ratio=4.0,
                 num_classes=1000, dropout=0.1, attention_dropout=0.1, stochastic_depth=0.1,
                 positional_embedding="sine", seq_len=None, *args, **kwargs):
        super().__init__()
        if positional_embedding not in ("sine", "learnable", "none"):
            positional_embedding = "sine"
        if seq_len is None and positional_embedding != "none":
            raise ValueError(f"positional embedding '{positional_embedding}' needs seq_len")
        self.embedding_dim, self.seq_len, self.seq_pool, self.num_tokens = embedding_dim, seq_len, seq_pool, 0
        if not seq_pool:
            seq_len += 1
            self.class_emb = nn.Parameter(torch.zeros(1, 1, embedding_dim))
            self.num_tokens = 1
        else:
            self.attention_pool = nn.Linear(embedding_dim, 1)
        if positional_embedding == "learnable":
            self.positional_emb = nn.Parameter(torch.zeros(1, seq_len, embedding_dim))
            nn.init.trunc_normal_(self.positional_emb, std=0.2)
        elif positional_embedding == "sine":
            self.positional_emb = nn.Parameter(sinusoidal_embedding(seq_len, embedding_dim), requires_grad=False)
        else:
            self.positional_emb = None
        self._tokens = seq_len
        self.dropout = nn.Dropout(dropout)
        rates = torch.linspace(0, stochastic_depth, num_layers).tolist()
        self.blocks = nn.ModuleList([
            MaskedTransformerEncoderLayer(embedding_dim, num_heads, int(embedding_dim * mlp_ratio), dropout,
                                          attention_dropout, rates[i]) for i in range(num_layers)])
        self.norm = nn.LayerNorm(embedding_dim)
        self.fc = nn.Linear(embedding_dim, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x, mask=None):
        want = None if self._tokens is None else self._tokens - self.num_tokens
        if want is not None and x.size(1) < want:  # short sequences are right-padded to the embedding table
            pad = want - x.size(1)
            x = F.pad(x, (0, 0, 0, pad))
            if mask is not None:
                mask = F.pad(mask, (0, pad), value=False)
        if not self.seq_pool:
            x = torch.cat((self.class_emb.expand(x.shape[0], -1, -1), x), dim=1)
            if mask is not None:
                mask = torch.cat([torch.ones(mask.shape[0], 1, dtype=torch.bool, device=mask.device), mask > 0], dim=1)
        if self.positional_emb is not None:
            x = x + self.positional_emb
        x = self.dropout(x)
        for blk in self.blocks:
            x = blk(x, mask=mask)
        x = self.norm(x)
        if self.seq_pool:
            logit = self.attention_pool(x)
            if mask is not None:
                logit = logit.masked_fill(~mask.unsqueeze(-1), -torch.finfo(logit.dtype).max)
            x = seq_pool(x, logit)
        else:
            x = x[:, 0]
        return self.fc(x)


class TextCCT(nn.Module):
    """Embedder -> TextTokenizer -> MaskedTransformerClassifier (the text-classification assembly of the
    Compact-Transformers paper)."""

    def __init__(self, seq_len=64, word_embedding_dim=300, embedding_dim=128, kernel_size=2, stride=1, padding=1,
                 pooling_kernel_size=2, pooling_stride=2, pooling_padding=1, vocab_size=100000, padding_idx=1,
                 num_layers=2, num_heads=2, mlp_ratio=1.0, num_classes=2, positional_embedding="sine", **kw):
        super().__init__()
        self.embedder = Embedder(word_embedding_dim, vocab_size, padding_idx, **{k: kw[k] for k in
                                                                                ("pretrained_weight", "embed_freeze")
                                                                                if k in kw})
        self.tokenizer = TextTokenizer(kernel_size, stride, padding, pooling_kernel_size, pooling_stride,
                                       pooling_padding, word_embedding_dim, embedding_dim, nn.ReLU, True)
        self.classifier = MaskedTransformerClassifier(
            True, embedding_dim, num_layers, num_heads, mlp_ratio, num_classes,
            kw.get("dropout", 0.1), kw.get("attention_dropout", 0.1), kw.get("stochastic_depth", 0.1),
            positional_embedding, self.tokenizer.seq_len(seq_len, word_embedding_dim))

    def forward(self, x, mask=None):
        x, mask = self.embedder(x, mask)
        x, mask = self.tokenizer(x, mask)
        return self.classifier(x, mask)

"""State-dict adaptation helpers for Compact-Transformer checkpoints (reference:
``fllib/models/backbones/cctnets/utils/helpers.py`` — ``resize_pos_embed``, ``pe_check``, ``fc_check``).
Nothing is downloaded here; these serve checkpoints a user already has on disk."""
import logging
import math

import torch
import torch.nn.functional as F

_logger = logging.getLogger(__name__)


def resize_pos_embed(posemb: torch.Tensor, posemb_new: torch.Tensor, num_tokens: int = 1) -> torch.Tensor:
    """Bilinearly resample a ``[1, T + g*g, C]`` positional table to the grid size of ``posemb_new``;
    the first ``num_tokens`` (class token) rows are kept as they are."""
    n_new = posemb_new.shape[1] - num_tokens
    tok, grid = posemb[:, :num_tokens], posemb[0, num_tokens:]
    g_old, g_new = int(math.sqrt(grid.shape[0])), int(math.sqrt(n_new))
    if g_old * g_old != grid.shape[0] or g_new * g_new != n_new:
        raise ValueError("positional tables must hold square token grids")
    grid = grid.reshape(1, g_old, g_old, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(g_new, g_new), mode="bilinear", align_corners=False)
    grid = grid.permute(0, 2, 3, 1).reshape(1, g_new * g_new, -1)
    return torch.cat([tok, grid], dim=1)


def pe_check(model, state_dict, pe_key="classifier.positional_emb"):
    own = model.state_dict()
    if pe_key is not None and pe_key in state_dict and pe_key in own and own[pe_key].shape != state_dict[pe_key].shape:
        ntok = int(getattr(model.classifier, "num_tokens", 0 if getattr(model.classifier, "seq_pool", True) else 1))
        state_dict[pe_key] = resize_pos_embed(state_dict[pe_key], own[pe_key], num_tokens=ntok)
    return state_dict


def fc_check(model, state_dict, fc_key="classifier.fc"):
    own = model.state_dict()
    for key in (f"{fc_key}.weight", f"{fc_key}.bias"):
        if key in state_dict and key in own and own[key].shape != state_dict[key].shape:
            _logger.warning("Removing %s, number of classes has changed.", key)
            state_dict[key] = own[key]
    return state_dict

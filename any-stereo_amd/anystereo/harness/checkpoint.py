"""Checkpoint compatibility: the reference saves `model.state_dict()` of an
nn.DataParallel wrapper, so keys carry a `module.` prefix (train_continuous_IGEV.py:184,243-245);
restore is a strict load (evaluation.py:637-638)."""
from __future__ import annotations


def load_reference_state_dict(model, state_dict, strict: bool = True):
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
    return model.load_state_dict(sd, strict=strict)

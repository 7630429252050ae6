"""Attacks (same ten public names as ``blades/adversaries``)."""
from .adversary import Adversary, AdversaryConfig
from .simple import ALIEAdversary, IPMAdversary, NoiseAdversary, SignFlipAdversary, LabelFlipAdversary
from .optimized import (AdaptiveAdversary, MinMaxAdversary, SignGuardAdversary,
                        AttackclippedclusteringAdversary)

__all__ = [
    "Adversary",
    "AdversaryConfig",
    "SignFlipAdversary",
    "ALIEAdversary",
    "LabelFlipAdversary",
    "AdaptiveAdversary",
    "IPMAdversary",
    "NoiseAdversary",
    "SignGuardAdversary",
    "MinMaxAdversary",
    "AttackclippedclusteringAdversary",
]

"""Robust aggregators (same nine public names as ``fllib/aggregators``)."""
from .base import Aggregator
from .simple import Mean, Median, Trimmedmean, GeoMed, DnC
from .multikrum import Multikrum
from .centeredclipping import Centeredclipping
from .clippedclustering import Clippedclustering
from .signguard import Signguard

__all__ = [
    "Aggregator",
    "Mean",
    "Median",
    "Trimmedmean",
    "GeoMed",
    "DnC",
    "Clippedclustering",
    "Multikrum",
    "Centeredclipping",
    "Signguard",
]

"""Host-side mirror of the reference's ``nnet`` package surface (nnet/__init__.py:15-26)."""
from .config import parse_config
from .class_prior import get_class_prior

__all__ = ["parse_config", "get_class_prior"]

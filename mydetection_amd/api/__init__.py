"""Inference entry point of the package: `Detector` (mirror of the reference's api.detection.Detector)."""
from .detection import Detector

__all__ = ['Detector']

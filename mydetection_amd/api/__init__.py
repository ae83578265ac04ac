from .detection import Detector

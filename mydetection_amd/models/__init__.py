"""Model assembly through the plug-in seam: `name_to_model` and the four registry factories."""
from .general import name_to_model
from .registry import get_backbone, get_det_layer, get_fpn, get_rpn

__all__ = ['name_to_model', 'get_backbone', 'get_fpn', 'get_rpn', 'get_det_layer']

from .general import name_to_model
from .registry import get_backbone, get_fpn, get_rpn, get_det_layer

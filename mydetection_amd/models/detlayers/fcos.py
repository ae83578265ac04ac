"""FCOS decode layer with a separate centerness branch (reference: models/detlayers/fcos.py)."""
from .fcos2 import _FCOSInference


class FCOSLayer(_FCOSInference):
    '''
    'FCOS' (reference: models/detlayers/fcos.py:10-68): same decode as the FCOS2 layers, the centerness logit comes
    from the head's own 'center' branch (EfDetHead_wCenter).  Training (:70-190) is out of scope.
    '''
    conf_key = 'center'

    def __init__(self, level_i: int, cfg: dict):
        super().__init__()
        self.anch_min = cfg['model.fcos.anchors'][level_i]
        self.anch_max = cfg['model.fcos.anchors'][level_i + 1]
        self.stride = cfg['model.fpn.out_strides'][level_i]
        self.n_cls = cfg['general.num_class']
        self.center_region = 0.5
        self.ltrb_setting = 'exp_sl1'

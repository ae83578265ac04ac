"""Model assembly (reference: models/general.py)."""
from typing import List
import json

import torch

from .. import ops
from .registry import get_backbone, get_fpn, get_rpn, get_det_layer
from ..utils.structures import ImageObjects


def load_config(model_name):
    """The flat cfg dict of `model_name`: a maintainer's own `configs/<model_name>.json` under the project root when
    there is one (the reference's convention, models/general.py:12-16), otherwise the built-in table."""
    import os
    from .. import PROJECT_ROOT, configs
    path = f'{PROJECT_ROOT}/configs/{model_name}.json'
    if os.path.exists(path):
        return json.load(open(path, 'r'))
    return configs.get(model_name)


def name_to_model(model_name):
    '''(model, cfg) for configs/<model_name>.json  (reference: models/general.py:9-24)'''
    cfg = load_config(model_name)
    if cfg['base'] == 'OneStageBBox':
        model = OneStageBBox(cfg)
    else:
        raise Exception('Unknown model name')
    return model, cfg


def state_dict_template(model_name):
    """Keys/shapes/dtypes of the model's state_dict without allocating it (meta device)."""
    import contextlib
    import io
    with torch.device('meta'), contextlib.redirect_stdout(io.StringIO()):
        model = OneStageBBox(load_config(model_name))
    return model.state_dict()


class OneStageBBox(torch.nn.Module):
    '''
    backbone -> fpn -> head -> per-level decode -> List[ImageObjects]
    (reference: models/general.py:27-97).  Every stage is a HIP kernel chain; the per-level
    decoders write straight into the level-concatenated candidate arrays (levels in pyramid
    order along dim 1, models/general.py:74-76), which stay in HBM.
    '''
    def __init__(self, cfg: dict):
        super().__init__()
        self.backbone = get_backbone(cfg)
        self.fpn = get_fpn(cfg)
        self.rpn = get_rpn(cfg)

        det_layer = get_det_layer(cfg)
        self.det_layers = torch.nn.ModuleList()
        for level_i in range(len(cfg['model.fpn.out_channels'])):
            self.det_layers.append(det_layer(level_i=level_i, cfg=cfg))

        self.check_gt_assignment = cfg.get('train.check_gt_assignment', False)
        self.bb_format = cfg.get('general.pred_bbox_format', 'cxcywh')
        self.input_format = cfg['general.input_format']
        self.weights_epoch = 0
        # how many parts api.Detector evaluates an even batch in (graph.GraphedPath batch lanes): the EfficientNet-based
        # models are ~130 launches of 10-150 us per step and gain 5 % from two lanes, Darknet-53's launches fill the chip
        # alone and lose 5 % (profiles/r03_lanes.md)
        self.batch_lanes_hint = 2 if 'efficientnet' in str(cfg.get('model.backbone.name', '')).lower() else 1

    # captured hipGraphs (graph.GraphedPath) record the addresses of the kernel-ready parameter copies; the epoch tells
    # them that the parameters were replaced (in-place edits of single tensors are seen by the eager path through the
    # tensors' version counters, but not by a captured graph: call Detector.reset_graphs() after such an edit)
    def load_state_dict(self, *args, **kwargs):
        self.weights_epoch += 1
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.weights_epoch += 1
        return super()._apply(fn, *args, **kwargs)

    def forward_candidates(self, x):
        '''x [B,3,H,W] -> (bbox [B,N,4], class_idx [B,N] i64, score [B,N]) on the device.'''
        assert x.dim() == 4
        self.img_size = x.shape[2:4]
        features = self.backbone(x)
        features = self.fpn(features)
        if getattr(self.rpn, 'can_decode_retina', None) is not None and self.rpn.can_decode_retina(self.det_layers):
            # EfDetHead + RetinaLayer: decode in the epilogue of the towers' last layers (no class logits in memory)
            nB = x.shape[0]
            n_total = sum(d.num_anchors * f.shape[2] * f.shape[3] for d, f in zip(self.det_layers, features))
            bbs = torch.empty((nB, n_total, 4), dtype=torch.float32, device=x.device)
            cls_idx = torch.empty((nB, n_total), dtype=torch.int64, device=x.device)
            scores = torch.empty((nB, n_total), dtype=torch.float32, device=x.device)
            self.rpn.decode_retina(features, self.det_layers, self.img_size, bbs, cls_idx, scores)
            return bbs, cls_idx, scores
        all_branch_preds = self.rpn(features)

        counts = []
        for raw in all_branch_preds:
            shp = raw['bbox'].shape
            counts.append(int(torch.Size(shp[1:-1]).numel()))
        nB, n_total = x.shape[0], sum(counts)
        bbs = torch.empty((nB, n_total, 4), dtype=torch.float32, device=x.device)
        cls_idx = torch.empty((nB, n_total), dtype=torch.int64, device=x.device)
        scores = torch.empty((nB, n_total), dtype=torch.float32, device=x.device)
        descs = [getattr(layer, '_describe', lambda *_: None)(raw, self.img_size)
                 for layer, raw in zip(self.det_layers, all_branch_preds)]
        if all(d is not None for d in descs) and len({(d['mode'], d['layout'], d['A'], d['C']) for d in descs}) == 1:
            # every level in ONE launch, written at its offset of the level-concatenated arrays
            levels, n_off = [], 0
            for d, n in zip(descs, counts):
                levels.append(dict(d['level'], n_off=n_off))
                n_off += n
            d = descs[0]
            ops.decode_levels(d['mode'], levels, *d['layout'], d['A'], d['C'], nB, self.img_size, bbs, cls_idx, scores)
            return bbs, cls_idx, scores
        n_off = 0
        for i, raw_preds in enumerate(all_branch_preds):
            self.det_layers[i](raw_preds, self.img_size, None, _out=(bbs, cls_idx, scores, n_off))
            n_off += counts[i]
        return bbs, cls_idx, scores

    def forward(self, x, labels: List[ImageObjects] = None):
        '''
        x: a batch of images, e.g. shape(8,3,608,608)
        labels: must be None (training is outside the inference hot path)
        '''
        if labels is not None:
            raise NotImplementedError('training (labels != None) is outside the inference hot path')
        batch_bbs, batch_cls_idx, batch_scores = self.forward_candidates(x)
        batch_pred_objects = []
        for bbs, cls_idx, scores in zip(batch_bbs, batch_cls_idx, batch_scores):
            p_objs = ImageObjects(bboxes=bbs, cats=cls_idx, scores=scores,
                                  bb_format=self.bb_format, img_hw=self.img_size)
            batch_pred_objects.append(p_objs)
        return batch_pred_objects

"""API for the object detectors (reference: api/detection.py:19-205).

Same constructor and methods; the model forward and all post-processing run as HIP kernels
on the MI355X and only the surviving detections come back to the host.
"""
import os

import numpy as np
import PIL.Image
import torch

from .. import ops

from ..models.general import name_to_model
from ..utils import image_ops as imgUtils
from ..utils.structures import ImageObjects


class Detector():
    '''Wrapper for image object detectors

    Args:
        model_name: str, see mydetection_amd.configs.NAMES for available names
        model_and_cfg: (model, cfg) built elsewhere
        weights_path: checkpoint with a 'model' state_dict (reference key names)
        cpu: must be False -- this package has no CPU path
    '''
    def __init__(self, model_name: str = None, model_and_cfg: tuple = None,
                 weights_path: str = None, cpu=False):
        if cpu:
            raise RuntimeError('mydetection_amd.Detector(cpu=True): there is no CPU path; '
                               'use the reference for CPU inference')
        if model_and_cfg:
            self.model, cfg = model_and_cfg
        else:
            self.model, cfg = name_to_model(model_name)
            self.model.eval()
            self.model = self.model.cuda()

        self._init_preprocess(cfg)
        self._init_postprocess(cfg)

        n_params = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        print('Number of parameters:', n_params)
        if weights_path:
            self.model.load_state_dict(torch.load(weights_path)['model'])
        self.on_cpu = False
        # hipGraph replay of forward + post-process per (batch, H, W, conf, nms), captured the second time a shape is seen
        # (a one-off shape is not worth two warm-up passes); MYDET_GRAPH=0 keeps every call eager
        self.use_graph = os.environ.get('MYDET_GRAPH', '1') != '0'
        from ..graph import GraphCache
        self._graphs = GraphCache(int(os.environ.get('MYDET_MAX_GRAPHS', self._MAX_GRAPHS)))

    def _init_preprocess(self, cfg):
        self.divisibe = cfg['general.input_divisibility']
        self.input_size = cfg.get('test.default_input_size', None)
        self.preprocess = cfg['test.preprocessing']

    def _init_postprocess(self, cfg):
        self.conf_thres = cfg['test.default_conf_thres']
        self.nms_thres = cfg['test.nms_thres']

    def evaluation_predict(self, eval_info: dict, **kwargs):
        '''
        COCO-style detections of a whole evaluation set (reference: api/detection.py:58-76, a per-image loop).
        Images are decoded on the host and then handled `batch_size` (default 16) at a time: device resize + pad +
        normalise, ONE forward and ONE batched post-process per group of equal input size, boxes mapped back and
        converted to json rows on the device.  The list has the reference's order (image by image).
        '''
        img_dir = eval_info['image_dir']
        infos = list(eval_info['image_info']['images'])
        kwargs = dict(kwargs)
        batch_size = int(kwargs.pop('batch_size', 16))
        cat_map = kwargs.pop('catIdx2id', None)
        detection_json = []
        for i in range(0, len(infos), batch_size):
            chunk = infos[i:i + batch_size]
            imgs = [imgUtils.imread_pil(os.path.join(img_dir, info['file_name'])) for info in chunk]
            detection_json += self._json_batch(imgs, [info['id'] for info in chunk], eval_info['eval_type'], cat_map, **kwargs)
        return detection_json

    def predict_imgDir(self, img_dir, **kwargs):
        """reference: api/detection.py:93-110 (per-image loop); batched like evaluation_predict."""
        kwargs = dict(kwargs)
        batch_size = int(kwargs.pop('batch_size', 16))
        names = os.listdir(img_dir)
        detection_json = []
        for i in range(0, len(names), batch_size):
            chunk = names[i:i + batch_size]
            ids = []
            for imname in chunk:
                assert imname[-4] == '.'
                ids.append(int(imname[:-4]) if imname[:-4].isdigit() else imname[:-4])
            imgs = [imgUtils.imread_pil(os.path.join(img_dir, n)) for n in chunk]
            detection_json += self._json_batch(imgs, ids, 'x1y1wh', None, **kwargs)
        return detection_json

    def _json_batch(self, pil_imgs, img_ids, eval_type, cat_map, **kwargs):
        from ..utils.structures import batched_to_json
        out = [None] * len(pil_imgs)
        for idxs, rec in self._records_by_size(pil_imgs, **kwargs):
            rows = batched_to_json(rec, [img_ids[j] for j in idxs], eval_type, cat_map)
            counts = ops.check_counts(rec['count'].cpu().tolist())
            o = 0
            for j, k in zip(idxs, counts):
                out[j] = rows[o:o + k]
                o += k
        return [d for per_img in out for d in per_img]

    def detect_one(self, **kwargs):
        '''
        object detection in one single image: (img_path: str) or (pil_img: PIL.Image);
        see _predict_pil() for the optional arguments.  Drawing (return_img/show_img) is not
        part of the inference path and is not provided.
        '''
        assert 'pil_img' in kwargs or 'img_path' in kwargs
        img = kwargs.pop('pil_img', None) or imgUtils.imread_pil(kwargs.pop('img_path'))
        if kwargs.get('return_img', False) or kwargs.get('show_img', False):
            raise NotImplementedError('visualisation is outside the inference hot path')
        return self._predict_pil(img, **kwargs)

    def _predict_pil(self, pil_img, **kwargs):
        '''
        Args:
            pil_img, preprocessing (str), input_size (int), conf_thres (float), nms_thres (float)
        '''
        assert isinstance(pil_img, PIL.Image.Image), 'input must be a PIL.Image'
        return self.predict_batch([pil_img], **kwargs)[0]

    def _geometry(self, ori_h, ori_w, pre_proc_name, input_size=None):
        """What api/detection.py:177-205 does to an image of (ori_h, ori_w), as numbers: the resize target (h, w) or
        None, the (top, left) offset of the resized image inside the network input, the input size (H, W), and
        pad_info for bboxes_to_original_ (None when the boxes are already in image coordinates)."""
        assert isinstance(self.divisibe, int)
        div = self.divisibe

        def up(v):
            return int(np.ceil(v / div) * div)
        if pre_proc_name == 'pad_divisible':
            return None, (0, 0), (up(ori_h), up(ori_w)), None
        if pre_proc_name == 'resize_pad_divisible':
            assert input_size is not None
            factor = input_size / max(ori_h, ori_w)                  # utils/image_ops.py:30-33 (resize_pil, shorter=False)
            th, tw = round(ori_h * factor), round(ori_w * factor)
            return (th, tw), (0, 0), (up(th), up(tw)), (ori_w, ori_h, 0, 0, tw, th)
        if pre_proc_name == 'resize_pad_square':
            assert input_size is not None
            scale = input_size / max(ori_w, ori_h)                   # utils/image_ops.py:55-137 (rect_to_square, aug=False)
            rw, rh = int(ori_w * scale), int(ori_h * scale)
            left, top = (input_size - rw) // 2, (input_size - rh) // 2
            return (rh, rw), (top, left), (input_size, input_size), (ori_w, ori_h, left, top, rw, rh)
        raise Exception('Unknown preprocessing name')

    def preprocess_batch(self, pil_imgs, **kwargs):
        """Network inputs of a list of PIL images, grouped by input size: yields (indices, x [n,3,H,W] float32 on the
        device, pad_infos, image sizes).  Per image the host only decodes the file; resize (PIL-exact: the reference's
        tvf.resize of a PIL image), zero padding, /255 and normalisation (api/detection.py:158-163) are HIP kernels on
        the uint8 pixels."""
        pre_proc = kwargs.get('preprocessing', self.preprocess)
        input_size = kwargs.get('input_size', self.input_size)
        groups = {}
        for j, img in enumerate(pil_imgs):
            assert isinstance(img, PIL.Image.Image), 'input must be a PIL.Image'
            geo = self._geometry(img.height, img.width, pre_proc, input_size)
            groups.setdefault(geo[2], []).append((j, img, geo))
        dev = next(self.model.parameters()).device
        for (Hp, Wp), items in groups.items():
            buf = torch.zeros((len(items), Hp, Wp, 3), dtype=torch.uint8, device=dev)     # zero padding lives here
            for n, (j, img, (target, (top, left), _, _)) in enumerate(items):
                u8 = torch.from_numpy(np.array(img.convert('RGB'), dtype=np.uint8)).to(dev, non_blocking=True)
                ops.resize_bilinear_u8(u8, target or (img.height, img.width), buf[n], top, left)
            x = ops.preprocess_u8(buf, (Hp, Wp), self.model.input_format)
            yield ([j for j, _, _ in items], x, [g[3] for _, _, g in items],
                   [(img.height, img.width) if g[3] is not None else (Hp, Wp) for _, img, g in items])

    def _records_by_size(self, pil_imgs, **kwargs):
        """Detection records of a list of PIL images, grouped by network input size: yields (indices, records) with the
        boxes already in the coordinates of the original images."""
        conf_thres = kwargs.get('conf_thres', self.conf_thres)
        nms_thres = kwargs.get('nms_thres', self.nms_thres)
        for idxs, x, pads, hws in self.preprocess_batch(pil_imgs, **kwargs):
            rec = self._records(x, conf_thres, nms_thres)
            if any(p is not None for p in pads):
                ops.records_to_original_(rec, pads)
            rec['img_hw'] = hws
            yield idxs, rec

    _MAX_GRAPHS = 6

    def reset_graphs(self):
        """Forget every captured hipGraph (after editing parameters in place, or to release the graphs' activation pools)."""
        self._graphs.clear()

    def _records(self, x, conf_thres, nms_thres):
        """Detection records of one network input batch (boxes in network-input coordinates): a hipGraph replay when
        this (shape, thresholds) has been seen before, the eager launch sequence otherwise."""
        from ..utils.structures import batched_post_process
        key = (tuple(x.shape), float(conf_thres), float(nms_thres))
        if self.use_graph:
            cache = self._graphs
            g = cache.lookup(key)                                    # LRU; drops a graph captured before a weight change
            if g is None and cache.should_capture(key):
                from ..graph import GraphedPath
                # each graph owns its activations; the lane count is the detector's (below), not a timing decision
                g = cache.insert(key, GraphedPath(self.model, x, conf_thres, nms_thres, lanes=self.batch_lanes(x.shape[0])))
            if g is not None:
                return {k: v.clone() for k, v in g(x).items()}       # the graph's own record buffers are overwritten by the next replay
            cache.note_eager(key)
        with torch.no_grad():
            lanes = self.batch_lanes(x.shape[0])
            if lanes == 1:
                bb, ci, sc = self.model.forward_candidates(x)
                return batched_post_process(bb, ci, sc, conf_thres, nms_thres)
            # the eager form of a laned graph: the same parts of the batch, one after the other -- bit-identical to the replay
            records = torch.empty((x.shape[0], ops._lib.REC_WORDS), dtype=torch.int32, device=x.device)
            lo = 0
            for part in x.tensor_split(lanes):
                bb, ci, sc = self.model.forward_candidates(part)
                batched_post_process(bb, ci, sc, conf_thres, nms_thres, records=records[lo:lo + part.shape[0]])
                lo += part.shape[0]
            return ops.record_views(records)

    def batch_lanes(self, batch):
        """How many parts a batch of this size is evaluated in (graph.GraphedPath: parallel graph branches).  A fixed rule
        -- MYDET_LANES when it is a number, else the model's `batch_lanes_hint` (2 for the EfficientNet-based models,
        whose step is many short launches; 1 for Darknet-53) for even batches -- so that the eager calls that precede a
        capture and the replays that follow it give the same bits."""
        env = os.environ.get('MYDET_LANES', 'auto')
        want = int(env) if env.isdigit() else int(getattr(self.model, 'batch_lanes_hint', 1))
        if want <= 1 or batch < 2:
            return 1
        return min(want, batch) if env.isdigit() else (want if batch % want == 0 else 1)

    def predict_batch(self, pil_imgs, **kwargs):
        """Batched form of detect_one (the reference loops image by image, api/detection.py:67-74): images that share a
        network input size go through ONE forward + ONE batched post-process.  Returns a list of ImageObjects in the
        original image coordinates, in input order."""
        from ..parallel import records_to_objects
        out = [None] * len(pil_imgs)
        for idxs, rec in self._records_by_size(pil_imgs, **kwargs):
            objs = records_to_objects(rec, bb_format=self.model.bb_format)
            for j, o, hw in zip(idxs, objs, rec['img_hw']):
                o.img_hw = hw
                out[j] = o
        return out

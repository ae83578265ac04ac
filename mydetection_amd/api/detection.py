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

    def _init_preprocess(self, cfg):
        self.divisibe = cfg['general.input_divisibility']
        self.input_size = cfg.get('test.default_input_size', None)
        self.preprocess = cfg['test.preprocessing']

    def _init_postprocess(self, cfg):
        self.conf_thres = cfg['test.default_conf_thres']
        self.nms_thres = cfg['test.nms_thres']

    def evaluation_predict(self, eval_info: dict, **kwargs):
        img_dir = eval_info['image_dir']
        detection_json = []
        for imgInfo in eval_info['image_info']['images']:
            impath = os.path.join(img_dir, imgInfo['file_name'])
            detections = self.detect_one(img_path=impath, **kwargs)
            detection_json += detections.to_json(img_id=imgInfo['id'], eval_type=eval_info['eval_type'],
                                                 catIdx2id=kwargs.get('catIdx2id', None))
        return detection_json

    def predict_imgDir(self, img_dir, **kwargs):
        detection_json = []
        for imname in os.listdir(img_dir):
            detections = self.detect_one(img_path=os.path.join(img_dir, imname), **kwargs)
            assert imname[-4] == '.'
            img_id = int(imname[:-4]) if imname[:-4].isdigit() else imname[:-4]
            detection_json += detections.to_json(img_id=img_id, eval_type='x1y1wh')
        return detection_json

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
        pre_proc = kwargs.get('preprocessing', self.preprocess)
        input_size = kwargs.get('input_size', self.input_size)
        conf_thres = kwargs.get('conf_thres', self.conf_thres)
        nms_thres = kwargs.get('nms_thres', self.nms_thres)

        # host: PIL resize only; pad + to_tensor + normalise run in one HIP kernel on the uint8 image
        pil_img, pad_info, out_hw = self._preprocess_pil(pil_img, pre_proc, input_size, pad_on_device=True)
        u8 = torch.from_numpy(np.array(pil_img.convert('RGB'), dtype=np.uint8)).cuda()
        input_ = ops.preprocess_u8(u8, out_hw, self.model.input_format)
        assert input_.dim() == 4
        with torch.no_grad():
            dts = self.model(input_)
        assert isinstance(dts, list)
        dts: ImageObjects = dts[0]
        dts = dts.post_process(conf_thres, nms_thres)
        if pad_info is not None:
            dts.bboxes_to_original_(pad_info)
        return dts

    def _preprocess_pil(self, pil_img, pre_proc_name, input_size=None, pad_on_device=False):
        """reference: api/detection.py:177-205.  With pad_on_device the right/bottom zero padding of the
        '*_divisible' modes is left to the device kernel and the padded (H, W) is returned as third value."""
        assert isinstance(pil_img, PIL.Image.Image), 'input must be a PIL.Image'
        assert isinstance(self.divisibe, int)
        ori_h, ori_w = pil_img.height, pil_img.width
        div = self.divisibe

        def padded(img):
            return (int(np.ceil(img.height / div) * div), int(np.ceil(img.width / div) * div))
        if pre_proc_name == 'pad_divisible':
            out_hw = padded(pil_img)
            if not pad_on_device:
                pil_img = imgUtils.pad_to_divisible(pil_img, div)
            pad_info = None
        elif pre_proc_name == 'resize_pad_divisible':
            assert input_size is not None
            pil_img = imgUtils.resize_pil(pil_img, input_size, shorter=False)
            new_h, new_w = pil_img.height, pil_img.width
            out_hw = padded(pil_img)
            if not pad_on_device:
                pil_img = imgUtils.pad_to_divisible(pil_img, div)
            pad_info = (ori_w, ori_h, 0, 0, new_w, new_h)
        elif pre_proc_name == 'resize_pad_square':
            assert input_size is not None
            pil_img, _, pad_info = imgUtils.rect_to_square(pil_img, None, input_size, aug=False)
            out_hw = (pil_img.height, pil_img.width)
        else:
            raise Exception('Unknown preprocessing name')
        return (pil_img, pad_info, out_hw) if pad_on_device else (pil_img, pad_info)

    def predict_batch(self, pil_imgs, **kwargs):
        """Batched form of detect_one for images that preprocess to the same size (e.g. 'resize_pad_square'):
        one forward + one batched post-process for the whole list (the reference loops image by image,
        api/detection.py:67-74).  Returns a list of ImageObjects in the original image coordinates."""
        from ..parallel import records_to_objects
        from ..utils.structures import batched_post_process
        pre_proc = kwargs.get('preprocessing', self.preprocess)
        input_size = kwargs.get('input_size', self.input_size)
        conf_thres = kwargs.get('conf_thres', self.conf_thres)
        nms_thres = kwargs.get('nms_thres', self.nms_thres)
        u8s, pads, hw = [], [], None
        for img in pil_imgs:
            p_img, pad_info, out_hw = self._preprocess_pil(img, pre_proc, input_size, pad_on_device=True)
            arr = np.array(p_img.convert('RGB'), dtype=np.uint8)
            assert hw is None or (hw == out_hw and arr.shape == u8s[0].shape), 'images must preprocess to one size'
            hw = out_hw
            u8s.append(arr)
            pads.append(pad_info)
        u8 = torch.from_numpy(np.stack(u8s)).cuda()
        x = ops.preprocess_u8(u8, hw, self.model.input_format)
        with torch.no_grad():
            bb, ci, sc = self.model.forward_candidates(x)
            rec = batched_post_process(bb, ci, sc, conf_thres, nms_thres)
        objs = records_to_objects(rec, img_hw=tuple(hw), bb_format=self.model.bb_format)
        for o, pad_info in zip(objs, pads):
            if pad_info is not None:
                o.bboxes_to_original_(pad_info)
        return objs

"""Detection container + post-processing (reference: utils/structures.py:11-259).

Same fields and methods as the reference ``ImageObjects``; ``post_process`` / ``nms`` run the
batched HIP kernel (filter -> top-512 -> class-aware NMS) on the device that holds the
candidates instead of copying all N candidates to the host and looping over classes with
torchvision.ops.nms.  Results (order included) are those of the reference: class id
ascending, score descending inside a class.
"""
import torch

from .. import ops

TOPK = 512          # utils/structures.py:99-101


class ImageObjects():
    '''
    A group of image bounding boxes

    Args:
        bboxes: 2-d tensor, torch.float32
        cats: 1-d tensor, torch.int64, categories
        scores (optional): 1-d tensor, torch.float32, scores
        bb_format (optional): 'cxcywh' (the hot path); 'cxcywhd' is accepted by the container
                              but rotated boxes are outside the scope of the kernels
        img_hw: tuple-like, image (height, width)
    '''
    def __init__(self, bboxes, cats, masks=None, scores=None, bb_format='cxcywh', img_hw=None):
        self.bboxes: torch.FloatTensor = bboxes
        self.cats: torch.LongTensor = cats
        self.masks: torch.BoolTensor = masks
        self.scores: torch.FloatTensor = scores
        self._bb_format: str = bb_format
        self.img_hw: tuple = img_hw
        self.sanity_check()

    def __getitem__(self, idx):
        if isinstance(idx, int):
            idx = slice(idx, idx + 1)
        b = self.bboxes[idx, :]
        c = self.cats[idx]
        m = self.masks[idx, :, :] if self.masks is not None else None
        s = self.scores[idx] if self.scores is not None else None
        return ImageObjects(b, c, m, s, self._bb_format, self.img_hw)

    def __len__(self):
        return self.bboxes.shape[0]

    def cpu_(self):
        '''Move all attributes to CPU in-place'''
        self.bboxes = self.bboxes.cpu()
        self.cats = self.cats.cpu()
        self.scores = self.scores.cpu() if self.scores is not None else None
        self.masks = self.masks.cpu() if self.masks is not None else None

    def cuda_(self, device='cuda'):
        '''Move all attributes to the GPU in-place (the kernels need them there)'''
        self.bboxes = self.bboxes.to(device)
        self.cats = self.cats.to(device)
        self.scores = self.scores.to(device) if self.scores is not None else None
        self.masks = self.masks.to(device) if self.masks is not None else None

    def sort_by_score_(self, descending=True):
        '''Sort the bounding boxes by scores in-place'''
        assert self.scores is not None
        assert self.masks is None, 'sorting with masks is not currently supported'
        idxs = torch.argsort(self.scores, descending=descending)
        self.bboxes = self.bboxes[idxs, :]
        self.cats = self.cats[idxs]
        self.scores = self.scores[idxs]

    def category_filter_(self, categories) -> None:
        '''Keep the objects in the given category set and discard others in-place.'''
        assert self.masks is None, 'filtering with masks is not currently supported'
        keep_cats = torch.as_tensor(list(categories), dtype=torch.int64, device=self.cats.device)
        assert self.cats.dim() == 1 and keep_cats.dim() == 1
        keep_mask = (self.cats.unsqueeze(1) == keep_cats.unsqueeze(0)).any(dim=1)
        self.bboxes = self.bboxes[keep_mask]
        self.cats = self.cats[keep_mask]
        if self.scores is not None:
            self.scores = self.scores[keep_mask]

    # ------------------------------------------------------------------ post-processing
    def _device_fields(self):
        if not self.bboxes.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError('ImageObjects post-processing runs on MI355X only; no GPU is visible '
                                   '(mydetection_amd has no CPU path)')
            self.cuda_()
        return self.bboxes.contiguous(), self.cats.contiguous(), self.scores.contiguous()

    def _from_records(self, rec, b=0):
        k = ops.check_counts([int(rec['count'][b])])[0]          # the one host sync: how many survived
        return ImageObjects(rec['bbox'][b, :k], rec['class_idx'][b, :k], None, rec['score'][b, :k],
                            self._bb_format, img_hw=self.img_hw)

    def post_process(self, conf_thres, nms_thres):
        '''
        Confidence threshold + top-512 + class-aware NMS (reference: utils/structures.py:92-106),
        one HIP launch.  Returns a new ImageObjects whose tensors stay on the device.
        '''
        assert self.masks is None
        assert self.scores is not None
        if self._bb_format != 'cxcywh':
            raise NotImplementedError()
        bb, cats, sc = self._device_fields()
        rec = ops.postprocess(bb[None], cats[None], sc[None], conf_thres, nms_thres, TOPK)
        return self._from_records(rec)

    def nms(self, nms_thres=0.45):
        return ImageObjects.non_max_suppression(self, nms_thres)

    @staticmethod
    def non_max_suppression(dts, nms_thres: float):
        '''
        Class-aware NMS (reference: utils/structures.py:111-173).  At most 512 boxes, as in the
        reference call chain (post_process caps at 512 before calling nms).
        '''
        assert isinstance(dts, ImageObjects)
        assert dts.masks is None, 'nms with masks is not currently supported'
        assert dts.scores is not None
        if dts.bboxes.shape[0] == 0:
            return dts
        if dts._bb_format != 'cxcywh':
            raise NotImplementedError()
        if len(dts) > TOPK:
            raise NotImplementedError(f'non_max_suppression handles at most {TOPK} boxes per image')
        bb, cats, sc = dts._device_fields()
        rec = ops.postprocess(bb[None], cats[None], sc[None], float('-inf'), nms_thres, TOPK)
        return dts._from_records(rec)

    def bboxes_to_original_(self, pad_info):
        '''
        Recover the bbox from the padded image to the original image
        (reference: utils/structures.py:175-189).  pad_info: (ori w, ori h, tl x, tl y, imw, imh)
        '''
        assert self.masks is None, 'this func with masks is not currently supported'
        assert len(pad_info) == 6
        ori_w, ori_h = pad_info[0], pad_info[1]
        if len(self) > 0:
            if not self.bboxes.is_cuda:
                self._device_fields()
            self.bboxes = self.bboxes.contiguous()
            ops.bboxes_to_original_(self.bboxes, pad_info)
        self.img_hw = (ori_h, ori_w)

    def sanity_check(self):
        '''Integrity check (reference: utils/structures.py:191-213).'''
        assert self.bboxes.dtype == torch.float and self.bboxes.dim() == 2
        if self._bb_format == 'cxcywh':
            assert self.bboxes.shape[1] == 4
        elif self._bb_format == 'cxcywhd':
            assert self.bboxes.shape[1] == 5
        else:
            raise NotImplementedError()
        assert self.cats.dtype == torch.int64, 'Incorrect data type of categories'
        assert self.cats.dim() == 1 and self.cats.shape[0] == self.bboxes.shape[0]
        if self.masks is not None:
            assert self.masks.dtype == torch.bool and self.masks.dim() == 3
            assert self.masks.shape == (self.cats.shape[0],) + tuple(self.img_hw)
        if self.scores is not None:
            assert self.scores.shape[0] == self.bboxes.shape[0]
        assert self.img_hw is None or len(self.img_hw) == 2

    def to_json(self, img_id, eval_type='x1y1wh', catIdx2id=None) -> list:
        '''
        COCO-like json (reference: utils/structures.py:221-259).  The numbers (x1 = cx - w/2 ... in the reference's
        Python-float, i.e. double, arithmetic; category ids) come from one HIP launch and one device->host copy.
        '''
        assert self.bboxes.dim() == 2
        assert self.bboxes.shape[0] == self.cats.shape[0] == self.scores.shape[0]
        if eval_type != 'x1y1wh':
            raise NotImplementedError()
        assert self._bb_format == 'cxcywh'
        if len(self) == 0:
            return []
        bb, cats, sc = self._device_fields()
        table, host_map = _category_table(catIdx2id, bb.device)
        rows, cat = ops.detections_to_json(bb.contiguous()[None], sc.contiguous()[None], cats.contiguous()[None], None, table)
        return _json_rows(rows[0].cpu().tolist(), cat[0].cpu().tolist(), img_id, host_map)


def _category_table(catIdx2id, device):
    """(device int64 table or None, host mapping or None) for to_json's category lookup: the COCO ids by default
    (utils/constants.py:2, COCO_CATEGORY_LIST[c]['id']); an integer list / dict becomes a device table, anything else
    (e.g. string ids) is mapped on the host from the class indices."""
    from .constants import COCO_CATEGORY_IDS
    src = COCO_CATEGORY_IDS if catIdx2id is None else catIdx2id
    try:
        if isinstance(src, dict):
            n = max(src) + 1
            vals = [int(src.get(i, -1)) for i in range(n)]
            if any(not isinstance(v, int) for v in src.values()):
                raise TypeError
        else:
            vals = [v for v in src]
            if any(not isinstance(v, int) for v in vals):
                raise TypeError
        key = (tuple(vals), str(device))
        hit = _category_table.cache.get(key)
        if hit is None:
            hit = torch.tensor(vals, dtype=torch.int64, device=device)
            _category_table.cache[key] = hit
        return hit, None
    except (TypeError, ValueError):
        return None, src


_category_table.cache = {}


def _json_rows(rows, cats, img_id, host_map):
    out = []
    for r, c in zip(rows, cats):
        out.append({'image_id': img_id, 'category_id': host_map[int(c)] if host_map is not None else c,
                    'bbox': r[:4], 'score': r[4]})
    return out


def batched_to_json(rec, img_ids, eval_type='x1y1wh', catIdx2id=None) -> list:
    '''
    `to_json` of every image of a batch of detection records (the concatenation the reference builds image by image,
    api/detection.py:67-74): one launch, one device->host copy.  img_ids: one id per image.
    '''
    if eval_type != 'x1y1wh':
        raise NotImplementedError()
    table, host_map = _category_table(catIdx2id, rec['bbox'].device)
    rows, cat = ops.detections_to_json(rec['bbox'], rec['score'], rec['class_idx'], rec['count'], table)
    counts = ops.check_counts(rec['count'].cpu().tolist())
    rows, cat = rows.cpu(), cat.cpu()
    out = []
    for b, (k, img_id) in enumerate(zip(counts, img_ids)):
        out += _json_rows(rows[b, :k].tolist(), cat[b, :k].tolist(), img_id, host_map)
    return out


def batched_post_process(bboxes, cats, scores, conf_thres, nms_thres, records=None):
    '''
    The batched form of `for d in dts: d.post_process(...)` (examples/train.py:229-232):
    bboxes [B,N,4], cats [B,N], scores [B,N] on the device -> fixed-size records
    {count [B], bbox [B,512,4], class_idx [B,512], score [B,512], index [B,512]}, no host sync.
    records: optional int32 [B, REC_WORDS] buffer to write (rows of a larger batch's record buffer).
    '''
    return ops.postprocess(bboxes, cats, scores, conf_thres, nms_thres, TOPK, records=records)

"""Observed parity numbers (not a test: the tests assert the tolerances, this prints what was actually measured).

    python tests/parity_report.py > profiles/rNN_parity.json        (MI355X; imports oracle/, the CPU checker)

For every model configuration: the HIP path's candidates (scores, boxes, class ids) on two seeded 640x640 images against
the float32 CPU oracle (= the reference's arithmetic, tests/test_oracle_golden.py) -- the largest absolute error, the
largest error in units of the test tolerance (|err| / (1e-4 + 1e-4 |ref|): must stay below 1), class-id agreement --
and whether the post-processed detection records equal the oracle's post_process of the oracle's candidates."""
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mydetection_amd import synth                                          # noqa: E402
from mydetection_amd.models.general import name_to_model                   # noqa: E402
from mydetection_amd.utils.structures import batched_post_process          # noqa: E402
from oracle import efficientdet as oe, postprocess as opp, yolov3 as oy    # noqa: E402

out = {}
for name in ('yolov3_80', 'efficientdet-d1', 'd1_fcs2_atss', 'd1_fcs2', 'd1_fcs', 'd1_yv3'):
    with contextlib.redirect_stdout(sys.stderr):           # the factories print notices; stdout carries only the report
        m, cfg = name_to_model(name)
    m.load_state_dict(synth.make_state_dict(m.state_dict(), name), strict=True)
    m = m.eval().cuda()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    x = (synth.make_images if name == 'yolov3_80' else synth.make_normalized_images)(2, 640, seed=7)
    with torch.no_grad():
        if name == 'yolov3_80':
            ob, oc, os_ = oy.forward(x, sd)
        else:
            ob, oc, os_ = oe.forward(x, sd, name)
        bb, ci, sc = m.forward_candidates(x.cuda())
    bb, ci, sc = bb.cpu(), ci.cpu(), sc.cpu()
    es, eb = (sc - os_).abs(), (bb - ob).abs()
    entry = {'candidates_per_image': int(sc.shape[1]),
             'score_max_abs_err': float(es.max()), 'score_max_err_in_tolerances': float((es / (1e-4 + 1e-4 * os_.abs())).max()),
             'box_max_abs_err': float(eb.max()), 'box_max_err_in_tolerances': float((eb / (1e-4 + 1e-4 * ob.abs())).max()),
             'class_id_agreement': float((ci == oc).float().mean())}
    conf, nms = cfg['test.ap_conf_thres'], cfg['test.nms_thres']
    rec = batched_post_process(bb.cuda(), ci.cuda(), sc.cuda(), conf, nms)
    same = []
    for i in range(2):
        rb, rc, rs, src = opp.post_process(ob[i].numpy(), oc[i].numpy(), os_[i].numpy(), conf, nms)
        k = int(rec['count'][i])
        same.append(bool(k == len(src) and np.array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
                         and np.array_equal(rec['class_idx'][i, :k].cpu().numpy(), rc)))
        entry[f'detections_image{i}'] = [k, len(src)]
    entry['detection_sets_equal_oracle'] = same
    out[name] = entry
    del m
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))

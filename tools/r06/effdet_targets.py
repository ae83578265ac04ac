"""Round 6 (VERDICT r05 #2a): sweep final-layer targets of the EfficientDet family (mydetection_amd/synth.py: _EFDET_TARGETS) so that
the post-processing decisions differ by threshold.  The pyramid features of a few images are computed once with the oracle; every
setting regenerates only the final head layers and runs the oracle's head + decode.  CPU only (a tool, not product).
    python tools/r06/effdet_targets.py efficientdet-d1 640 0,1,2 "[{'object': (3.2, -4.3)}, ...]" """
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mydetection_amd import synth
from mydetection_amd.models.general import state_dict_template
from oracle import efficientdet as oe, decoders, postprocess as opp

torch.set_num_threads(8)
config = sys.argv[1] if len(sys.argv) > 1 else 'efficientdet-d1'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
seeds = [int(s) for s in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['0', '1'])]
grid = eval(sys.argv[4]) if len(sys.argv) > 4 else [{}]
tmpl = state_dict_template(config)
sd = synth.make_state_dict(tmpl, config)
calib = synth.load_calibration(config)
atss = config != 'efficientdet-d1'
FEATS = {}
for seed in seeds:
    x = synth.make_normalized_images(1, size, seed=seed)
    with torch.no_grad():
        FEATS[seed] = oe.bifpn(oe.backbone(x, sd, c6c7='conv' if atss else 'maxpool'), sd)
finals = [k for k in tmpl if synth._efdet_last_kind(k)]
base = dict(synth._EFDET_TARGETS)
for over in grid:
    synth._EFDET_TARGETS.clear(); synth._EFDET_TARGETS.update(base); synth._EFDET_TARGETS.update(over)
    for k in finals:
        sd[k] = synth.make_tensor(k, tmpl[k].shape, tmpl[k].dtype, calib)
    res = []
    for seed in seeds:
        with torch.no_grad():
            raws = oe.raw_dicts(oe.head(FEATS[seed], sd), 1 if atss else 9, 80, atss)
            outs = [decoders.fcos_decode(r, (size, size), oe.STRIDES[l]) if atss else
                    decoders.retina_decode(r, (size, size), oe.STRIDES[l], decoders.retina_anchors(oe.STRIDES[l])) for l, r in enumerate(raws)]
        bb, ci, sc = (torch.cat([o[j] for o in outs], 1)[0].numpy() for j in range(3))
        row = []
        for t in (0.005, 0.05, 0.5):
            b_, c_, s_, src = opp.post_process(bb, ci, sc, t, 0.5)
            row.append((int((sc >= t).sum()), len(src), len(np.unique(c_))))
        res.append(row)
    print(over, '| (pass, detections, classes) at 0.005 / 0.05 / 0.5 per image:', res, flush=True)

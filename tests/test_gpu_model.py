"""GPU parity of the assembled YOLOv3-80 path through the drop-in surface
(models.general.name_to_model / registry factories / ImageObjects.post_process) against the golden
vectors produced by the imported reference (batch 1, 512x512 = BASELINE configs[0]) and, at larger
sizes, against the CPU oracle and size-independent properties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-4        # north_star: boxes/scores within 1e-4 fp32 (allclose(1e-4, 1e-4))


@pytest.fixture(scope='module')
def model():
    assert torch.cuda.is_available()
    from mydetection_amd import synth
    from mydetection_amd.models.general import name_to_model
    m, cfg = name_to_model('yolov3_80')
    m.load_state_dict(synth.make_state_dict(m.state_dict()), strict=True)
    return m.eval().cuda(), cfg


def test_stage_parity_vs_reference_golden(model, golden):
    from mydetection_amd import synth
    m, cfg = model
    g = golden('yolov3_b1_512')
    x = synth.make_images(int(g['batch']), int(g['size']), seed=int(g['image_seed'])).cuda()
    with torch.no_grad():
        c = m.backbone(x)
        p = m.fpn(c)
        raws = m.rpn(p)
    for key, feats in (('backbone', c), ('fpn', p)):
        for lvl, f in enumerate(feats):
            assert tuple(g[f'{key}_{lvl}_shape']) == tuple(f.shape)
            f = f.contiguous().cpu().numpy()                      # logical NCHW order
            np.testing.assert_allclose(f.reshape(-1)[g[f'{key}_{lvl}_idx']], g[f'{key}_{lvl}_val'], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(np.sqrt((f.astype(np.float64) ** 2).sum()), g[f'{key}_{lvl}_l2'], rtol=1e-5)
    for lvl, raw in enumerate(raws):
        head = raw.packed['box'][0].contiguous().cpu().numpy()          # [B,255,H,W]
        assert tuple(g[f'head_{lvl}_shape']) == head.shape
        np.testing.assert_allclose(head.reshape(-1)[g[f'head_{lvl}_idx']], g[f'head_{lvl}_val'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(raws[2].packed['box'][0].contiguous().cpu().numpy(), g['head_2_full'], rtol=RTOL, atol=ATOL)
    # reference-shaped raw views
    assert raws[0]['bbox'].shape == (1, 3, 64, 64, 4) and raws[0]['class'].shape == (1, 3, 64, 64, 80)


def _class_ids_exact_where_safe(got, ref, margin, what):
    """Class ids are exact wherever the reference's two largest class probabilities differ by more than float32
    round-off (the fixture stores that gap per candidate); such candidates must be all but a handful."""
    safe = margin > 2e-5
    assert safe.mean() > 0.995, f'{what}: fixture has too many tied classes'
    np.testing.assert_array_equal(got[safe], ref[safe], err_msg=what)


def _check_yolo_golden(m, g, what, strict=False):
    """Candidates and detections of a YOLO-head model against a fixture made by the imported reference: every candidate
    within north_star's 1e-4, class ids exact where the fixture says they are well defined, detections at the three
    settings exact (count, class ids, order; scores / boxes 1e-4) whenever the fixture's decision margin exceeds twice
    the score error observed here -- thousands of long-tailed scores pass 0.005, so gaps at the top-512 cut are ~1e-6
    (oracle/gen_golden.py:gen_yolov3) -- and in every case equal to the oracle's post_process of THESE candidates.
    strict (the yolov3_80 fixtures, whose calibrated head gives non-trivial, different detection sets at all three
    settings with margins of 1e-5 and more): every setting must keep >= 50 detections in >= 10 classes, the fixture's
    margin must exceed twice the score error observed here, and the detections must equal the reference's at all three.
    strict='margin-permitting' (the same fixtures evaluated on the F(4x4) kernels, whose round-off is ~3x larger): the
    non-vacuity checks, and equality wherever the margin still exceeds twice the observed error (at least two settings)."""
    from mydetection_amd import synth
    from oracle import postprocess as opp
    x = synth.make_images(int(g['batch']), int(g['size']), seed=int(g['image_seed'])).cuda()
    with torch.no_grad():
        d = m(x)[0]
    n = g['bboxes_0'].shape[0]
    assert d.bboxes.shape == (n, 4) and d.cats.dtype == torch.int64
    cats, scores, boxes = d.cats.cpu().numpy(), d.scores.cpu().numpy(), d.bboxes.cpu().numpy()
    np.testing.assert_allclose(scores, g['scores_0'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(boxes, g['bboxes_0'], rtol=RTOL, atol=ATOL)
    _class_ids_exact_where_safe(cats, g['cats_0'], g['cls_margin_0'], what)
    err = float(np.abs(scores - g['scores_0']).max())
    exact_tags = 0
    # the post-process kernel on the REFERENCE's own candidates reproduces the reference's detections exactly at all three
    # settings (count, classes, order, scores and boxes bit for bit): with the 1e-4 gate on every candidate above, this pins
    # the detections to the reference without leaning on score gaps that are narrower than float32 round-off
    from mydetection_amd.utils.structures import ImageObjects
    ref_d = ImageObjects(torch.from_numpy(g['bboxes_0']).cuda(), torch.from_numpy(g['cats_0']).cuda(), None,
                         torch.from_numpy(g['scores_0']).cuda(), d._bb_format, d.img_hw)
    for tag in ('ap', 'mid', 'demo'):
        if float(g[f'pp_{tag}_margin']) > 0.0:            # (exact score ties are broken by torch.topk's unspecified order)
            rr = ref_d.post_process(float(g[f'pp_{tag}_conf']), float(g[f'pp_{tag}_nms']))
            np.testing.assert_array_equal(rr.cats.cpu().numpy(), g[f'pp_{tag}_cats_0'], err_msg=f'{what} {tag}')
            np.testing.assert_array_equal(rr.scores.cpu().numpy(), g[f'pp_{tag}_scores_0'], err_msg=f'{what} {tag}')
            np.testing.assert_array_equal(rr.bboxes.cpu().numpy(), g[f'pp_{tag}_bboxes_0'], err_msg=f'{what} {tag}')
    for tag in ('ap', 'mid', 'demo'):
        conf, nms = float(g[f'pp_{tag}_conf']), float(g[f'pp_{tag}_nms'])
        r = d.post_process(conf, nms)
        ob, oc, os_, _ = opp.post_process(boxes, cats, scores, conf, nms)
        assert len(r) == len(oc)
        np.testing.assert_array_equal(r.cats.cpu().numpy(), oc)
        np.testing.assert_array_equal(r.scores.cpu().numpy(), os_)
        np.testing.assert_array_equal(r.bboxes.cpu().numpy(), ob)
        ref_c, ref_s, ref_b = g[f'pp_{tag}_cats_0'], g[f'pp_{tag}_scores_0'], g[f'pp_{tag}_bboxes_0']
        same = len(r) == len(ref_c) and np.array_equal(r.cats.cpu().numpy(), ref_c)
        if strict:
            assert len(ref_c) >= 50 and len(np.unique(ref_c)) >= 10, f'{what} {tag}: vacuous fixture ({len(ref_c)} detections)'
        if strict is True:
            assert float(g[f'pp_{tag}_margin']) > 2 * err, f'{what} {tag}: score error {err:.1e} is not inside the fixture margin'
        if float(g[f'pp_{tag}_margin']) > 2 * err:
            assert same, f'{what} {tag}: decisions differ from the reference although its margin {float(g[f"pp_{tag}_margin"]):.1e} > 2 x {err:.1e}'
        if same:
            exact_tags += 1
            np.testing.assert_allclose(r.scores.cpu().numpy(), ref_s, rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(r.bboxes.cpu().numpy(), ref_b, rtol=RTOL, atol=ATOL)
        else:       # a decision inside the round-off band flipped: the detection sets may differ by the candidates involved
            assert strict is not True and abs(len(r) - len(ref_c)) <= 4, f'{what} {tag}: {len(r)} vs {len(ref_c)} detections'
    assert exact_tags >= (3 if strict is True else 2), f'{what}: detections equal the reference\'s at only {exact_tags} of 3 settings'
    if strict:      # the three settings decide different things: the top-512 cut applies at the first, not at the others
        n_ap, n_mid, n_demo = (len(g[f'pp_{t}_cats_0']) for t in ('ap', 'mid', 'demo'))
        assert n_ap > n_mid > n_demo and int((g['scores_0'] >= float(g['pp_ap_conf'])).sum()) > 512 > int((g['scores_0'] >= float(g['pp_mid_conf'])).sum())
    return err


def test_end_to_end_vs_reference_golden(model, golden):
    """BASELINE configs[0] shape (batch 1, 512 x 512) against the imported reference."""
    m, cfg = model
    _check_yolo_golden(m, golden('yolov3_b1_512'), 'yolov3 512', strict=True)


def test_end_to_end_vs_reference_golden_640(model, golden):
    """The benchmark resolution (640 x 640, batch 1) pinned by the imported reference itself: stage samples, head logits,
    all 25 200 candidates, detections at three settings."""
    from mydetection_amd import synth
    m, cfg = model
    g = golden('yolov3_b1_640')
    x = synth.make_images(1, 640, seed=int(g['image_seed'])).cuda()
    with torch.no_grad():
        c = m.backbone(x)
        p = m.fpn(c)
        raws = m.rpn(p)
    for key, feats in (('backbone', c), ('fpn', p)):
        for lvl, f in enumerate(feats):
            assert tuple(g[f'{key}_{lvl}_shape']) == tuple(f.shape)
            f = f.contiguous().cpu().numpy()
            np.testing.assert_allclose(f.reshape(-1)[g[f'{key}_{lvl}_idx']], g[f'{key}_{lvl}_val'], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(np.sqrt((f.astype(np.float64) ** 2).sum()), g[f'{key}_{lvl}_l2'], rtol=1e-5)
    for lvl, raw in enumerate(raws):
        head = raw.packed['box'][0].contiguous().cpu().numpy()
        np.testing.assert_allclose(head.reshape(-1)[g[f'head_{lvl}_idx']], g[f'head_{lvl}_val'], rtol=RTOL, atol=ATOL)
    assert g['bboxes_0'].shape == (25200, 4)
    _check_yolo_golden(m, g, 'yolov3 640', strict=True)


def test_ultralytics_plugins_vs_reference_golden(golden):
    """Registry plug-ins 'ultralytics' backbone + FPN (Focus, Conv, Bottleneck, BottleneckCSP, SPP) under the YOLO head,
    configs/u5m_yv3.json, against the imported reference (batch 1, 256 x 256): stage samples, head logits, every
    candidate within 1e-4, class ids exact where defined, detections at three settings; and 'u5m_fcs2' (the same trunk
    under the anchor-free FCOS2 decode) end to end against its CPU oracle composition."""
    from mydetection_amd import synth
    from mydetection_amd.models.general import name_to_model
    g = golden('u5m_yv3_b1_256')
    m, cfg = name_to_model('u5m_yv3')
    m.load_state_dict(synth.make_state_dict(m.state_dict(), 'u5m_yv3'), strict=True)
    m = m.eval().cuda()
    x = synth.make_images(1, 256, seed=int(g['image_seed'])).cuda()
    with torch.no_grad():
        c = m.backbone(x)
        p = m.fpn(c)
        raws = m.rpn(p)
    for key, feats in (('backbone', c), ('fpn', p)):
        for lvl, f in enumerate(feats):
            assert tuple(g[f'{key}_{lvl}_shape']) == tuple(f.shape)
            f = f.contiguous().cpu().numpy()
            np.testing.assert_allclose(f.reshape(-1)[g[f'{key}_{lvl}_idx']], g[f'{key}_{lvl}_val'], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(np.sqrt((f.astype(np.float64) ** 2).sum()), g[f'{key}_{lvl}_l2'], rtol=1e-5)
    for lvl, raw in enumerate(raws):
        head = raw.packed['box'][0].contiguous().cpu().numpy()
        assert tuple(g[f'head_{lvl}_shape']) == head.shape
        np.testing.assert_allclose(head.reshape(-1)[g[f'head_{lvl}_idx']], g[f'head_{lvl}_val'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(raws[2].packed['box'][0].contiguous().cpu().numpy(), g['head_2_full'], rtol=RTOL, atol=ATOL)
    _check_yolo_golden(m, g, 'u5m_yv3 256')
    # a batch of 2 at 320 x 256 against the oracle (non-square, every level odd-sized somewhere)
    from oracle import ultralytics as ou
    x2 = synth.make_images(2, (320, 256), seed=11)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ob, oc, os_ = ou.forward(x2, sd)
        bb, ci, sc = m.forward_candidates(x2.cuda())
    np.testing.assert_allclose(sc.cpu().numpy(), os_.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(bb.cpu().numpy(), ob.numpy(), rtol=RTOL, atol=ATOL)
    # u5m_fcs2: same trunk, one anchor, FCOS2 decode (configs/u5m_fcs2.json)
    from oracle import decoders
    m2, cfg2 = name_to_model('u5m_fcs2')
    sd2 = synth.make_state_dict(m2.state_dict(), 'u5m_fcs2')
    m2.load_state_dict(sd2, strict=True)
    m2 = m2.eval().cuda()
    with torch.no_grad():
        bb, ci, sc = m2.forward_candidates(x2.cuda())
        outs, gaps = [], []
        for lvl, r in enumerate(oy_head(ou.forward_features(x2, sd2), sd2)):
            t = r.permute(0, 2, 3, 1)
            outs.append(decoders.fcos_decode({'bbox': t[..., 0:4], 'conf': t[..., 4:5], 'class': t[..., 5:]}, (320, 256), (8, 16, 32)[lvl]))
            top2 = torch.sigmoid(t[..., 5:]).reshape(t.shape[0], -1, t.shape[-1] - 5).topk(2, dim=-1).values
            gaps.append(top2[..., 0] - top2[..., 1])
    ob, oc, os_ = (torch.cat([o[j] for o in outs], dim=1) for j in range(3))
    assert bb.shape == ob.shape
    np.testing.assert_allclose(sc.cpu().numpy(), os_.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(bb.cpu().numpy(), ob.numpy(), rtol=RTOL, atol=ATOL)
    # class ids: exact wherever the oracle's own two largest class probabilities are further apart than round-off
    _class_ids_exact_where_safe(ci.cpu().numpy(), oc.numpy(), torch.cat(gaps, dim=1).numpy(), 'u5m_fcs2')


def oy_head(feats, sd):
    from oracle import yolov3 as oy
    return oy.yolo_head(feats, sd)


def test_batch_consistency_and_post_process_vs_oracle(model):
    """Batch of 3 at 256x320: every image equals its single-image run (no cross-image coupling) -- to 1e-5, not bit
    for bit: small grids are cut along K to fill the chip, and where a tile's K range is cut depends on the grid --
    and post-processing of the GPU candidates equals the oracle's on the same candidates."""
    from mydetection_amd import synth
    from mydetection_amd.utils.structures import batched_post_process
    from oracle import postprocess as pp
    m, cfg = model
    x = synth.make_images(3, (256, 320), seed=3).cuda()
    with torch.no_grad():
        bb, ci, sc = m.forward_candidates(x)
        for i in range(3):
            b1, c1, s1 = m.forward_candidates(x[i:i + 1])
            np.testing.assert_allclose(s1[0].cpu().numpy(), sc[i].cpu().numpy(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(b1[0].cpu().numpy(), bb[i].cpu().numpy(), rtol=1e-5, atol=1e-5)
            assert (c1[0] != ci[i]).float().mean().item() < 1e-3
            b2, c2, s2 = m.forward_candidates(x[i:i + 1])            # ... and a run is reproducible bit for bit
            assert torch.equal(b1, b2) and torch.equal(c1, c2) and torch.equal(s1, s2)
    rec = batched_post_process(bb, ci, sc, 0.005, 0.45)
    for i in range(3):
        ob, oc, os_, src = pp.post_process(bb[i].cpu().numpy(), ci[i].cpu().numpy(), sc[i].cpu().numpy(), 0.005, 0.45)
        k = int(rec['count'][i])
        assert k == len(src)
        np.testing.assert_array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
        np.testing.assert_array_equal(rec['class_idx'][i, :k].cpu().numpy(), oc)
        np.testing.assert_array_equal(rec['bbox'][i, :k].cpu().numpy(), ob)


def test_forward_vs_oracle_640_batch2(model):
    """640x640 (the benchmark resolution), batch 2, against the CPU oracle forward."""
    from mydetection_amd import synth
    from oracle import yolov3 as oy
    m, cfg = model
    x = synth.make_images(2, 640, seed=5)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    torch.set_num_threads(max(1, torch.get_num_threads()))
    with torch.no_grad():
        ob, oc, os_, raws = oy.forward(x, sd, return_raw=True)
        bb, ci, sc = m.forward_candidates(x.cuda())
    assert bb.shape == (2, 25200, 4)
    np.testing.assert_allclose(sc.cpu().numpy(), os_.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(bb.cpu().numpy(), ob.numpy(), rtol=RTOL, atol=ATOL)
    # class ids exact wherever the oracle's own two largest class probabilities are further apart than round-off
    margin = []
    for r in raws:
        t = torch.sigmoid(r.view(2, 3, 85, *r.shape[2:])[:, :, 5:].permute(0, 1, 3, 4, 2)).reshape(2, -1, 80).topk(2, dim=-1).values
        margin.append(t[..., 0] - t[..., 1])
    for i in range(2):
        _class_ids_exact_where_safe(ci[i].cpu().numpy(), oc[i].numpy(), torch.cat(margin, dim=1)[i].numpy(), f'yolov3 640 image {i}')


def test_detector_api(model, tmp_path):
    """api.detection.Detector on a PIL image: plumbing (preprocess, forward, post_process, box rescale, json)."""
    import PIL.Image
    from mydetection_amd import synth
    from mydetection_amd.api import Detector
    m, cfg = model
    det = Detector(model_and_cfg=(m, cfg))
    img = (synth.make_images(1, (300, 400), seed=9)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    path = tmp_path / '17.png'
    PIL.Image.fromarray(img).save(path)
    dts = det.detect_one(img_path=str(path), input_size=320, conf_thres=0.005)
    assert dts.img_hw == (300, 400)
    js = dts.to_json(img_id=17)
    assert len(js) == len(dts) > 0
    assert set(js[0]) == {'image_id', 'category_id', 'bbox', 'score'}
    with pytest.raises(Exception):
        det.detect_one(pil_img=PIL.Image.fromarray(img), preprocessing='bogus')


# ------------------- EfficientDet-D1 / D1-FCOS2-ATSS, and the registry plug-ins 'FCOS2' (d1_fcs2), 'effrpn_ct' + 'FCOS' (d1_fcs), EfDetHead + 'YOLO' (d1_yv3)
@pytest.fixture(scope='module', params=['efficientdet-d1', 'd1_fcs2_atss', 'd1_fcs2', 'd1_fcs', 'd1_yv3', 'd1_fcs2_p3'])
def effdet(request):
    assert torch.cuda.is_available()
    from mydetection_amd import synth
    from mydetection_amd.models.general import name_to_model
    name = request.param
    m, cfg = name_to_model(name)
    m.load_state_dict(synth.make_state_dict(m.state_dict(), name), strict=True)
    return name, m.eval().cuda(), cfg


def test_effdet_family_vs_reference_golden(effdet, golden):
    """Full path through the registry seam (EfficientNet-B1 -> 4x BiFPN -> EfDetHead -> Retina/FCOS/YOLO decode ->
    post_process) against the imported reference (batch 1, 256x256): north_star's gate -- boxes/scores within 1e-4
    (rtol and atol), class ids and detection counts exact -- at every stage, on every candidate and at all three
    post-process settings (each keeps hundreds of detections)."""
    name, m, cfg = effdet
    _check_effdet_golden(name, m, golden(name.replace('-', '_') + '_b1_256'))


def test_effdet_vs_reference_golden_640(effdet, golden):
    """BASELINE configs[2] / [3] at the benchmark resolution (640 x 640, batch 1), pinned by the imported reference itself:
    stage samples of trunk / first BiFPN layer / pyramid, head logits, all 76 725 / 8 525 candidates within 1e-4, class ids
    exact where defined, detections (count, classes, order; scores / boxes 1e-4) at the three settings.  The fixture's
    image seed is the first whose post-processing decisions are all 2e-5 clear of a boundary (oracle/gen_golden.py)."""
    name, m, cfg = effdet
    if name not in ('efficientdet-d1', 'd1_fcs2_atss'):
        pytest.skip('640 x 640 reference fixtures exist for the two benchmark configurations')
    g = golden(name.replace('-', '_') + '_b1_640')
    assert g['bboxes_0'].shape[0] == (76725 if name == 'efficientdet-d1' else 8525)
    _check_effdet_golden(name, m, g)


def _check_effdet_golden(name, m, g):
    from mydetection_amd import synth
    x = synth.make_normalized_images(int(g['batch']), int(g['size']), seed=int(g['image_seed'])).cuda()
    with torch.no_grad():
        c = m.backbone(x)
        p0 = m.fpn[0](c)
        p = m.fpn(c)
        raws = m.rpn(p)
        dts = m(x)
    for key, feats in (('backbone', c), ('bifpn0', p0), ('fpn', p)):
        for lvl, f in enumerate(feats):
            assert tuple(g[f'{key}_{lvl}_shape']) == tuple(f.shape)
            f = f.contiguous().cpu().numpy()
            np.testing.assert_allclose(f.reshape(-1)[g[f'{key}_{lvl}_idx']], g[f'{key}_{lvl}_val'], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(np.sqrt((f.astype(np.float64) ** 2).sum()), g[f'{key}_{lvl}_l2'], rtol=1e-5)
    for lvl, raw in enumerate(raws):                       # head logits through the reference-shaped views
        for k in raw:
            v = raw[k].contiguous().cpu().numpy().reshape(-1)
            np.testing.assert_allclose(v[g[f'head_{lvl}_{k}_idx']], g[f'head_{lvl}_{k}_val'], rtol=RTOL, atol=ATOL)
    d = dts[0]
    n = g['bboxes_0'].shape[0]
    assert d.bboxes.shape == (n, 4)
    np.testing.assert_allclose(d.scores.cpu().numpy(), g['scores_0'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(d.bboxes.cpu().numpy(), g['bboxes_0'], rtol=RTOL, atol=ATOL)
    _class_ids_exact_where_safe(d.cats.cpu().numpy(), g['cats_0'], g['cls_margin_0'], name)
    # Detections at the three settings.  Round 6 (VERDICT r05 #2a): the head recipe (synth._EFDET_TARGETS) gives an objectness-like
    # score distribution, so the settings decide different things -- at 640^2 more than 512 candidates pass 0.005 (the top-512 cut
    # applies), fewer than 512 pass 0.05, 50-150 pass the demo threshold, >= 30 classes among the detections.  Each fixture
    # records how far its decisions are from a boundary (pp_<tag>_margin): where that exceeds twice the score error observed
    # here the detections must equal the reference's exactly; the 640^2 fixtures (margins >= 5e-5) must be exact at all three.
    from oracle import postprocess as opp
    cats, scores, boxes = d.cats.cpu().numpy(), d.scores.cpu().numpy(), d.bboxes.cpu().numpy()
    err = float(np.abs(scores - g['scores_0']).max())
    big = int(g['size']) >= 640
    n_ap, n_mid, n_demo = (len(g[f'pp_{t}_cats_0']) for t in ('ap', 'mid', 'demo'))
    p_ap, p_mid, p_demo = (int((g['scores_0'] >= float(g[f'pp_{t}_conf'])).sum()) for t in ('ap', 'mid', 'demo'))
    assert n_ap > n_mid > n_demo >= 10, f'{name}: the three settings keep {n_ap} / {n_mid} / {n_demo} detections (vacuous fixture)'
    if big:
        assert p_ap > 512 > p_mid > p_demo >= 50 and n_demo >= 50, (p_ap, p_mid, p_demo, n_demo)
        assert len(np.unique(g['pp_ap_cats_0'])) >= 30 and len(np.unique(g['pp_demo_cats_0'])) >= 10
    exact_tags = 0
    for tag in ('ap', 'mid', 'demo'):
        conf, nms = float(g[f'pp_{tag}_conf']), float(g[f'pp_{tag}_nms'])
        margin = float(g[f'pp_{tag}_margin'])
        r = d.post_process(conf, nms)
        ob, oc, os_, _ = opp.post_process(boxes, cats, scores, conf, nms)      # always: == the oracle's post-process of THESE candidates
        assert len(r) == len(oc)
        np.testing.assert_array_equal(r.cats.cpu().numpy(), oc)
        np.testing.assert_array_equal(r.scores.cpu().numpy(), os_)
        np.testing.assert_array_equal(r.bboxes.cpu().numpy(), ob)
        ref_c, ref_s, ref_b = g[f'pp_{tag}_cats_0'], g[f'pp_{tag}_scores_0'], g[f'pp_{tag}_bboxes_0']
        same = len(r) == len(ref_c) and np.array_equal(r.cats.cpu().numpy(), ref_c)
        if big:
            assert margin > 2 * err, f'{name} {tag}: score error {err:.1e} is not inside the fixture margin {margin:.1e}'
        if margin > 2 * err:
            assert same, f'{name} {tag}: {len(r)} vs {len(ref_c)} detections although the margin {margin:.1e} > 2 x {err:.1e}'
        if same:
            exact_tags += 1
            np.testing.assert_allclose(r.scores.cpu().numpy(), ref_s, rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(r.bboxes.cpu().numpy(), ref_b, rtol=RTOL, atol=ATOL)
        else:       # a decision inside the round-off band flipped: the sets may differ by the candidates involved
            assert abs(len(r) - len(ref_c)) <= 4, f'{name} {tag}: {len(r)} vs {len(ref_c)} detections'
    assert exact_tags >= (3 if big else 2), f'{name}: detections equal the reference\'s at only {exact_tags} of 3 settings (score error {err:.1e})'


def _wino4_launches(fn):
    """Run fn() with the per-launch timer on; returns (result, number of F(4x4) conv launches it issued)."""
    from mydetection_amd import ops
    ops.TIMER = ops.KernelTimer()
    try:
        out = fn()
    finally:
        timer, ops.TIMER = ops.TIMER, None
    torch.cuda.synchronize()
    return out, len(timer.spans.get('conv_wino4', []))


def test_f4x4_kernels_vs_reference_goldens(model, golden, monkeypatch):
    """The F(4x4,3x3) pair (wino4_input_kernel + conv_wino4_kernel: half of the headline step) against the imported
    reference end to end.  ops.conv2d hands a layer to F(4x4) only from 512 workgroups up, so at the fixtures' batch 1
    every 3x3 layer runs F(2x2); with WINO4_MIN_ITEMS = 1 every 3x3 stride-1 layer with Cin >= 64 takes the F(4x4) pair
    instead (31 of YOLOv3's 32; the C6 / C7 convs and the dense class layer of D1-FCOS-ATSS) and the same reference
    gates apply: stage samples, all candidates within 1e-4, class ids, detections at three settings
    (reference: models/modules.py:56-95, models/backbones.py:6-57, 183-200, models/rpns.py:155-158)."""
    from mydetection_amd import ops, synth
    from mydetection_amd.models.general import name_to_model
    m, cfg = model
    monkeypatch.setattr(ops, 'WINO4_MIN_ITEMS', 1)
    g = golden('yolov3_b1_640')
    x = synth.make_images(1, 640, seed=int(g['image_seed'])).cuda()
    with torch.no_grad():
        (c, p), n4 = _wino4_launches(lambda: (lambda c: (c, m.fpn(c)))(m.backbone(x)))
    assert n4 == 31, f'{n4} F(4x4) launches in the YOLOv3 forward, expected 31'
    for key, feats in (('backbone', c), ('fpn', p)):
        for lvl, f in enumerate(feats):
            f = f.contiguous().cpu().numpy()
            np.testing.assert_allclose(f.reshape(-1)[g[f'{key}_{lvl}_idx']], g[f'{key}_{lvl}_val'], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(np.sqrt((f.astype(np.float64) ** 2).sum()), g[f'{key}_{lvl}_l2'], rtol=1e-5)
    _check_yolo_golden(m, g, 'yolov3 640 on F(4x4)', strict='margin-permitting')
    _check_yolo_golden(m, golden('yolov3_b1_512'), 'yolov3 512 on F(4x4)', strict='margin-permitting')
    name = 'd1_fcs2_atss'
    m2, _ = name_to_model(name)
    m2.load_state_dict(synth.make_state_dict(m2.state_dict(), name), strict=True)
    m2 = m2.eval().cuda()
    g2 = golden('d1_fcs2_atss_b1_640')
    x2 = synth.make_normalized_images(1, 640, seed=int(g2['image_seed'])).cuda()
    with torch.no_grad():
        _, n4 = _wino4_launches(lambda: m2.forward_candidates(x2))
    assert n4 >= 7, f'{n4} F(4x4) launches in the D1-FCOS-ATSS forward (C6, C7 and five dense class layers expected)'
    _check_effdet_golden(name, m2, g2)


def _family_launches(fn, family):
    """Run fn() with the per-launch timer on; returns (result, number of launches of `family` it issued)."""
    from mydetection_amd import ops
    ops.TIMER = ops.KernelTimer()
    try:
        out = fn()
    finally:
        timer, ops.TIMER = ops.TIMER, None
    torch.cuda.synchronize()
    return out, len(timer.spans.get(family, []))


def test_split_bf16_kernels_vs_reference_goldens(model, golden, monkeypatch):
    """conv_igemm_b3_kernel (float32-exact three-piece bfloat16 operands on the bf16 matrix instruction; the SE-gated form for the
    project convs) against the imported reference end to end (VERDICT r05 #2c).  The dispatch rules hand a layer to it only at
    production sizes (8 192 rows and 3 GFLOP; 3 000 rows for the EfficientNet expand / gated project convs), so the batch-1
    fixtures never reach it; with the limits at 1 every eligible direct conv takes it -- YOLOv3: the 1x1 layers, the stride-2 3x3
    layers and the heads; EfficientNet-B1: every expand conv and every gated project conv with >= 64 outputs -- and the same
    reference gates apply: every candidate within 1e-4, class ids, detections at the three settings
    (reference: models/modules.py:76-95, external/efficientnet/model.py:71-98)."""
    from mydetection_amd import ops, synth
    from mydetection_amd.models.general import name_to_model
    m, cfg = model
    assert ops.SPLIT_BF16
    for k, v in (('B3_MIN_ROWS', 1), ('B3_MIN_FLOP', 0.0), ('B3_EXPAND_MIN_ROWS', 1), ('B3_GATED_MIN_ROWS', 1)):
        monkeypatch.setattr(ops, k, v)
    g = golden('yolov3_b1_640')
    x = synth.make_images(1, 640, seed=int(g['image_seed'])).cuda()
    with torch.no_grad():
        _, n3 = _family_launches(lambda: m.forward_candidates(x), 'conv_igemm_b3')
    assert n3 >= 35, f'{n3} split-bf16 launches in the YOLOv3 forward (1x1, stride-2 3x3 and head convs expected)'
    _check_yolo_golden(m, g, 'yolov3 640 on split-bf16', strict='margin-permitting')
    _check_yolo_golden(m, golden('yolov3_b1_512'), 'yolov3 512 on split-bf16', strict='margin-permitting')
    for name, least in (('efficientdet-d1', 30), ('d1_fcs2_atss', 30)):
        m2, _ = name_to_model(name)
        m2.load_state_dict(synth.make_state_dict(m2.state_dict(), name), strict=True)
        m2 = m2.eval().cuda()
        g2 = golden(name.replace('-', '_') + '_b1_640')
        x2 = synth.make_normalized_images(1, 640, seed=int(g2['image_seed'])).cuda()
        with torch.no_grad():
            _, n3 = _family_launches(lambda: m2.forward_candidates(x2), 'conv_igemm_b3')
        assert n3 >= least, f'{name}: {n3} split-bf16 launches (expand convs and gated project convs of the 23 MBConv blocks expected)'
        _check_effdet_golden(name, m2, g2)


def test_effdet_family_vs_oracle_640(effdet):
    """640x640 (benchmark resolution), batch 2, against the float32 CPU oracle (= the reference's arithmetic,
    tests/test_oracle_golden.py): every score and box within 1e-4 (rtol and atol), class ids exact wherever the
    oracle's own top-2 class gap exceeds round-off, and the post-processed detections of BOTH images identical
    (count, candidate indices, classes, order) at the AP and demo thresholds -- all four (image, threshold) pairs: the
    images are picked so that no decision of theirs sits within 1e-5 of a boundary."""
    from mydetection_amd import synth
    from mydetection_amd.utils.structures import batched_post_process
    from oracle import efficientdet as oe, postprocess as opp
    name, m, cfg = effdet
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    nms = cfg['test.nms_thres']
    # two images all of whose decisions are margin-safe under the ORACLE's own float32 scores at both thresholds (a detection
    # set is only defined then): chosen here, deterministically, from the seed sequence 7, 8, ...
    chosen = []
    for seed in range(7, 47):
        xi = synth.make_normalized_images(1, 640, seed=seed)
        with torch.no_grad():
            _, c1, s1 = oe.forward(xi, sd, name)
        if all(opp.decision_margins(s1[0].numpy(), c1[0].numpy(), conf, eps=4e-5) is None for conf in (0.005, 0.05, 0.5)):
            chosen.append(xi)
            if len(chosen) == 2:
                break
    assert len(chosen) == 2, f'{name}: no two margin-safe image seeds in 7..46'
    x = torch.cat(chosen, dim=0)
    with torch.no_grad():
        ob, oc, os_, margin = oe.forward(x, sd, name, with_margin=True)
        bb, ci, sc = m.forward_candidates(x.cuda())
    assert bb.shape == ob.shape and bb.shape[1] == {'efficientdet-d1': 76725, 'd1_yv3': 25575, 'd1_fcs2_p3': 8400}.get(name, 8525)
    np.testing.assert_allclose(sc.cpu().numpy(), os_.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(bb.cpu().numpy(), ob.numpy(), rtol=RTOL, atol=ATOL)
    for i in range(2):
        _class_ids_exact_where_safe(ci[i].cpu().numpy(), oc[i].numpy(), margin[i].numpy(), f'{name} 640 image {i}')
    flips = (ci.cpu() != oc)
    counts = []
    for conf in (0.005, 0.05, 0.5):
        rec = batched_post_process(bb, ci, sc, conf, nms)
        for i in range(2):
            rb, rc, rs, src = opp.post_process(ob[i].numpy(), oc[i].numpy(), os_[i].numpy(), conf, nms)
            k = int(rec['count'][i])
            counts.append(k)
            assert len(src) >= (50 if conf < 0.5 else 8)
            order = torch.argsort(os_[i], descending=True, stable=True)
            sel = order[os_[i][order] >= conf][:512]             # the candidates that enter NMS
            assert not flips[i][sel].any(), 'a class tie among the candidates that enter NMS'
            # (the batch-of-2 oracle forward may round differently from the solo runs that chose the images: re-check)
            assert opp.decision_margins(os_[i].numpy(), oc[i].numpy(), conf, eps=5e-6) is None
            assert k == len(src), f'{name} conf {conf} image {i}: {k} vs {len(src)} detections'
            np.testing.assert_array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
            np.testing.assert_array_equal(rec['class_idx'][i, :k].cpu().numpy(), rc)
            np.testing.assert_allclose(rec['score'][i, :k].cpu().numpy(), rs, rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(rec['bbox'][i, :k].cpu().numpy(), rb, rtol=RTOL, atol=ATOL)
    # (the two BASELINE configurations: three different sets per image; the registry compositions share the recipe but not its
    # tuning -- d1_yv3 has 25 575 candidates and more than 512 of them can pass 0.05 as well)
    if name in ('efficientdet-d1', 'd1_fcs2_atss'):
        assert counts[0] > counts[2] > counts[4] and counts[1] > counts[3] > counts[5], f'{name}: the thresholds decide the same thing: {counts}'
    else:
        assert counts[0] >= counts[2] > counts[4] and counts[1] >= counts[3] > counts[5], f'{name}: the thresholds decide the same thing: {counts}'


@pytest.mark.parametrize('name', ['efficientdet-d1', 'd1_fcs2_atss'])
def test_effdet_stiff_weights_relative_to_float64(name):
    """The second, ill-conditioned synthetic parameter set (synth recipe 'stiff': pre-activations centred on the swish's
    curved part, residual branches barely damped -- how trained weights behave): float32 round-off grows with depth and
    the float32 CPU reference itself sits ~1e-4 from an exact evaluation, so no absolute gate is meaningful.  The gate
    is relative: the HIP path (F(4x4) / F(2x2) Winograd, fused MBConv and pyramid kernels, fast sigmoid) must be as close
    to the FLOAT64 oracle as the float32 CPU path is -- max error within 3x (an extreme-value statistic), rms within
    1.5x -- and agree with the float32 oracle within the sum of the two round-off bands on 99.9 % of the scores; class ids
    exact outside the round-off band of the float64 class gap."""
    from mydetection_amd import synth
    from mydetection_amd.models.general import name_to_model
    from oracle import efficientdet as oe
    m, cfg = name_to_model(name)
    sd = synth.make_state_dict(m.state_dict(), name, recipe='stiff')
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    x = synth.make_normalized_images(2, 384, seed=13)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        ob, oc, os_ = oe.forward(x, sd, name)
        ob64, oc64, os64, margin64 = oe.forward(x.double(), sd64, name, with_margin=True)
        bb, ci, sc = m.forward_candidates(x.cuda())
    sc, bb = sc.cpu().double(), bb.cpu().double()
    bscale = 1.0 + ob64.abs() / 640.0            # box errors relative to the box scale
    err_gpu_s, err_cpu_s = (sc - os64).abs().max().item(), (os_.double() - os64).abs().max().item()
    err_gpu_b, err_cpu_b = ((bb - ob64).abs() / bscale).max().item(), ((ob.double() - ob64).abs() / bscale).max().item()
    assert err_cpu_s > 2e-5, f'{name}: the stiff recipe is supposed to amplify round-off (float32 reference error {err_cpu_s:.1e})'
    assert err_gpu_s <= max(ATOL, 3.0 * err_cpu_s), (err_gpu_s, err_cpu_s)
    assert err_gpu_b <= max(2e-3, 3.0 * err_cpu_b), (err_gpu_b, err_cpu_b)

    def rms(t):
        return t.pow(2).mean().sqrt().item()
    assert rms(sc - os64) <= 1.5 * rms(os_.double() - os64) + 1e-7, (rms(sc - os64), rms(os_.double() - os64))
    assert rms((bb - ob64) / bscale) <= 1.5 * rms((ob.double() - ob64) / bscale) + 1e-6
    # two float32 evaluations each sit up to err_cpu_s from the exact one: all but 0.1 % of the scores within their sum
    tol = max(ATOL, 2.0 * err_cpu_s)
    bad = ((sc - os_.double()).abs() > tol + RTOL * os_.double().abs()).sum().item()
    assert bad <= sc.numel() // 1000, f'{bad} of {sc.numel()} scores differ from the float32 oracle by more than {tol:.1e}'
    safe = margin64 > 4.0 * err_cpu_s                 # class gap outside what round-off moves a probability by
    assert safe.any()
    assert torch.equal(ci.cpu()[safe], oc64[safe])


@pytest.mark.parametrize('name,batch', [('efficientdet-d1', 16), ('d1_fcs2_atss', 32)])
def test_effdet_full_size_properties_640(name, batch):
    """BASELINE configs[2] and configs[3] at their benchmark size (efficientdet-d1 batch 16, d1_fcs2_atss batch 32,
    640x640: the grouped pyramid launches, split-K tails and every tile shape of the benchmark are live), through
    properties that need no CPU forward:
      * the same batch twice gives the same bits;
      * image i of the batch equals its solo run and permuting the images permutes the candidates (no cross-image
        coupling), to 1e-5;
      * the hipGraph replay of the step equals the eager records;
      * NMS invariants per image (count <= 512, score >= conf, class ascending / score descending inside a class,
        kept boxes of one class pairwise IoU <= thr, unique indices pointing at their candidates), idempotence;
      * the batched records equal the oracle's post_process on the GPU candidates for a sample of images;
      * four images of the batch, evaluated as bench.py does (two batch lanes, split-bf16 expand / gated project convs, in-launch
        squeeze-excite gates), against the CPU oracle forward: candidates 1e-4, class ids, detection sets (round 6)."""
    from mydetection_amd import synth
    from mydetection_amd.graph import GraphedPath
    from mydetection_amd.models.general import name_to_model
    from mydetection_amd.utils.bbox_ops import bboxes_iou
    from mydetection_amd.utils.structures import batched_post_process
    from oracle import postprocess as pp
    m, cfg = name_to_model(name)
    m.load_state_dict(synth.make_state_dict(m.state_dict(), name), strict=True)
    m = m.eval().cuda()
    conf, thr = cfg['test.ap_conf_thres'], cfg['test.nms_thres']
    x = synth.make_normalized_images(batch, 640, seed=13).cuda()
    perm = torch.randperm(batch, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        bb, ci, sc = m.forward_candidates(x)
        b2, c2, s2 = m.forward_candidates(x)
        bp, cp, sp = m.forward_candidates(x[perm].contiguous())
        assert torch.equal(b2, bb) and torch.equal(c2, ci) and torch.equal(s2, sc)
        # to 1e-5, not bit for bit: where a tile's K range is cut (split-K tail) depends on its position in the grid
        np.testing.assert_allclose(sp.cpu().numpy(), sc[perm].cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(bp.cpu().numpy(), bb[perm].cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert (cp != ci[perm]).float().mean().item() < 1e-4
        for i in (0, batch - 1):
            # solo run vs the same image inside the batch: two float32 evaluations with different summation orders -- 3e-5, a
            # third of north_star's tolerance.  Where they part (tools/solo_vs_batch.py, profiles/r04_solo_vs_batch.txt):
            # blocks 0 and 1 agree bit for bit; the first difference (6.7e-6 absolute) is MBConv block 2's project conv
            # 96->24 @160^2, which runs pw_skinny_kernel (k order permuted to the MFMA operand layout) once the launch has
            # M = B*160^2 >= 65 536 rows and conv_igemm_kernel (k in order) for one image (M = 25 600); blocks 3-7 make the
            # same switch, the small-grid layers of the 20^2 stages are cut along K in the batch-1 run only.  The
            # difference stays at 0.7-1.3e-5 absolute through the trunk and BiFPN (no growth: the conditioned weights), and
            # reaches the candidates as <= 2.3e-5 relative on the scores (measured: D1 image 15 of 16, FCOS image 31 of 32)
            b1, c1, s1 = m.forward_candidates(x[i:i + 1])
            np.testing.assert_allclose(s1[0].cpu().numpy(), sc[i].cpu().numpy(), rtol=3e-5, atol=1e-5)
            np.testing.assert_allclose(b1[0].cpu().numpy(), bb[i].cpu().numpy(), rtol=3e-5, atol=1e-5)
    assert bb.shape[1] == (76725 if name == 'efficientdet-d1' else 8525)
    assert torch.isfinite(sc).all() and torch.isfinite(bb).all()
    rec = batched_post_process(bb, ci, sc, conf, thr)
    graphed = GraphedPath(m, x, conf, thr, lanes=1)
    rg = graphed(x)
    for k in ('count', 'index', 'class_idx', 'score', 'bbox'):
        assert torch.equal(rg[k], rec[k]), k
    cnt = rec['count'].cpu().numpy()
    assert (cnt <= 512).all() and (cnt >= 50).all()
    for i in range(batch):
        k = int(cnt[i])
        idx = rec['index'][i, :k].long()
        cls, s, b = rec['class_idx'][i, :k], rec['score'][i, :k], rec['bbox'][i, :k]
        assert idx.unique().numel() == k
        assert torch.equal(b, bb[i][idx]) and torch.equal(cls, ci[i][idx]) and torch.equal(s, sc[i][idx])
        assert (s >= conf).all()
        same = cls[1:] == cls[:-1]
        assert (cls[1:] >= cls[:-1]).all() and (s[1:][same] <= s[:-1][same]).all()
        iou = bboxes_iou(b, b, xyxy=False)
        clash = (iou > thr + 1e-5) & (cls[:, None] == cls[None, :])
        clash.fill_diagonal_(False)
        assert not clash.any()
        again = batched_post_process(b[None], cls[None], s[None], conf, thr)
        assert int(again['count'][0]) == k and torch.equal(again['bbox'][0, :k], b)
    for i in (0, batch // 2, batch - 1):
        ob, oc, os_, src = pp.post_process(bb[i].cpu().numpy(), ci[i].cpu().numpy(), sc[i].cpu().numpy(), conf, thr)
        k = int(cnt[i])
        assert k == len(src)
        np.testing.assert_array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
    # Four images of THIS production batch against the CPU oracle forward (VERDICT r05 #2b), evaluated the way bench.py runs the
    # step: two batch lanes replayed from one hipGraph, so the lane-sized dispatch is live -- split-bf16 expand convs, gated
    # split-bf16 project convs, in-launch squeeze-excite gates (asserted below).  Every candidate within north_star's 1e-4, class
    # ids exact where the oracle's two best classes are further apart than round-off; at each of three thresholds the kept
    # candidates agree with the oracle's detections to >= 97 % (Jaccard), and exactly -- count, ids, classes, order -- on every
    # (image, threshold) pair whose decisions are all further than twice the observed score error from flipping: at least three.
    from mydetection_amd import ops
    from oracle import efficientdet as oe
    lane = batch // 2
    assert ops.SPLIT_BF16 and ops.SE_IN_DW
    assert ops.b3_takes(lane * 400, 192, 1152, 1, ops.B3_EXPAND_MIN_ROWS)                                     # expand conv, 20^2 stage
    assert ops.b3_takes(lane * 400, 1152, 192, 1, ops.B3_GATED_MIN_ROWS, min_cout=ops.B3_GATED_MIN_COUT)       # gated project conv
    run = GraphedPath(m, x, conf, thr, lanes=2)
    assert run.lanes == 2
    run()
    torch.cuda.synchronize()
    lb, lc, ls = (t.clone() for t in run.cand)
    np.testing.assert_allclose(ls.cpu().numpy(), sc.cpu().numpy(), rtol=3e-5, atol=1e-5)         # lanes vs one stream: solo-vs-batch kind
    sd_cpu = {k: v.cpu() for k, v in m.state_dict().items()}
    pick = [0, lane - 1, lane, batch - 1]
    with torch.no_grad():
        ob, oc, os_, margin = oe.forward(x[pick].cpu(), sd_cpu, name, with_margin=True)
    got_s, got_b, got_c = ls[pick].cpu().numpy(), lb[pick].cpu().numpy(), lc[pick].cpu().numpy()
    np.testing.assert_allclose(got_s, os_.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(got_b, ob.numpy(), rtol=RTOL, atol=ATOL)
    err = float(np.abs(got_s - os_.numpy()).max())
    n_safe, worst, kept = 0, 1.0, []
    for j, i in enumerate(pick):
        _class_ids_exact_where_safe(got_c[j], oc[j].numpy(), margin[j].numpy(), f'{name} batch-{batch} image {i}')
        for t in (conf, 0.05, 0.5):
            _, rc, _, ri = pp.post_process(ob[j].numpy(), oc[j].numpy(), os_[j].numpy(), t, thr)
            r_t = batched_post_process(lb[i:i + 1], lc[i:i + 1], ls[i:i + 1], t, thr)
            k = int(r_t['count'][0])
            kept.append(k)
            mine = r_t['index'][0, :k].cpu().numpy().astype(np.int64)
            assert k >= 8 and len(ri) >= 8, f'image {i} conf {t}: vacuous ({k} / {len(ri)} detections)'
            jac = len(np.intersect1d(mine, ri)) / len(np.union1d(mine, ri))
            worst = min(worst, jac)
            assert jac >= 0.97, f'image {i} conf {t}: kept candidates agree to {jac:.3f} only ({k} vs {len(ri)})'
            if pp.decision_margins(os_[j].numpy(), oc[j].numpy(), t, eps=max(2.0 * err, 2e-6)) is None:
                n_safe += 1
                assert k == len(ri), f'image {i} conf {t}: {k} vs {len(ri)} detections'
                np.testing.assert_array_equal(mine, ri)
                np.testing.assert_array_equal(r_t['class_idx'][0, :k].cpu().numpy(), rc)
    print(f'{name} batch {batch} (two lanes) vs oracle: score error {err:.1e}, {n_safe} of {3 * len(pick)} pairs margin-safe and exact, '
          f'worst Jaccard {worst:.4f}, detections per (image, threshold) {kept}')
    assert all(kept[3 * j] > kept[3 * j + 1] > kept[3 * j + 2] for j in range(len(pick))), f'the thresholds decide the same thing: {kept}'
    assert n_safe >= 3, f'only {n_safe} of {3 * len(pick)} (image, threshold) pairs are margin-safe at score error {err:.1e}'


def test_device_preprocess_and_batched_detector(model):
    """Row 8f: pad + to_tensor + normalise on the device (bit-exact vs the reference's host arithmetic) and the
    batched Detector.predict_batch == per-image detect_one."""
    import PIL.Image
    from mydetection_amd import ops, synth
    from mydetection_amd.api import Detector
    from mydetection_amd.utils import image_ops
    m, cfg = model
    rng = np.random.Generator(np.random.PCG64(21))
    u8 = rng.integers(0, 256, size=(2, 37, 53, 3), dtype=np.uint8)
    for fmt in ('RGB_1', 'RGB_1_norm'):
        out = ops.preprocess_u8(torch.from_numpy(u8).cuda(), (64, 64), fmt).cpu()
        for b in range(2):
            pil = image_ops.pad_to_divisible(PIL.Image.fromarray(u8[b]), 64)          # zero pad (uint8), then
            ref = image_ops.format_tensor_img(image_ops.to_tensor(pil), fmt)          # /255, normalise
            assert torch.equal(out[b], ref), fmt
    det = Detector(model_and_cfg=(m, cfg))
    imgs = [PIL.Image.fromarray((synth.make_images(1, (240, 320), seed=30 + i)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8))
            for i in range(3)]
    kw = dict(preprocessing='resize_pad_square', input_size=320, conf_thres=0.005)
    batch = det.predict_batch(imgs, **kw)
    assert len(batch) == 3
    for img, d in zip(imgs, batch):
        one = det.detect_one(pil_img=img, **kw)
        assert len(one) == len(d) > 0 and d.img_hw == (240, 320)
        # to 2e-5, not bit for bit: a batch of 3 and a batch of 1 cut their small grids along K differently, and the
        # calibrated head turns 1e-6 of feature round-off into 1e-5 of score (gain 4 on the features' spatial variation)
        assert torch.equal(one.cats, d.cats)
        np.testing.assert_allclose(one.scores.cpu().numpy(), d.scores.cpu().numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(one.bboxes.cpu().numpy(), d.bboxes.cpu().numpy(), rtol=1e-5, atol=1e-4)
    # the second call with a shape is captured into a hipGraph, later ones replay it: same detections, bit for bit
    assert det.use_graph and any(k[0][0] == 1 for k in det._graphs.graphs), 'three detect_one calls: the batch-of-1 path is a graph by now'
    for _ in range(3):
        again = det.predict_batch(imgs, **kw)
        for d, e in zip(batch, again):
            assert torch.equal(d.cats, e.cats) and torch.equal(d.scores, e.scores) and torch.equal(d.bboxes, e.bboxes)
    assert any(k[0][0] == 3 for k in det._graphs.graphs) and len(det._graphs.graphs) == 2
    det.use_graph = False
    eager = det.predict_batch(imgs, **kw)
    for d, e in zip(batch, eager):
        assert torch.equal(d.cats, e.cats) and torch.equal(d.scores, e.scores) and torch.equal(d.bboxes, e.bboxes)


def test_graph_survives_workspace_growth_and_weight_reload(model, monkeypatch):
    """A captured hipGraph replays raw addresses: (1) when a larger layer makes ops.wino4_workspace swap its buffer
    for a bigger one, the superseded buffer must stay allocated while a graph that recorded it can still be replayed
    -- a replay must give the eager records and must not touch tensors allocated after the growth; (2) a graph
    captured before load_state_dict must not be replayed with the old kernel-ready weights."""
    from mydetection_amd import ops, synth
    from mydetection_amd.api import Detector
    from mydetection_amd.utils.structures import batched_post_process
    m, cfg = model
    monkeypatch.setattr(ops, 'WINO4_MIN_ITEMS', 1)            # every 3x3 layer with Cin >= 64 on the F(4x4) pair -> uses the workspace
    dev = torch.device('cuda', torch.cuda.current_device())
    for k in [k for k in ops._WINO4_WS if k[:2] == (dev.type, dev.index)]:
        ops._WINO4_WS.pop(k)                                  # start from no workspace: the small shape sizes it
    det = Detector(model_and_cfg=(m, cfg))
    conf, nms = 0.005, 0.45
    small = synth.make_images(1, 128, seed=5).cuda()
    large = synth.make_images(2, 256, seed=6).cuda()
    with torch.no_grad():
        eager_small = {k: v.clone() for k, v in batched_post_process(*m.forward_candidates(small), conf, nms).items()}
    assert int(eager_small['count'][0]) > 0
    det._records(small, conf, nms)                             # seen once
    first = det._records(small, conf, nms)                     # captured + replayed
    assert len(det._graphs.graphs) == 1
    g = next(iter(det._graphs.graphs.values()))
    ws_small = ops._WINO4_WS[ops._scratch_key(dev)]
    assert any(t.data_ptr() == ws_small.data_ptr() for t in g._held[0])
    for k in ('count', 'bbox', 'score', 'class_idx', 'index'):
        assert torch.equal(first[k], eager_small[k]), k
    with torch.no_grad():                                      # a larger input, eagerly: the workspace is replaced
        m.forward_candidates(large)
    ws_large = ops._WINO4_WS[ops._scratch_key(dev)]
    assert ws_large.numel() > ws_small.numel() and ws_large.data_ptr() != ws_small.data_ptr()
    ptr_small, n_small = ws_small.data_ptr(), ws_small.numel()
    del ws_small
    torch.cuda.synchronize()
    # bystanders: fresh allocations of the superseded buffer's size class; had that block gone back to the caching
    # allocator, one of these would sit on it and the replay's transform launch would scribble over it
    bystanders = [torch.full((n_small,), 7.25, dtype=torch.float32, device=dev) for _ in range(4)]
    assert all(b.data_ptr() != ptr_small for b in bystanders)
    again = det._records(small, conf, nms)                     # replay of the graph captured with the OLD workspace
    torch.cuda.synchronize()
    for k in ('count', 'bbox', 'score', 'class_idx', 'index'):
        assert torch.equal(again[k], eager_small[k]), k
    for b in bystanders:
        assert bool((b == 7.25).all())
    # (2) new weights: the cached graph is dropped, the records follow the new parameters
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd2 = dict(sd)
    key = 'rpn.heads.conv_0.bias'
    sd2[key] = sd[key] + 0.5
    m.load_state_dict(sd2, strict=True)
    try:
        assert g.stale()
        with torch.no_grad():
            eager_new = batched_post_process(*m.forward_candidates(small), conf, nms)
        got = det._records(small, conf, nms)
        assert g not in det._graphs.graphs.values()
        assert not torch.equal(eager_new['score'], eager_small['score'])
        for k in ('count', 'bbox', 'score', 'class_idx', 'index'):
            assert torch.equal(got[k], eager_new[k]), k
    finally:
        m.load_state_dict(sd, strict=True)
        for k in [k for k in ops._WINO4_WS if k[:2] == (dev.type, dev.index)]:
            ops._WINO4_WS.pop(k)                              # keys are (type, index, lane, host thread)


def test_postprocess_flags_out_of_range_class_ids():
    """Class ids outside the 12 bits of the NMS sort key must not alias silently: the image's count becomes -1 and the
    host readers raise (include/mydet.h: MYDET_COUNT_BAD_CLASS)."""
    from mydetection_amd import ops
    from mydetection_amd.parallel import records_to_objects
    rng = np.random.Generator(np.random.PCG64(3))
    n = 300
    bb = torch.from_numpy(rng.uniform(10, 200, size=(2, n, 4)).astype(np.float32)).cuda()
    sc = torch.from_numpy(rng.uniform(0.1, 1.0, size=(2, n)).astype(np.float32)).cuda()
    ci = torch.from_numpy(rng.integers(0, 80, size=(2, n))).cuda()
    ci[1, 17] = 4096 + 3                                       # would alias to class 3
    rec = ops.postprocess(bb, ci, sc, 0.05, 0.45)
    cnt = rec['count'].cpu().tolist()
    assert cnt[0] > 0 and cnt[1] == -1
    assert not bool(rec['bbox'][1].any()) and not bool(rec['score'][1].any())
    with pytest.raises(ValueError):
        records_to_objects(rec)
    ci[1, 17] = -5
    assert ops.postprocess(bb, ci, sc, 0.05, 0.45)['count'].cpu().tolist()[1] == -1
    sc[1, 17] = 0.0                                            # below the threshold: never selected, never looked at
    assert ops.postprocess(bb, ci, sc, 0.05, 0.45)['count'].cpu().tolist()[1] > 0


def test_full_size_properties_batch32_640(model):
    """BASELINE configs[1] size (batch 32, 640x640; the stream-K Winograd schedule and every conv tile shape of the
    benchmark are live here), through properties that need no CPU forward:
      * the same batch twice gives the same bits (the split-K / stream-K schedules sum in a fixed order, no atomics);
      * permuting the images permutes the candidates (no cross-image coupling) -- scores to 3e-5, boxes to 5e-5, not bit for bit:
        where a tile's K range is cut depends on the tile's position in the schedule;
      * the Winograd layers agree with the same network on the direct implicit-GEMM kernel within the tolerance;
      * NMS output invariants per image: count <= 512, score >= conf, class ascending / score descending inside a
        class, kept boxes of one class pairwise IoU <= thr, every kept index unique and pointing at its candidate;
      * post-processing is idempotent on its own output (re-running filter + NMS on the kept set keeps it all);
      * the batched records equal the oracle's post_process on the GPU candidates for a sample of images."""
    from mydetection_amd import ops, synth
    from mydetection_amd.utils.structures import batched_post_process
    from mydetection_amd.utils.bbox_ops import bboxes_iou
    from oracle import postprocess as pp
    m, cfg = model
    conf, thr = 0.005, 0.45
    x = synth.make_images(32, 640, seed=11).cuda()
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        bb, ci, sc = m.forward_candidates(x)
        b2, c2, s2 = m.forward_candidates(x)
        bp, cp, sp = m.forward_candidates(x[perm].contiguous())
        assert ops.WINOGRAD
        # (the patch-resident 3x3 kernel is live in this batch: the first stride-2 layers and the 32 -> 64 layer of the first DarkBlock)
        assert ops.p3_takes(32, 320, 320, 32, 64, 3, 2, (1, 1, 1, 1)) and ops.p3_takes(32, 320, 320, 32, 64, 3, 1, (1, 1, 1, 1))
        assert ops.p3_takes(32, 40, 40, 256, 512, 3, 2, (1, 1, 1, 1)) and ops.p3_takes(32, 20, 20, 512, 1024, 3, 2, (1, 1, 1, 1))     # with strip tiles
        ops.WINOGRAD = False
        try:
            bd, cd, sd_ = m.forward_candidates(x)
        finally:
            ops.WINOGRAD = True
    assert bb.shape == (32, 25200, 4)
    assert torch.equal(b2, bb) and torch.equal(c2, ci) and torch.equal(s2, sc)
    np.testing.assert_allclose(sp.cpu().numpy(), sc[perm].cpu().numpy(), rtol=1e-4, atol=3e-5)      # (head gain: see synth._YOLO_TARGETS)
    np.testing.assert_allclose(bp.cpu().numpy(), bb[perm].cpu().numpy(), rtol=5e-5, atol=5e-5)
    assert (cp != ci[perm]).float().mean().item() < 1e-4
    np.testing.assert_allclose(sc.cpu().numpy(), sd_.cpu().numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(bb.cpu().numpy(), bd.cpu().numpy(), rtol=RTOL, atol=ATOL)
    assert (ci != cd).float().mean().item() < 1e-4          # argmax flips only between numerically tied classes

    rec = batched_post_process(bb, ci, sc, conf, thr)
    cnt = rec['count'].cpu().numpy()
    assert (cnt <= 512).all() and cnt.sum() > 0
    for i in range(32):
        k = int(cnt[i])
        idx = rec['index'][i, :k].long()
        cls, s, b = rec['class_idx'][i, :k], rec['score'][i, :k], rec['bbox'][i, :k]
        assert idx.unique().numel() == k
        assert torch.equal(b, bb[i][idx]) and torch.equal(cls, ci[i][idx]) and torch.equal(s, sc[i][idx])
        assert (s >= conf).all()
        same = cls[1:] == cls[:-1]
        assert (cls[1:] >= cls[:-1]).all() and (s[1:][same] <= s[:-1][same]).all()
        if k > 1:
            iou = bboxes_iou(b, b, xyxy=False)
            clash = (iou > thr + 1e-5) & (cls[:, None] == cls[None, :])      # margin: NMS forms the IoU in xyxy
            clash.fill_diagonal_(False)
            assert not clash.any()
        again = batched_post_process(b[None], cls[None], s[None], conf, thr)
        assert int(again['count'][0]) == k and torch.equal(again['bbox'][0, :k], b)
    for i in (0, 13, 31):
        ob, oc, os_, src = pp.post_process(bb[i].cpu().numpy(), ci[i].cpu().numpy(), sc[i].cpu().numpy(), conf, thr)
        k = int(cnt[i])
        assert k == len(src)
        np.testing.assert_array_equal(rec['index'][i, :k].cpu().numpy().astype(np.int64), src)
    # sixteen images of THIS production batch (F(4x4) on 31 layers, K-cut tails, the 128 x 128 tiles) against the CPU oracle
    # forward: every candidate within north_star's 1e-4, class ids exact where the oracle's two best classes are further
    # apart than round-off; at each of three thresholds the kept candidates agree with the oracle's own detections to
    # >= 97 % (Jaccard index of the kept candidate ids: what differs are decisions inside the round-off band), and
    # exactly -- count, ids, classes, order -- on every (image, threshold) pair whose decisions are all further than
    # twice the observed score error from flipping (oracle.postprocess.decision_margins)
    from oracle import yolov3 as oy
    sd_cpu = {k: v.cpu() for k, v in m.state_dict().items()}
    pick = list(range(0, 32, 2))        # sixteen images (~4 s of oracle forward on the box's 16 threads): 48 (image, threshold) pairs
    with torch.no_grad():
        ob, oc, os_, raws = oy.forward(x[pick].cpu(), sd_cpu, return_raw=True)
    got_s, got_b, got_c = sc[pick].cpu().numpy(), bb[pick].cpu().numpy(), ci[pick].cpu().numpy()
    np.testing.assert_allclose(got_s, os_.numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(got_b, ob.numpy(), rtol=RTOL, atol=ATOL)
    n_p = len(pick)
    margin = []
    for r in raws:
        t = torch.sigmoid(r.view(n_p, 3, 85, *r.shape[2:])[:, :, 5:].permute(0, 1, 3, 4, 2)).reshape(n_p, -1, 80).topk(2, dim=-1).values
        margin.append(t[..., 0] - t[..., 1])
    margin = torch.cat(margin, dim=1).numpy()
    err = float(np.abs(got_s - os_.numpy()).max())
    n_safe, worst = 0, 1.0
    for j, i in enumerate(pick):
        _class_ids_exact_where_safe(got_c[j], oc[j].numpy(), margin[j], f'batch-32 image {i}')
        for t in (conf, 0.05, 0.5):
            _, rc, _, ri = pp.post_process(ob[j].numpy(), oc[j].numpy(), os_[j].numpy(), t, thr)
            r_t = batched_post_process(bb[i:i + 1], ci[i:i + 1], sc[i:i + 1], t, thr)
            k = int(r_t['count'][0])
            mine = r_t['index'][0, :k].cpu().numpy().astype(np.int64)
            assert k >= 20 and len(ri) >= 20, f'image {i} conf {t}: vacuous ({k} / {len(ri)} detections)'
            jac = len(np.intersect1d(mine, ri)) / len(np.union1d(mine, ri))
            worst = min(worst, jac)
            assert jac >= 0.97, f'image {i} conf {t}: kept candidates agree to {jac:.3f} only ({k} vs {len(ri)})'
            if pp.decision_margins(os_[j].numpy(), oc[j].numpy(), t, eps=max(2.0 * err, 2e-6)) is None:
                n_safe += 1
                assert k == len(ri), f'image {i} conf {t}: {k} vs {len(ri)} detections'
                np.testing.assert_array_equal(mine, ri)
                np.testing.assert_array_equal(r_t['class_idx'][0, :k].cpu().numpy(), rc)
    print(f'batch-32 vs oracle: score error {err:.1e}, {n_safe} of {3 * n_p} pairs margin-safe and exact, worst Jaccard {worst:.4f}')
    assert n_safe >= 3, f'only {n_safe} of {3 * n_p} (image, threshold) pairs are margin-safe at score error {err:.1e}'


def test_hipgraph_replay_equals_eager(model):
    """GraphedPath (one hipGraph for forward + post-process, incl. the split-K / stream-K workspace traffic and the
    fixup launches): replays equal the eager records bit for bit, for the captured batch and for a new one."""
    from mydetection_amd import synth
    from mydetection_amd.graph import GraphedPath
    from mydetection_amd.utils.structures import batched_post_process
    m, cfg = model
    x0 = synth.make_images(4, 320, seed=21).cuda()
    x1 = synth.make_images(4, 320, seed=22).cuda()
    run = GraphedPath(m, x0, 0.005, 0.45, lanes=1)
    for x in (x0, x1, x0):
        rec = {k: v.clone() for k, v in run(x).items()}
        with torch.no_grad():
            ref = batched_post_process(*m.forward_candidates(x), 0.005, 0.45)
        assert int(ref['count'].sum()) > 0
        for k in ('count', 'index', 'class_idx', 'score', 'bbox'):
            assert torch.equal(rec[k], ref[k]), k


def test_detector_lanes_same_bits_before_and_after_capture(monkeypatch):
    """api.Detector on an EfficientNet-based model evaluates an even batch as two half batches (batch lanes): the eager calls
    that precede the capture run the halves one after the other, the replays run them as parallel graph branches -- the
    detections of the first (eager), second (captured) and later (replayed) calls are bit-identical."""
    import PIL.Image
    from mydetection_amd import synth
    from mydetection_amd.api import Detector
    from mydetection_amd.models.general import name_to_model
    monkeypatch.delenv('MYDET_LANES', raising=False)           # the rule under test is the default one
    m, cfg = name_to_model('efficientdet-d1')
    m.load_state_dict(synth.make_state_dict(m.state_dict(), 'efficientdet-d1'), strict=True)
    m = m.eval().cuda()
    assert m.batch_lanes_hint == 2
    det = Detector(model_and_cfg=(m, cfg))
    assert det.batch_lanes(4) == 2 and det.batch_lanes(3) == 1 and det.batch_lanes(1) == 1
    imgs = [PIL.Image.fromarray((synth.make_images(1, (256, 256), seed=60 + i)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8))
            for i in range(4)]
    kw = dict(preprocessing='resize_pad_square', input_size=256, conf_thres=0.005)
    runs = [det.predict_batch(imgs, **kw) for _ in range(4)]
    assert det.use_graph and len(det._graphs.graphs) == 1 and next(iter(det._graphs.graphs.values())).lanes == 2
    assert all(len(d) > 0 for d in runs[0])
    for later in runs[1:]:
        for d, e in zip(runs[0], later):
            assert torch.equal(d.cats, e.cats) and torch.equal(d.scores, e.scores) and torch.equal(d.bboxes, e.bboxes)


@pytest.mark.parametrize('variant', ['plain', 'saturated_and_tied'])
def test_fused_retina_decode_equals_two_launch_path(monkeypatch, variant):
    """EfDetHead + RetinaLayer: the decode in the epilogue of the towers' last layers (ops.sepconv_decode_retina: no class
    logits in memory) gives the candidates of the two-launch path (last sepconv writes logits, decode kernel reads
    them) bit for bit -- boxes, scores, class indices -- also with logits in the saturated range of the float32
    sigmoid (>= 5: sigmoid values decide; >= 17.4: all equal 1.0, the first class wins) and with exactly tied classes."""
    from mydetection_amd import ops, synth
    from mydetection_amd.models.general import name_to_model
    m, cfg = name_to_model('efficientdet-d1')
    sd = synth.make_state_dict(m.state_dict(), 'efficientdet-d1')
    if variant == 'saturated_and_tied':
        for lvl in range(5):
            wk, bk = f'rpn.class_nets.{lvl}.3.pointwise.weight', f'rpn.class_nets.{lvl}.3.pointwise.bias'
            w, b = sd[wk].clone(), sd[bk].clone()
            A, C = 9, cfg['general.num_class']
            w, b = w.view(A, C, -1), b.view(A, C)
            w[0, 7], b[0, 7] = w[0, 3], b[0, 3]            # anchor 0: classes 3 and 7 are the same function
            b[0, 3] += 3.0
            b[0, 7] += 3.0                                  # ... and usually the maximum
            b[1, 70:80] = 16.0                              # anchor 1: several classes above 5 (absolute: the recipe's own class bias is -7 .. -23 by level)
            b[2, 5] = 45.0
            b[2, 60] = 46.0                                 # anchor 2: two classes at sigmoid == 1.0: class 5 wins
            w[3, 70], b[3, 70] = w[3, 20], b[3, 20]         # anchor 3: a tie between lanes (classes 20 and 70)
            b[3, 20] = 12.0
            b[3, 70] = 12.0
            sd[wk], sd[bk] = w.view(A * C, -1, 1, 1), b.view(-1)
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    x = synth.make_normalized_images(3, 384, seed=77).cuda()
    assert m.rpn.can_decode_retina(m.det_layers)
    with torch.no_grad():
        bb, ci, sc = m.forward_candidates(x)
        monkeypatch.setattr(ops, 'FUSED_DECODE', False)
        assert not m.rpn.can_decode_retina(m.det_layers)
        bb2, ci2, sc2 = m.forward_candidates(x)
    assert bb.shape == bb2.shape == (3, 9 * (48 * 48 + 24 * 24 + 12 * 12 + 6 * 6 + 3 * 3), 4)
    assert torch.equal(sc, sc2) and torch.equal(ci, ci2) and torch.equal(bb, bb2)
    if variant == 'saturated_and_tied':
        n3 = 48 * 48                                        # level 0, anchors 0..3
        a0, a1, a2, a3 = (ci[:, k * n3:(k + 1) * n3] for k in range(4))
        assert (a0 == 3).float().mean() > 0.2 and not (a0 == 7).any()
        assert ((a1 >= 70) & (a1 < 80)).all()
        assert (a2 == 5).all() and bool((sc[:, 2 * n3:3 * n3] == 1.0).all())
        assert (a3 == 20).float().mean() > 0.5 and not (a3 == 70).any()


@pytest.mark.parametrize('name', ['yolov3_80', 'efficientdet-d1'])
def test_hipgraph_batch_lanes(name):
    """Batch lanes (GraphedPath(lanes=2): the two halves of the batch as parallel branches of one hipGraph, each with
    its own scratch): a replay equals the same decomposition issued from the host bit for bit -- also after a replay
    with other images, i.e. no lane reads the other's scratch --, equals the two half batches run one after the other
    on the default stream, and stays within the solo-vs-batch float tolerance of the full-batch pass.  'auto' times
    both captures and keeps one of them."""
    from mydetection_amd import synth
    from mydetection_amd.graph import GraphedPath
    from mydetection_amd.models.general import name_to_model
    from mydetection_amd.utils.structures import batched_post_process
    m, cfg = name_to_model(name)
    m.load_state_dict(synth.make_state_dict(m.state_dict(), name), strict=True)
    m = m.eval().cuda()
    conf, thr = cfg['test.ap_conf_thres'], cfg['test.nms_thres']
    mk = synth.make_normalized_images if name == 'efficientdet-d1' else synth.make_images
    x0, x1 = mk(4, 384, seed=31).cuda(), mk(4, 384, seed=32).cuda()
    run = GraphedPath(m, x0, conf, thr, lanes=2)
    assert run.lanes == 2
    for x in (x0, x1, x0):
        rec = {k: v.clone() for k, v in run(x).items()}
        cand = run.cand
        ref = {k: v.clone() for k, v in run.eager(x).items()}
        with torch.no_grad():
            halves = [batched_post_process(*m.forward_candidates(h), conf, thr) for h in x.chunk(2)]
            full = m.forward_candidates(x)
        assert int(ref['count'].sum()) > 0
        for k in ('count', 'index', 'class_idx', 'score', 'bbox'):
            assert torch.equal(rec[k], ref[k]), k
            assert torch.equal(rec[k], torch.cat([h[k] for h in halves])), k
        np.testing.assert_allclose(cand[2].cpu().numpy(), full[2].cpu().numpy(), rtol=3e-5, atol=1e-5)
        np.testing.assert_allclose(cand[0].cpu().numpy(), full[0].cpu().numpy(), rtol=3e-5, atol=1e-5)
    auto = GraphedPath(m, x0, conf, thr, lanes='auto')
    assert auto.lanes in (1, 2) and auto.tuned_ms > 0
    a = auto(x1)
    b = auto.eager(x1)
    for k in ('count', 'index', 'class_idx', 'score', 'bbox'):
        assert torch.equal(a[k], b[k]), k
    odd = GraphedPath(m, x0[:3].contiguous(), conf, thr, lanes='auto')
    assert odd.lanes == 1


@pytest.mark.parametrize('name,batch', [('efficientdet-d1', 16), ('d1_fcs2_atss', 32)])
def test_two_lane_replays_repeatable_full_size(name, batch):
    """BASELINE configs[2] / [3] exactly as bench.py runs them (640x640, two batch lanes, split-bf16 expand convs from 3 000 rows,
    squeeze-excite gate finished inside the depthwise launch): 60 replays give the same candidates bit for bit, a replay equals the
    host-launched decomposition, and both stay within the solo-vs-batch tolerance of the one-stream full-batch pass.
    This is the configuration in which round 5 found gates off by up to 0.2 in a third of the wide blocks of EVERY replay (the
    other lane's bfloat16-MFMA convs on the same CUs as the finishing workgroup's expand conv: csrc/se_tail.h) -- invisible to
    every single-stream test, to the 384^2 lane test above and, by luck of timing, to bench.py's parity check until then."""
    from mydetection_amd import ops, synth
    from mydetection_amd.graph import GraphedPath
    from mydetection_amd.models.general import name_to_model
    m, cfg = name_to_model(name)
    m.load_state_dict(synth.make_state_dict(m.state_dict(), name), strict=True)
    m = m.eval().cuda()
    conf, thr = cfg['test.ap_conf_thres'], cfg['test.nms_thres']
    x = synth.make_normalized_images(batch, 640, seed=13).cuda()
    assert ops.SE_IN_DW and ops.b3_takes(batch // 2 * 400, 192, 1152, 1, ops.B3_EXPAND_MIN_ROWS)      # the combination is live
    run = GraphedPath(m, x, conf, thr, lanes=2)
    run()
    torch.cuda.synchronize()
    first = [t.clone() for t in run.cand]
    for _ in range(60):
        run()
        torch.cuda.synchronize()
        for a, b in zip(run.cand, first):
            assert torch.equal(a, b)
    rec = {k: v.clone() for k, v in run().items()}
    ref = run.eager()
    for k in ('count', 'index', 'class_idx', 'score', 'bbox'):
        assert torch.equal(rec[k], ref[k]), k
    assert ops.se_tail_timeouts(x.device) == 0            # no finishing workgroup ever gave up its poll (it would write a NaN gate)
    with torch.no_grad():
        full = m.forward_candidates(x)
    # (a lane of B / 2 images and the full batch take different kernels for some layers -- row limits of the split-bf16 forms --, so
    # this is the solo-vs-batch kind of difference: float32 round-off through the network, 4e-5 observed on one box of 4.9 M boxes)
    np.testing.assert_allclose(first[2].cpu().numpy(), full[2].cpu().numpy(), rtol=3e-5, atol=1e-5)
    np.testing.assert_allclose(first[0].cpu().numpy(), full[0].cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_device_preprocessing_vs_reference_golden(model, golden):
    """Row 8f rank 1: resize (PIL-exact) + zero pad + /255 + normalise as HIP kernels on the uint8 image, against the
    tensor the reference's own chain (api/detection.py:158-163,177-205 on utils/image_ops.py) builds: bit for bit."""
    import PIL.Image
    from mydetection_amd.api import Detector
    m, cfg = model
    g = golden('preprocess_json')
    det = Detector(model_and_cfg=(m, cfg))
    for i in range(int(g['n_cases'])):
        mode, fmt, size, div = str(g[f'c{i}_mode']), str(g[f'c{i}_fmt']), int(g[f'c{i}_size']) or None, int(g[f'c{i}_div'])
        det.divisibe = div
        m.input_format = fmt
        try:
            (idxs, x, pads, hws), = det.preprocess_batch([PIL.Image.fromarray(g[f'c{i}_image'])], preprocessing=mode, input_size=size)
        finally:
            m.input_format = cfg['general.input_format']
        assert torch.equal(x[0].cpu(), torch.from_numpy(g[f'c{i}_tensor'])), (i, mode)
        want = g[f'c{i}_pad_info']
        assert (pads[0] is None and want.size == 0) or list(pads[0]) == list(want)


def test_to_json_vs_reference_golden(golden):
    """Row 8f rank 2: ImageObjects.to_json / batched_to_json numbers come from one HIP launch in the reference's
    double arithmetic: equal to the reference's json rows exactly (bbox doubles, scores, category ids)."""
    from mydetection_amd.utils.structures import ImageObjects, batched_to_json
    from mydetection_amd import parallel
    g = golden('preprocess_json')
    bb, cats, sc = torch.from_numpy(g['json_bboxes']), torch.from_numpy(g['json_cats']), torch.from_numpy(g['json_scores'])
    d = ImageObjects(bb.cuda(), cats.cuda(), None, sc.cuda(), 'cxcywh', (512, 512))
    for tag, kw in (('coco', {}), ('table', {'catIdx2id': [int(v) for v in g['json_table']]}),
                    ('table', {'catIdx2id': {i: int(v) for i, v in enumerate(g['json_table'])}})):
        js = d.to_json(img_id=5, **kw)
        assert len(js) == len(cats) and set(js[0]) == {'image_id', 'category_id', 'bbox', 'score'} and js[0]['image_id'] == 5
        assert np.array_equal(np.array([r['bbox'] for r in js]), g[f'json_{tag}_bbox'])
        assert np.array_equal(np.array([r['score'] for r in js]), g[f'json_{tag}_score'])
        assert [r['category_id'] for r in js] == list(g[f'json_{tag}_cat'])
    names = d.to_json(img_id='a', catIdx2id={i: f'c{i}' for i in range(80)})           # non-integer ids: host mapping
    assert [r['category_id'] for r in names] == [f'c{int(c)}' for c in cats]
    # batched: two images' records (second one empty)
    n = len(cats)
    rec = parallel.make_records({'count': torch.tensor([n, 0], dtype=torch.int32), 'bbox': torch.zeros(2, 512, 4),
                                 'score': torch.zeros(2, 512), 'class_idx': torch.zeros(2, 512, dtype=torch.int64),
                                 'index': torch.zeros(2, 512, dtype=torch.int32)})
    rec['bbox'][0, :n], rec['score'][0, :n], rec['class_idx'][0, :n] = bb, sc, cats
    rec = parallel.record_views(rec['records'].cuda())
    js = batched_to_json(rec, [7, 8])
    assert len(js) == n and all(r['image_id'] == 7 for r in js)
    assert np.array_equal(np.array([r['bbox'] for r in js]), g['json_coco_bbox'])


def test_batched_evaluation_predict_equals_per_image(model, tmp_path):
    """Row 8f rank 3: evaluation_predict (batched: device preprocessing, one forward per input size, batched post-process,
    to_original and json on the device) returns exactly the rows of the reference's per-image loop
    (detect_one + to_json, api/detection.py:67-74), in the same order, for images of different sizes."""
    import PIL.Image
    from mydetection_amd import synth
    from mydetection_amd.api import Detector
    m, cfg = model
    det = Detector(model_and_cfg=(m, cfg))
    sizes = [(240, 320), (200, 200), (240, 320), (180, 300), (200, 200)]
    infos = []
    for i, hw in enumerate(sizes):
        img = (synth.make_images(1, hw, seed=40 + i)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
        PIL.Image.fromarray(img).save(tmp_path / f'{100 + i}.png')
        infos.append({'file_name': f'{100 + i}.png', 'id': 100 + i})
    eval_info = {'image_dir': str(tmp_path), 'image_info': {'images': infos}, 'eval_type': 'x1y1wh'}
    for kw in (dict(preprocessing='resize_pad_square', input_size=256, conf_thres=0.005),
               dict(preprocessing='resize_pad_divisible', input_size=224, conf_thres=0.05)):
        batched = det.evaluation_predict(eval_info, batch_size=4, **kw)
        loop = []
        for info in infos:
            loop += det.detect_one(img_path=str(tmp_path / info['file_name']), **kw).to_json(img_id=info['id'])
        assert len(batched) == len(loop) > 0
        assert [r['image_id'] for r in batched] == [r['image_id'] for r in loop]
        assert [r['category_id'] for r in batched] == [r['category_id'] for r in loop]
        # numbers to 1e-5, not bit for bit: a batch of 2 and a batch of 1 cut their small grids along K differently
        np.testing.assert_allclose(np.array([r['bbox'] for r in batched]), np.array([r['bbox'] for r in loop]), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(np.array([r['score'] for r in batched]), np.array([r['score'] for r in loop]), rtol=1e-4, atol=2e-5)
        assert all(isinstance(r['bbox'][0], float) and isinstance(r['score'], float) for r in batched)

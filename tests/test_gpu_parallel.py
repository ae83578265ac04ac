"""GPU: the multi-GPU exchange on real RCCL with a one-rank group (a 1-GPU box cannot hold more ranks), and
bench.py end to end in that mode with --verify.  The N > 1 logic (sharding, padding, ordering) is covered on the
CPU by tests/test_parallel_gloo.py."""
import contextlib
import io
import json
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_one_rank_gather_equals_local():
    import torch.distributed as dist
    from mydetection_amd import parallel, synth
    from mydetection_amd.models.general import name_to_model
    from mydetection_amd.utils.structures import batched_post_process
    assert torch.cuda.is_available()
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1, device_id=dev)
    try:
        m, cfg = name_to_model('yolov3_80')
        m.load_state_dict(synth.make_state_dict(m.state_dict()), strict=True)
        m = m.eval().cuda()
        x = synth.make_image_set(0, 3, 256).cuda()
        with torch.no_grad():
            rec = batched_post_process(*m.forward_candidates(x), 0.005, 0.45)
        assert 'records' in rec and rec['records'].shape == (3, parallel.WORDS)       # the kernel wrote the wire format
        assert int(rec['count'].sum()) > 0 and (rec['records'][:, 1:4] == 0).all()
        local = {k: v.clone() for k, v in rec.items()}
        out = parallel.gather_detections(rec, always=True, total=3)                    # RCCL all_gather_into_tensor
        torch.cuda.synchronize()
        assert out['records'].data_ptr() != rec['records'].data_ptr()
        for k in local:
            assert torch.equal(out[k], local[k]), k
        with pytest.raises(ValueError):
            parallel.gather_detections(rec, always=True, total=5)
    finally:
        dist.destroy_process_group()


def test_bench_rehearsal_with_verify(monkeypatch, capsys):
    """bench.py as rank 0 of a one-rank RCCL job: one JSON line, records verified against the 1-GPU pass."""
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '1')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    monkeypatch.setenv('MYDET_REHEARSE_RCCL', '1')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '3',
                                      '--size', '256', '--verify', '--no-cpu-baseline'])
    sys.path.insert(0, ROOT)
    import bench
    bench.main()
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['verify'] == {'ok': True, 'images': 3, 'against': '1-GPU pass over the same global batch (rank 0)'}
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['roofline']['frac'] <= 1.0
    assert {'metric', 'unit', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'dtype', 'data', 'config',
            'roofline', 'stages'} <= set(out)

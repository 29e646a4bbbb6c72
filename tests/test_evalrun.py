"""Eval harness / CLI: folder layout reader, windows, is_first, metrics, score files."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def dataset(tmp_path_factory):
    import make_synth_dataset
    root = str(tmp_path_factory.mktemp('ds'))
    make_synth_dataset.make(root, clips=2, frames=3, h=32, w=48)
    return root


@pytest.fixture(scope='module')
def dataset_long(tmp_path_factory):
    import make_synth_dataset
    root = str(tmp_path_factory.mktemp('ds_long'))
    make_synth_dataset.make(root, clips=2, frames=7, h=32, w=48)
    return root


def _cfg(root, out, extra=()):
    from refvsr_amd import evalrun
    return evalrun.build_config(['--config', 'config_RefVSR_small_L1', '--mode', 'unit', '--data_offset', root,
                                 '--output_offset', out, '--frame_num', '3'] + list(extra))


def test_clipset_layout_windows_and_is_first(dataset, tmp_path):
    from refvsr_amd import evalrun
    cfg = _cfg(dataset, str(tmp_path))
    assert cfg.EVAL.LR_data_path.endswith(os.path.join('RealMCVSR', 'test', 'LRx4'))
    ds = evalrun.ClipSet(cfg)
    assert len(ds) == 6
    firsts = [ds[i]['is_first'] for i in range(6)]
    assert firsts == [True, False, False, True, False, False]
    it = ds[0]
    assert it['LR_UW'].shape == (3, 3, 32, 48) and it['HR_UW'].shape == (3, 128, 192)
    assert torch.equal(it['LR_UW'][0], it['LR_UW'][1])            # edge frame repeated (datasets.py:233-234)
    assert it['video_name'] == '0001' and ds[3]['video_name'] == '0002' and it['frame_name'] == '0000.png'
    assert float(it['LR_UW'].max()) <= 1.0 and it['LR_UW'].dtype == torch.float32
    cfg2 = _cfg(dataset, str(tmp_path), ['--vid_name', '0002'])
    ds2 = evalrun.ClipSet(cfg2)
    assert ds2[0].get('is_continue') and not ds2[3].get('is_continue')


def test_metrics():
    from refvsr_amd import evalrun
    g = torch.Generator().manual_seed(0)
    a = torch.rand(3, 40, 52, generator=g)
    b = (a + 0.05 * torch.randn(3, 40, 52, generator=g)).clamp(0, 1)
    assert evalrun.ssim(a, a) == pytest.approx(1.0)
    s = evalrun.ssim(a, b)
    assert 0.5 < s < 1.0
    # direct numpy restatement of the skimage definition on one channel window
    x, y = a[0, :7, :7].double().numpy(), b[0, :7, :7].double().numpy()
    ux, uy = x.mean(), y.mean()
    vx, vy, vxy = x.var(ddof=1), y.var(ddof=1), ((x - ux) * (y - uy)).sum() / 48.0
    want = ((2 * ux * uy + 1e-4) * (2 * vxy + 9e-4)) / ((ux ** 2 + uy ** 2 + 1e-4) * (vx + vy + 9e-4))
    a1, b1 = a[:1, :7, :7].repeat(3, 1, 1), b[:1, :7, :7].repeat(3, 1, 1)
    assert evalrun.ssim(a1, b1) == pytest.approx(want, rel=1e-9)
    assert evalrun.psnr(a, a * 0 + a.mean()) == pytest.approx(10 * np.log10(1.0 / float(((a - a.mean()) ** 2).mean())), rel=1e-6)


def test_cli_refuses_cpu(dataset, tmp_path):
    from refvsr_amd import evalrun
    cfg = _cfg(dataset, str(tmp_path), ['--cpu'])
    with pytest.raises(RuntimeError, match='no CPU path'):
        evalrun.evaluate(cfg)


@pytest.mark.gpu
def test_cli_end_to_end_on_gpu(dataset, tmp_path):
    """Checkpoint file with a DataParallel `module.` prefix -> CLI -> PNG outputs + score file."""
    from refvsr_amd import evalrun, get_config, make_state_dict
    from oracle import refvsr_oracle as orc
    cfg0 = get_config('p', 'm', 'config_RefVSR_small_L1')
    sd = make_state_dict(cfg0, 1234)
    ck = str(tmp_path / 'RefVSR_small_L1.pytorch')
    torch.save({'module.' + k: v for k, v in sd.items()}, ck)
    cfg = _cfg(dataset, str(tmp_path / 'out'), ['--ckpt_abs_name', ck])
    res = evalrun.evaluate(cfg, log=lambda *_: None)
    assert res['frames'] == 6 and all(np.isfinite(res['psnr']))
    lines = open(res['score_file']).read().splitlines()
    assert lines[0].startswith('[EVAL unit|RealMCVSR|0001][1/2][1/3] 0000.png PSNR: ')
    assert any(ln.startswith('[MEAN EVAL unit|RealMCVSR|0002]') for ln in lines) and lines[-1].startswith('[TOTAL ')
    png = os.path.join(res['output_root'], 'png', 'output', '0001', '0000.png')
    assert os.path.exists(png) and os.path.exists(png.replace('png', 'jpg'))
    # PSNR of the first frame against the oracle's PSNR on the same PNG inputs
    ds = evalrun.ClipSet(cfg)
    o = orc.OracleNetwork(cfg, sd)
    it = ds[0]
    want = o.forward(it['LR_UW'][None], it['LR_REF_W'][None], True)['result'][0]
    assert abs(evalrun.psnr(want, it['HR_UW']) - res['psnr'][0]) < 1e-3
    back = evalrun.read_frame(png)
    assert float((back - want).abs().max()) < 2.0 / 255 + 5e-3          # 8-bit truncation + fp16 path


@pytest.mark.gpu
def test_cli_result_dtype_uint8_writes_the_same_png_bytes(dataset, tmp_path):
    """`--result_dtype uint8` (round 6): the output head stores rint(255 x) on the GPU, a quarter of the bytes crosses PCIe -- the PNG
    files are byte-identical to the float32 run's (whose quantisation happens on the CPU, like eval_qual_quan.py:117-119)."""
    from refvsr_amd import evalrun, get_config, make_state_dict
    sd = make_state_dict(get_config('p', 'm', 'config_RefVSR_small_L1'), 1234)
    ck = str(tmp_path / 'RefVSR_small_L1.pytorch')
    torch.save(sd, ck)
    res = {}
    for dt in ('float32', 'uint8'):
        cfg = _cfg(dataset, str(tmp_path / ('out_' + dt)), ['--ckpt_abs_name', ck, '--result_dtype', dt])
        res[dt] = evalrun.evaluate(cfg, log=lambda *_: None)
    assert res['float32']['frames'] == res['uint8']['frames'] == 6
    for clip in ('0001', '0002'):
        for fr in ('0000', '0001', '0002'):
            a = open(os.path.join(res['float32']['output_root'], 'png', 'output', clip, fr + '.png'), 'rb').read()
            b = open(os.path.join(res['uint8']['output_root'], 'png', 'output', clip, fr + '.png'), 'rb').read()
            assert a == b, 'PNG %s/%s differs' % (clip, fr)
    assert max(abs(p - q) for p, q in zip(res['float32']['psnr'], res['uint8']['psnr'])) < 0.05     # (scores of the 8-bit frame)


@pytest.mark.gpu
def test_cli_frame_groups_equal_the_reference_loop(dataset_long, tmp_path):
    """`--frame_group 4`: the CLI hands the network four consecutive windows of a clip per call (SRNet.forward_group) -- PSNR / SSIM
    of every frame, the score-file structure and the written PNGs equal the one-frame-per-call loop's."""
    from refvsr_amd import evalrun, get_config, make_state_dict
    sd = make_state_dict(get_config('p', 'm', 'config_RefVSR_small_L1'), 1234)
    ck = str(tmp_path / 'RefVSR_small_L1.pytorch')
    torch.save({'module.' + k: v for k, v in sd.items()}, ck)
    res = {}
    for g in (1, 4):
        cfg = _cfg(dataset_long, str(tmp_path / ('out%d' % g)), ['--ckpt_abs_name', ck, '--frame_group', str(g)])
        res[g] = evalrun.evaluate(cfg, log=lambda *_: None)
    assert res[1]['frames'] == res[4]['frames'] == 14
    assert res[1]['psnr'] == res[4]['psnr'] and res[1]['ssim'] == res[4]['ssim']
    l1, l4 = (open(res[g]['score_file']).read().splitlines() for g in (1, 4))
    strip = lambda ln: ln.split(' (')[0]                      # everything but the per-frame seconds
    assert [strip(a) for a in l1] == [strip(b) for b in l4]
    for clip, frame in (('0001', '0000'), ('0001', '0005'), ('0002', '0006')):
        a = evalrun.read_frame(os.path.join(res[1]['output_root'], 'png', 'output', clip, frame + '.png'))
        b = evalrun.read_frame(os.path.join(res[4]['output_root'], 'png', 'output', clip, frame + '.png'))
        assert torch.equal(a, b)

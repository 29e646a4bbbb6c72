"""Evaluation harness + `run.py`-compatible CLI for the HIP path.

Counterpart of the reference's eval stack, restated from its behaviour:
  * RealMCVSR folder layout and file listing     -- configs/config.py:120-152, data_loader/utils.py:247-287
  * per-frame sliding windows, edge-frame repeats -- data_loader/datasets.py:222-234
  * `is_first` per clip                            -- datasets.py:286-288 (the reference yields False for frame 0
    of a single-clip dataset and then crashes, RefVSR.py:257-258; here the first frame of every clip is True)
  * PSNR = 10 log10(1/mse)                         -- trainers/trainer.py:252-254
  * SSIM = skimage.structural_similarity defaults  -- evaluation/metrics.py:17-18 (7x7 uniform window, sample
    covariance, K1=0.01, K2=0.03, mean over channels) re-implemented (skimage is not in this image)
  * score file lines / output tree                 -- evaluation/eval_qual_quan.py:98-101,106-124,140-143
  * checkpoint loading (flat state dict, optional `module.` prefix) -- ckpt_manager.py:50-60

    python -m refvsr_amd.evalrun --mode amp_RefVSR_small_L1 --config config_RefVSR_small_L1 --data RealMCVSR \
        --ckpt_abs_name ckpt/RefVSR_small_L1.pytorch --data_offset /data --output_offset ./result --frame_num 5
"""
import argparse
import datetime
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

from .config import get_config, set_data_path
from .synth import window_indices


# ------------------------------------------------------------------------------------------------ data
def load_file_list(root_path):
    """Leaf folders under root_path (sorted) and their sorted files (data_loader/utils.py:247-287)."""
    folders, files = [], []
    for root, dirnames, filenames in os.walk(root_path):
        dirnames[:] = [d for d in dirnames if not d.startswith('@')]
        if not dirnames:
            fs = sorted(os.path.join(root, f) for f in filenames if not f.startswith('.') and f != 'Thumbs.db')
            folders.append(root)
            files.append(fs)
    order = np.argsort(np.array(folders)) if folders else []
    return [folders[i] for i in order], [files[i] for i in order]


def read_frame(path):
    """PIL image -> float32 [3,H,W] in [0,1] (data_loader/utils.py:12-41)."""
    from PIL import Image
    a = np.asarray(Image.open(path).convert('RGB'), dtype=np.float32) / 255.0
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


def write_frame(path, img, quality=None):
    from PIL import Image
    if img.dtype == torch.uint8:
        # config.result_dtype = 'uint8': the output head stored rint(255 x) itself (REFVSR_RESULT_U8) -- the bytes computed below
        im = Image.fromarray(img.detach().cpu().numpy().transpose(1, 2, 0))
    else:
        a = (img.detach().float().cpu().clamp(0, 1).numpy().transpose(1, 2, 0) * 255.0)
        # cv2.imwrite converts a float image with convertTo(CV_8U) = saturate_cast<uchar>: round to nearest, saturate
        # (eval_qual_quan.py:117-119 hands it output*255 as float)
        im = Image.fromarray(np.rint(a).clip(0, 255).astype(np.uint8))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    im.save(path, **({'quality': quality} if quality else {}))


class ClipSet(object):
    """One item per output frame of every clip, in the reference's order (datasets.py:150-316)."""

    def __init__(self, config):
        self.config = config
        E = config.EVAL
        _, self.lr_uw = load_file_list(os.path.join(E.LR_data_path, config.UW_path))
        _, self.lr_w = load_file_list(os.path.join(E.LR_data_path, config.W_path))
        _, self.hr_uw = load_file_list(os.path.join(E.HR_data_path, config.UW_path))
        assert len(self.lr_uw) == len(self.lr_w) == len(self.hr_uw) and self.lr_uw, \
            'no clips found under %s' % E.LR_data_path
        self.items = [(v, f) for v in range(len(self.lr_uw)) for f in range(len(self.lr_uw[v]))]
        self._cache = {}

    def __len__(self):
        return len(self.items)

    def _frame(self, path):
        if path not in self._cache:
            if len(self._cache) > 64:
                self._cache.clear()
            self._cache[path] = read_frame(path)
        return self._cache[path]

    def __getitem__(self, index):
        v, f = self.items[index]
        t = self.config.frame_num
        n = len(self.lr_uw[v])
        win = window_indices(f, n, t)
        name = os.path.basename(os.path.dirname(self.lr_uw[v][f]))
        vid_filter = getattr(self.config.EVAL, 'vid_name', None)
        if vid_filter is not None and name not in vid_filter:
            return {'is_continue': True, 'is_first': True, 'video_name': name, 'frame_len': n}
        return {
            'LR_UW': torch.stack([self._frame(self.lr_uw[v][i]) for i in win]),
            'LR_REF_W': torch.stack([self._frame(self.lr_w[v][i]) for i in win]),
            'HR_UW': self._frame(self.hr_uw[v][f]),
            'is_first': f == 0, 'video_name': name, 'video_idx': v, 'video_len': len(self.lr_uw),
            'frame_idx': f, 'frame_len': n, 'frame_name': os.path.basename(self.lr_uw[v][f]),
            'frame_ids': [(v, int(i)) for i in win],     # names the window's frames for the cross-window cache
        }


# ------------------------------------------------------------------------------------------------ metrics
def psnr(a, b):
    return float(10.0 * torch.log10(1.0 / torch.mean((a.float() - b.float()) ** 2)))


def ssim(a, b, data_range=1.0, win=7):
    """skimage.metrics.structural_similarity(a, b, data_range=1.0, multichannel=True) for [3,H,W] tensors."""
    a, b = a.double()[None], b.double()[None]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    norm = win * win / (win * win - 1.0)
    f = lambda x: F.avg_pool2d(x, win, stride=1)
    ua, ub = f(a), f(b)
    va, vb, vab = norm * (f(a * a) - ua * ua), norm * (f(b * b) - ub * ub), norm * (f(a * b) - ua * ub)
    s = ((2 * ua * ub + c1) * (2 * vab + c2)) / ((ua * ua + ub * ub + c1) * (va + vb + c2))
    return float(s.mean())


# ------------------------------------------------------------------------------------------------ eval loop
def load_checkpoint(net, path):
    sd = torch.load(path, map_location='cpu')
    if isinstance(sd, dict) and 'state_dict' in sd:
        sd = sd['state_dict']
    return net.load_state_dict(sd, strict=False)


def evaluate(config, net=None, log=print):
    """eval_qual_quan counterpart.  Returns dict(psnr=[..], ssim=[..], frames=N, seconds=[..])."""
    from . import SRNet
    E = config.EVAL
    if net is None:
        if config.device != 'cuda':
            raise RuntimeError('refvsr_amd has no CPU path; run on the GPU (drop --cpu)')
        net = SRNet(config).to('cuda').eval()
        if E.ckpt_abs_name:
            log('Loading checkpoint %s: %s' % (E.ckpt_abs_name, load_checkpoint(net, E.ckpt_abs_name)))
    ckpt_name = os.path.basename(E.ckpt_abs_name) if E.ckpt_abs_name else 'seeded'
    date = datetime.datetime.now().strftime('%Y_%m_%d_%H%M')
    root = os.path.join(E.LOG_DIR.save, E.eval_mode, ckpt_name.split('.')[0])
    out_root = os.path.join(root, E.data, date)
    os.makedirs(root, exist_ok=True)
    score_path = os.path.join(root, 'score_%s_%s.txt' % (E.data, E.eval_mode))
    ds = ClipSet(config)
    dev = next(net.parameters()).device
    res = {'psnr': [], 'ssim': [], 'seconds': [], 'frames': 0}
    # --frame_group G (extension, default 1 = the reference's loop): G consecutive windows of a clip per network call
    # (SRNet.forward_group: the backward branches of the G frames as multi-map launches; results bit-identical); the per-frame time in
    # the score lines is then the group's time / G
    G = max(1, int(getattr(E, 'frame_group', 1) or 1))
    was_pipelined = bool(getattr(config, 'pipelined', False))
    if G > 1:
        net.Network.set_pipelined(True)          # (restored at the end: the caller's plain net(...) calls keep their stream contract)
    st = {'clip_p': 0.0, 'clip_s': 0.0, 'clip_t': 0.0, 'clip_n': 0, 'first_line': True, 'prev': None}

    def emit(it, out, lr_c, dt):
        out_raw = out[0].cpu()
        # (result_dtype 'uint8' / 'float16', extensions: the scores are then those of the quantised frame; the PNG bytes are the same)
        out_cpu = out_raw.float() / 255.0 if out_raw.dtype == torch.uint8 else out_raw.float()
        p = s = 0.0
        if not getattr(E, 'qualitative_only', False):
            gt = it['HR_UW']
            p = psnr(out_cpu, gt)
            cmp_out = out_cpu
            if config.flag_HD_in:          # eval_qual_quan.py:86-87: SSIM against the LR-size GT
                cmp_out = F.interpolate(out_cpu[None], scale_factor=1.0 / config.scale, mode='bicubic',
                                        align_corners=False)[0]
            if cmp_out.shape == gt.shape:
                s = ssim(cmp_out, gt)
        line = '[EVAL {}|{}|{}][{}/{}][{}/{}] {} PSNR: {:.5f} SSIM: {:.5f} ({:.5f}sec)'.format(
            config.mode, E.data, it['video_name'], it['video_idx'] + 1, it['video_len'], it['frame_idx'] + 1,
            it['frame_len'], it['frame_name'], p, s, dt)
        log(line)
        with open(score_path, 'w' if st['first_line'] else 'a') as fh:
            fh.write(line + '\n')
        st['first_line'] = False
        if not getattr(E, 'quantitative_only', False):
            stem = it['frame_name'].split('.')[0]
            for fmt in ('png', 'jpg'):
                base = os.path.join(out_root, fmt)
                write_frame(os.path.join(base, 'input', it['video_name'], '%s.%s' % (stem, fmt)), lr_c)
                write_frame(os.path.join(base, 'output', it['video_name'], '%s.%s' % (stem, fmt)), out_raw if out_raw.dtype == torch.uint8 else out_cpu)
        st['clip_p'] += p
        st['clip_s'] += s
        st['clip_t'] += dt
        st['clip_n'] += 1
        st['prev'] = it
        res['psnr'].append(p)
        res['ssim'].append(s)
        res['seconds'].append(dt)
        res['frames'] += 1

    def flush(pending):
        if not pending:
            return
        t0 = time.time()
        lrs = torch.stack([it['LR_UW'] for it in pending], 0).to(dev).float().contiguous()
        rfs = torch.stack([it['LR_REF_W'] for it in pending], 0).to(dev).float().contiguous()
        outs = net.forward_group(lrs, rfs, [it['frame_ids'] for it in pending])['result']
        torch.cuda.synchronize()
        dt = (time.time() - t0) / len(pending)
        c = lrs.shape[1] // 2
        for b, it in enumerate(pending):
            emit(it, outs[b], lrs[b, c], dt)
        del pending[:]

    try:
        _evaluate_loop(net, ds, dev, E, G, st, emit, flush, config, score_path, log)
    finally:
        if G > 1 and not was_pipelined:
            torch.cuda.synchronize()
            net.Network.set_pipelined(False)
    clip_p, clip_s, clip_t, clip_n, prev = st['clip_p'], st['clip_s'], st['clip_t'], st['clip_n'], st['prev']
    if clip_n:
        _clip_summary(config, score_path, prev, clip_p, clip_s, clip_t, clip_n, log)
    n = max(res['frames'], 1)
    total = '\n[TOTAL {}|{}] PSNR: {:.5f} SSIM: {:.5f} ({:.5f}sec)'.format(
        ckpt_name, E.data, sum(res['psnr']) / n, sum(res['ssim']) / n, sum(res['seconds']) / n)
    log(total)
    with open(score_path, 'a') as fh:
        fh.write(total + '\n')
    res['score_file'], res['output_root'] = score_path, out_root
    return res


def _evaluate_loop(net, ds, dev, E, G, st, emit, flush, config, score_path, log):
    """The per-frame loop of evaluate() (eval_qual_quan.py:39-128)."""
    with torch.no_grad():
        pending = []
        for i in range(len(ds)):
            it = ds[i]
            if it.get('is_continue'):
                continue
            if it['is_first']:
                flush(pending)
                if st['clip_n']:
                    _clip_summary(config, score_path, st['prev'], st['clip_p'], st['clip_s'], st['clip_t'], st['clip_n'], log)
                    st['clip_p'] = st['clip_s'] = st['clip_t'] = 0.0
                    st['clip_n'] = 0
                net.Network.reset()
            use_ids = getattr(E, 'use_frame_ids', True) and 'frame_ids' in it
            if G > 1 and use_ids and not it['is_first']:
                pending.append(it)
                if len(pending) == G:
                    flush(pending)
                continue
            t0 = time.time()
            lr, rf = it['LR_UW'][None].to(dev), it['LR_REF_W'][None].to(dev)
            kw = {'frame_ids': it['frame_ids']} if use_ids else {}
            out = net(lr, rf, it['is_first'], is_log=False, is_train=False, **kw)['result']
            torch.cuda.synchronize()
            emit(it, out, lr[0, lr.shape[1] // 2], time.time() - t0)
        flush(pending)


def _clip_summary(config, score_path, it, p, s, t, n, log):
    line = '[MEAN EVAL {}|{}|{}][{}/{}] PSNR: {:.5f} SSIM: {:.5f} ({:.5f}sec)\n'.format(
        config.mode, config.EVAL.data, it['video_name'], it['video_idx'], it['video_len'], p / n, s / n, t / n)
    log(line)
    with open(score_path, 'a') as fh:
        fh.write(line + '\n')


# ------------------------------------------------------------------------------------------------ CLI
def build_config(argv=None):
    """run.py:218-417 flag surface (evaluation subset)."""
    ap = argparse.ArgumentParser(description='RefVSR evaluation on the MI355X HIP path')
    ap.add_argument('-proj', '--project', type=str, default='RefVSR_CVPR2022')
    ap.add_argument('-m', '--mode', type=str, default='eval')
    ap.add_argument('-c', '--config', type=str, default='config_RefVSR_small_L1')
    ap.add_argument('-data', '--data', type=str, default='RealMCVSR')
    ap.add_argument('-net', '--network', type=str, default=None)
    ap.add_argument('-data_offset', '--data_offset', type=str, default=None)
    ap.add_argument('-output_offset', '--output_offset', type=str, default='./result')
    ap.add_argument('-ckpt_abs_name', '--ckpt_abs_name', type=str, default=None)
    ap.add_argument('-cpu', '--cpu', action='store_true')
    ap.add_argument('-eval_mode', '--eval_mode', type=str, default='qual_quan')
    ap.add_argument('-test_set', '--test_set', type=str, default='test')
    ap.add_argument('-qualitative_only', '--qualitative_only', action='store_true')
    ap.add_argument('-quantitative_only', '--quantitative_only', action='store_true')
    ap.add_argument('-is_gradio', '--is_gradio', action='store_true')
    ap.add_argument('-frame_num', '--frame_num', type=int, default=None)
    ap.add_argument('-vid_name', '--vid_name', nargs='+', default=None)
    ap.add_argument('-ss', '--save_sample', action='store_true')
    ap.add_argument('--frame_group', type=int, default=1, help='extension: consecutive frames of a clip per network call (1 = the reference loop; 4 = '
                                                                'multi-map launches of the backward branches, same results)')
    ap.add_argument('--result_dtype', default='float32', choices=['float32', 'float16', 'uint8'],
                    help="extension: what the output head stores ('uint8' = rint(255 x), the bytes of the written PNG; the scores are then "
                         "those of the quantised frame)")
    args, _ = ap.parse_known_args(argv)
    cfg = get_config(args.project, args.mode, args.config, args.data)
    cfg.result_dtype = args.result_dtype
    if args.network:
        cfg.network = args.network
    if args.frame_num:
        cfg.frame_num = args.frame_num
    cfg.center_idx = cfg.frame_num // 2
    E = cfg.EVAL
    E.ckpt_abs_name = args.ckpt_abs_name
    E.qualitative_only, E.quantitative_only = args.qualitative_only, args.quantitative_only
    E.is_gradio, E.vid_name = args.is_gradio, args.vid_name
    E.eval_mode, E.test_set, E.data = args.eval_mode, args.test_set, args.data
    E.frame_group = args.frame_group
    cfg.save_sample = args.save_sample
    cfg.device = 'cpu' if args.cpu else 'cuda'
    cfg.cuda = not args.cpu
    if args.data_offset:
        cfg.data_offset = args.data_offset
    E.LOG_DIR = {'save': args.output_offset}
    return set_data_path(cfg, E.data, is_train=False)


def main(argv=None):
    cfg = build_config(argv)
    res = evaluate(cfg)
    return 0 if res['frames'] else 1


if __name__ == '__main__':
    sys.exit(main())

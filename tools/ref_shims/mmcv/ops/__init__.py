"""mmcv.ops stand-in for the ONE op the RefVSR_IR reference imports: the modulated deformable convolution (DCNv2) of
mmcv-full, which this image does not have (compiled CUDA / C++ extension, no network).  Pure-PyTorch restatement of the
published op (mmcv/ops/csrc/common/cuda/modulated_deform_conv_cuda_kernel.cuh, dmcn_im2col_bilinear): for output
pixel (y, x), deformable group g, tap k = ky*kw + kx the input is sampled bilinearly at
    (y*stride - pad + ky*dil + offset[g*2*K + 2k], x*stride - pad + kx*dil + offset[g*2*K + 2k + 1])
(samples at h <= -1, w <= -1, h >= H, w >= W contribute 0; corner pixels outside the map contribute 0), multiplied by
mask[g*K + k], and the columns are contracted with the weight [Cout, Cin/groups, kh, kw] (+ bias).
Used only by tools/gen_golden.py in the build container: fixtures generated through it pin the build's DCN against THIS
restatement of mmcv's op, not against mmcv's binary (stated as such in DESIGN.md)."""
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair


def _bilinear_zero(x, py, px):
    """x [n, c, H, W]; py, px [n, 1, Ho, Wo] float sample coordinates -> [n, c, Ho, Wo] (mmcv dmcn_im2col_bilinear)."""
    n, c, H, W = x.shape
    valid = (py > -1) & (px > -1) & (py < H) & (px < W)
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    ly, lx = py - y0, px - x0
    hy, hx = 1 - ly, 1 - lx
    y0, x0 = y0.long(), x0.long()
    out = 0
    for dy, dx, wgt in ((0, 0, hy * hx), (0, 1, hy * lx), (1, 0, ly * hx), (1, 1, ly * lx)):
        yy, xx = y0 + dy, x0 + dx
        ok = valid & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).expand(n, c, -1, -1).reshape(n, c, -1)
        v = torch.gather(x.reshape(n, c, H * W), 2, idx).view(n, c, *py.shape[-2:])
        out = out + v * (wgt * ok.to(x.dtype))
    return out


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
    assert groups == 1
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    n, cin, H, W = x.shape
    cout, _, kh, kw = weight.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    K = kh * kw
    cg = cin // deform_groups
    ys = (torch.arange(Ho, dtype=x.dtype) * sh - ph).view(1, 1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * sw - pw).view(1, 1, 1, Wo)
    cols = x.new_zeros(n, cin, K, Ho, Wo)
    for g in range(deform_groups):
        xg = x[:, g * cg:(g + 1) * cg]
        for k in range(K):
            ky, kx = divmod(k, kw)
            py = ys + ky * dh + offset[:, g * 2 * K + 2 * k:g * 2 * K + 2 * k + 1]
            px = xs + kx * dw + offset[:, g * 2 * K + 2 * k + 1:g * 2 * K + 2 * k + 2]
            cols[:, g * cg:(g + 1) * cg, k] = _bilinear_zero(xg, py, px) * mask[:, g * K + k:g * K + k + 1]
    out = torch.einsum('nckhw,ock->nohw', cols, weight.reshape(cout, cin, K))
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


class ModulatedDeformConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x, offset, mask):
        return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                       self.groups, self.deform_groups)

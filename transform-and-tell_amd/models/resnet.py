"""ResNet-152 trunk (tell/models/resnet.py:12-192) on the MI355X path.

NHWC activations: 1x1/s1 convolutions are NT GEMMs straight on the activation matrix
[B*H*W, Cin]; 3x3, 7x7 and strided 1x1 go through im2col rows; BatchNorm uses batch
statistics when `self.training` (the reference trains with the frozen trunk in train mode,
callback_apex_trainer.py:259) and running statistics in eval.  Parameter / buffer names are
torchvision's so that the reference's `resnet.*` checkpoint entries load.
Returns the region features as [B, 49, 2048] (== permute(0,2,3,1).view(B,49,2048),
transformer_faces_objects.py:335-341)."""
import os

import torch
import torch.nn as nn

from .. import hip, ops
from .. import runtime as rt

call = hip.call


def _conv_param(cout, cin, k):
    w = torch.empty(cout, cin, k, k)
    nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')        # resnet.py:50-53
    return nn.Parameter(w)


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = 1e-5, 0.1


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.weight = _conv_param(cout, cin, k)
        self.k, self.stride, self.padding, self.cin, self.cout = k, stride, padding, cin, cout

    def gemm_weight(self):
        """[Cout, KH*KW*Cin padded] working copy (compute dtype), cached per weights epoch."""
        def make():
            w = self.weight.detach().permute(0, 2, 3, 1).reshape(self.cout, -1)     # [Cout, KH, KW, Cin]
            K = w.shape[1]
            Kp = ops._round_up(K, 8)
            wp = torch.zeros(self.cout, Kp, dtype=torch.float32, device=w.device)
            wp[:, :K] = w
            return ops.cast(wp, rt.compute_dtype())
        return ops._cached(self.weight, ('convw',), make)

    def stem_weight(self, bn=None):
        """The 7x7 / stride 2 / padding 3 stem (resnet.py:92-96) for the implicit gather over NHWC4 pixels
        (csrc/gemm.hip, tell_nchw_to_nhwc4): [Cout, 256] = 8 kernel rows (the 8th zero) x 8 window columns (the first
        zero - the window of output column ow starts at input column 2 ow - 4) x 4 channels (the 4th zero).
        bn given: eval mode, BatchNorm folded in -> (weight, fp32 bias)."""
        def lay(w):                                   # w: [Cout, 3, 7, 7] fp32
            wk = torch.zeros(self.cout, 8, 8, 4, dtype=torch.float32, device=w.device)
            wk[:, :7, 1:, :3] = w.permute(0, 2, 3, 1)
            return ops.cast(wk.reshape(self.cout, 256), rt.compute_dtype())
        if bn is None:
            return ops._cached(self.weight, ('stemw',), lambda: lay(self.weight.detach().float()))

        def make():
            scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
            bias = (bn.bias.detach().float() - bn.running_mean.float() * scale).contiguous()
            return lay(self.weight.detach().float() * scale[:, None, None, None]), bias
        key = ('stemfold', _STATS_EPOCH[0], bn.weight._version, bn.bias._version, bn.running_mean._version,
               bn.running_var._version, bn.weight.data_ptr(), bn.running_var.data_ptr())
        return ops._cached(self.weight, key, make)

    def folded(self, bn):
        """Eval mode (running statistics, resnet.py:92-108 under model.eval()): the BatchNorm behind this convolution
        folded into it - ([Cout, KH*KW*Cin padded] working weight w * gamma / sqrt(var + eps), fp32 bias beta - mean *
        that) - cached per state of the weights AND of the running statistics (_STATS_EPOCH: train-mode passes update
        those buffers from inside kernels, behind torch's version counters)."""
        def make():
            scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
            w = self.weight.detach().float().permute(0, 2, 3, 1).reshape(self.cout, -1) * scale[:, None]
            K = w.shape[1]
            Kp = ops._round_up(K, 8)
            wp = torch.zeros(self.cout, Kp, dtype=torch.float32, device=w.device)
            wp[:, :K] = w
            bias = (bn.bias.detach().float() - bn.running_mean.float() * scale).contiguous()
            return ops.cast(wp, rt.compute_dtype()), bias
        key = ('fold', _STATS_EPOCH[0], bn.weight._version, bn.bias._version, bn.running_mean._version,
               bn.running_var._version, bn.weight.data_ptr(), bn.running_var.data_ptr())
        return ops._cached(self.weight, key, make)


_BN_WS = {}
_STATS_EPOCH = [0]       # bumped by every train-mode pass of a trunk: the running statistics changed (see _Conv.folded)


def stats_epoch():
    return _STATS_EPOCH[0]


def _stat_buffers(device, need_ws, C):
    """(workspace, [two (mean, invstd) slots]) - stream-ordered reuse across layers; two slots because a
    BatchNorm whose apply is deferred into the next conv's im2col is still live while that conv's own
    statistics are produced."""
    buf = _BN_WS.get(device)
    if buf is None or buf[0].numel() < need_ws or buf[1].numel() < 4 * C:
        buf = (torch.empty(max(need_ws, 1 << 20), dtype=torch.float32, device=device),
               torch.empty(4 * max(C, 4096), dtype=torch.float32, device=device))
        _BN_WS[device] = buf
    cap = buf[1].numel() // 4
    return buf[0], [(buf[1][2 * s * cap:2 * s * cap + C], buf[1][(2 * s + 1) * cap:(2 * s + 1) * cap + C])
                    for s in range(2)]


def conv_stats(x, B, H, W, conv, bn, training, pre=None, slot=0):
    """The convolution + the statistics of its BatchNorm, NOT yet applied.
    x: [B*H*W, Cin] NHWC rows.  pre: (mean, invstd, bn_module) of a producer BatchNorm(+ReLU) to apply while
    gathering (3x3 convs only).  -> (raw y [B*OH*OW, Cout], OH, OW, mean, invstd)."""
    k, s, p = conv.k, conv.stride, conv.padding
    OH = (H + 2 * p - k) // s + 1
    OW = (W + 2 * p - k) // s + 1
    wq = conv.gemm_weight()
    dtype = x.dtype
    if k == 1 and s == 1:
        assert pre is None
        a = x
    else:
        a = torch.empty(B * OH * OW, wq.shape[1], dtype=dtype, device=x.device)
        if pre is not None:
            pm, pi, pbn = pre
            call('tell_im2col_bn', x, a, B, H, W, conv.cin, k, k, s, p, OH, OW, pm, pi, pbn.weight.detach(),
                 pbn.bias.detach(), 1, hip.dt(dtype))
        else:
            call('tell_im2col', x, a, B, H, W, conv.cin, k, k, s, p, OH, OW, wq.shape[1], hip.dt(dtype))
    M, C = a.shape[0], wq.shape[0]
    if training:
        fused = dtype == torch.bfloat16 and C % 8 == 0        # statistics come out of the GEMM epilogue
        need = 2 * ((M + 63) // 64) * C if fused else 2 * hip.lib().tell_bn_chunks(M) * C
        ws, slots = _stat_buffers(x.device, need, C)
        mean, invstd = slots[slot]
        if fused:
            y = torch.empty(M, C, dtype=dtype, device=x.device)
            call('tell_gemm_bn_stats', a, a.stride(0), wq, wq.stride(0), y, y.stride(0), M, C, a.shape[1], bn.eps,
                 bn.momentum, mean, invstd, bn.running_mean, bn.running_var, ws)
        else:
            y = ops.gemm(a, wq)
            call('tell_bn_stats', y, M, C, bn.eps, bn.momentum, mean, invstd, bn.running_mean, bn.running_var, ws,
                 hip.dt(dtype))
    else:
        y = ops.gemm(a, wq)
        mean = bn.running_mean
        invstd = ops._cached(bn.running_var, ('invstd', _STATS_EPOCH[0]), lambda: torch.rsqrt(bn.running_var + bn.eps))
    return y, OH, OW, mean, invstd


def bn_apply(y, mean, invstd, bn, relu, residual=None):
    M, C = y.shape
    call('tell_bn_apply', y, mean, invstd, bn.weight.detach(), bn.bias.detach(), residual, y, M, C, int(relu),
         hip.dt(y.dtype))
    return y


def conv_bn_act(x, B, H, W, conv, bn, relu, residual, training):
    """x: [B*H*W, Cin] NHWC rows -> ([B*OH*OW, Cout], OH, OW) with BN (+residual) (+ReLU) applied."""
    if not training and (residual is None or relu):
        # eval: the BatchNorm is folded into the weights, bias / residual / ReLU ride in the GEMM epilogue
        k, s, p = conv.k, conv.stride, conv.padding
        OH = (H + 2 * p - k) // s + 1
        OW = (W + 2 * p - k) // s + 1
        wq, bias = conv.folded(bn)
        if k == 1 and s == 1:
            a = x
        else:
            a = torch.empty(B * OH * OW, wq.shape[1], dtype=x.dtype, device=x.device)
            call('tell_im2col', x, a, B, H, W, conv.cin, k, k, s, p, OH, OW, wq.shape[1], hip.dt(x.dtype))
        act = 4 if residual is not None else int(bool(relu))
        return ops.gemm(a, wq, bias=bias, bias_mode=1, act=act, aux=residual), OH, OW
    y, OH, OW, mean, invstd = conv_stats(x, B, H, W, conv, bn, training)
    return bn_apply(y, mean, invstd, bn, relu, residual), OH, OW


# ---- bf16 path: implicit-GEMM convolution (no im2col matrix in HBM) with the BatchNorm statistics in its epilogue
_ZERO_PAGE = {}


def _zero_page(device):
    z = _ZERO_PAGE.get(device)
    if z is None:
        z = _ZERO_PAGE[device] = torch.zeros(256, dtype=torch.uint8, device=device)   # padding ring of the gather
    return z


def implicit_ok(conv, dtype):
    c = conv.cin // 64
    return (dtype == torch.bfloat16 and conv.k in (1, 3) and conv.cin % 64 == 0 and c & (c - 1) == 0 and
            conv.cout % 8 == 0 and conv.padding == conv.k // 2)


def conv_bn_implicit(x, B, H, W, conv, bn, relu, residual, training, slot=0):
    """conv -> BatchNorm (-> + residual) (-> ReLU): implicit-GEMM convolution whose epilogue leaves the per-tile
    statistics (+ one small finish launch in train mode), then one elementwise pass.  x: [B*H*W, Cin] NHWC rows
    (post-activation).  -> ([B*OH*OW, Cout], OH, OW)."""
    k, s, p = conv.k, conv.stride, conv.padding
    OH = (H + 2 * p - k) // s + 1
    OW = (W + 2 * p - k) // s + 1
    M, C = B * OH * OW, conv.cout
    wq = conv.gemm_weight()
    y = torch.empty(M, C, dtype=x.dtype, device=x.device)
    if training:
        ws, _ = _stat_buffers(x.device, 2 * ((M + 63) // 64) * C + 2 * C + 256 * C, C)
        call('tell_conv_bn_act', x, wq, y, B, H, W, conv.cin, k, k, s, p, OH, OW, C, bn.eps, bn.momentum,
             bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, residual, int(relu), ws,
             _zero_page(x.device))
        return y, OH, OW
    if residual is None or relu:
        wf, bias = conv.folded(bn)                 # eval: BatchNorm folded into the weights, ONE launch per convolution
        call('tell_conv_bias_act', x, wf, y, B, H, W, conv.cin, k, k, s, p, OH, OW, C, bias, residual, int(bool(relu)),
             _zero_page(x.device))
        return y, OH, OW
    call('tell_conv_bn_stats', x, wq, y, B, H, W, conv.cin, k, k, s, p, OH, OW, C, bn.eps, bn.momentum, None,
         None, None, None, None, _zero_page(x.device))
    mean = bn.running_mean
    invstd = ops._cached(bn.running_var, ('invstd', _STATS_EPOCH[0]), lambda: torch.rsqrt(bn.running_var + bn.eps))
    return bn_apply(y, mean, invstd, bn, relu, residual), OH, OW


_STEM_IMPLICIT = os.environ.get('TELL_STEM_IMPLICIT', '1') != '0'          # A/B aid: 0 = im2col rows + plain GEMM


def stem_bn_act(x4, B, H, W, conv, bn, training):
    """conv1 (7x7, stride 2, padding 3) -> bn1 -> ReLU (resnet.py:94-97) on NHWC4 pixels x4 [B*H*W, 4]: the implicit
    gather of conv_bn_implicit with the stem's own window layout.  -> ([B*OH*OW, Cout], OH, OW)."""
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    M, C = B * OH * OW, conv.cout
    y = torch.empty(M, C, dtype=x4.dtype, device=x4.device)
    if training:
        ws, _ = _stat_buffers(x4.device, 2 * ((M + 63) // 64) * C + 2 * C + 256 * C, C)
        call('tell_conv_bn_act', x4, conv.stem_weight(), y, B, H, W, 4, 7, 7, 2, 3, OH, OW, C, bn.eps, bn.momentum,
             bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, None, 1, ws, _zero_page(x4.device))
    else:
        wf, bias = conv.stem_weight(bn)
        call('tell_conv_bias_act', x4, wf, y, B, H, W, 4, 7, 7, 2, 3, OH, OW, C, bias, None, 1, _zero_page(x4.device))
    return y, OH, OW


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _Conv(inplanes, planes, 1)
        self.bn1 = _BN(planes)
        self.conv2 = _Conv(planes, planes, 3, stride=stride, padding=1)
        self.bn2 = _BN(planes)
        self.conv3 = _Conv(planes, planes * 4, 1)
        self.bn3 = _BN(planes * 4)
        self.downsample = downsample

    def run(self, x, B, H, W, training):
        idt = x
        if all(implicit_ok(c, x.dtype) for c in (self.conv1, self.conv2, self.conv3)) and \
                (self.downsample is None or implicit_ok(self.downsample[0], x.dtype)):
            # bf16: every conv is an implicit GEMM (+ statistics epilogue, + finish in train mode), every BatchNorm
            # (+ReLU, + residual) one elementwise pass; the BatchNorms of conv1 / conv2 run on the block's two SMALL
            # tensors ([rows, planes]) instead of inside a 9x larger im2col matrix
            if self.downsample is not None:
                idt, _, _ = conv_bn_implicit(x, B, H, W, self.downsample[0], self.downsample[1], False, None, training, 0)
            y, _, _ = conv_bn_implicit(x, B, H, W, self.conv1, self.bn1, True, None, training, 1)
            y, OH, OW = conv_bn_implicit(y, B, H, W, self.conv2, self.bn2, True, None, training, 0)
            return conv_bn_implicit(y, B, OH, OW, self.conv3, self.bn3, True, idt, training, 1)
        if self.downsample is not None:
            idt, _, _ = conv_bn_act(x, B, H, W, self.downsample[0], self.downsample[1], False, None, training)
        vec = 8 if x.dtype == torch.bfloat16 else 4
        if self.conv2.cin % vec == 0 and training:
            # bn1 + ReLU are applied inside conv2's im2col gather: conv1's raw output is the only copy in HBM
            y1, _, _, m1, i1 = conv_stats(x, B, H, W, self.conv1, self.bn1, training, slot=0)
            y, OH, OW, m2, i2 = conv_stats(y1, B, H, W, self.conv2, self.bn2, training, pre=(m1, i1, self.bn1), slot=1)
            y = bn_apply(y, m2, i2, self.bn2, True)
        else:
            y, _, _ = conv_bn_act(x, B, H, W, self.conv1, self.bn1, True, None, training)
            y, OH, OW = conv_bn_act(y, B, H, W, self.conv2, self.bn2, True, None, training)
        y, _, _ = conv_bn_act(y, B, OH, OW, self.conv3, self.bn3, True, idt, training)
        return y, OH, OW


class ResNetFeatureExtractor(nn.Module):
    def __init__(self, layers=(3, 8, 36, 3), width=64, num_classes=1000):
        super().__init__()
        self.inplanes = width
        self.conv1 = _Conv(3, width, 7, stride=2, padding=3)
        self.bn1 = _BN(width)
        self.layer1 = self._make(width, layers[0], 1)
        self.layer2 = self._make(width * 2, layers[1], 2)
        self.layer3 = self._make(width * 4, layers[2], 2)
        self.layer4 = self._make(width * 8, layers[3], 2)
        self.fc = nn.Linear(width * 8 * 4, num_classes)          # kept for state_dict compatibility (:48)
        self.out_channels = width * 8 * 4

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(_Conv(self.inplanes, planes * 4, 1, stride=stride), _BN(planes * 4))
        seq = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        seq += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    @torch.no_grad()
    def forward(self, image, pool=False):
        assert not pool
        hip.require_gpu()
        dtype = rt.compute_dtype()
        B, C, H, W = image.shape
        tr = self.training
        if tr:
            _STATS_EPOCH[0] += 1                   # this pass rewrites every running_mean / running_var
        c1 = self.conv1
        if (_STEM_IMPLICIT and dtype == torch.bfloat16 and C <= 4 and W % 2 == 0 and (c1.k, c1.stride, c1.padding) == (7, 2, 3)
                and c1.cout % 8 == 0):
            # implicit 7x7 gather over NHWC4 pixels: no im2col matrix (B x 112 x 112 x 152 bf16 = 122 MB at B = 32)
            x = torch.empty(B * H * W, 4, dtype=dtype, device=image.device)
            call('tell_nchw_to_nhwc4', image.float().contiguous(), x, B, C, H, W)
            x, H, W = stem_bn_act(x, B, H, W, c1, self.bn1, tr)                        # :94-97
        else:
            x = torch.empty(B * H * W, C, dtype=dtype, device=image.device)
            call('tell_nchw_to_nhwc', image.float().contiguous(), x, B, C, H, W, hip.dt(dtype))
            x, H, W = conv_bn_act(x, B, H, W, c1, self.bn1, True, None, tr)            # :94-97
        OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty(B * OH * OW, x.shape[1], dtype=dtype, device=x.device)
        call('tell_maxpool3x3s2', x, y, B, H, W, x.shape[1], OH, OW, hip.dt(dtype))    # :98
        x, H, W = y, OH, OW
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):              # :101-108
            for block in stage:
                x, H, W = block.run(x, B, H, W, tr)
        return x.view(B, H * W, self.out_channels)


def resnet152(**kwargs):
    """resnet.py:184-192 without the pretrained-weight download (no network here): weights are
    the reference's own initialisers unless a checkpoint is loaded."""
    return ResNetFeatureExtractor((3, 8, 36, 3), **kwargs)

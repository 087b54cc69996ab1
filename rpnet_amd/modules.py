"""Host-side mirror of the reference's nn.Module surface for the RP-Net hot path.

Same class names, constructor/forward signatures, yaml keys and state_dict keys as
/root/reference/net/{modules,unet,rp_net}.py, so `test_rpnet.py` and reference
checkpoints drop in unchanged; the arithmetic runs in librpnet_hip.so (fp32 MFMA /
LDS-tiled HIP kernels) through rpnet_amd.functional.  The nn.Conv2d / nn.BatchNorm2d
children are parameter containers only — they are never called.

Internal activation layout is NHWC; NCHW tensors appear only at the module boundary.
"""
import itertools
import os

import torch
import torch.nn as nn

from . import functional as RF
from .schedule import Schedule

# one-pass gradient fan-in for the feature maps with many consumers; _FANIN = 0 (or Schedule.fanin) leaves the fan-in to autograd's pairwise
# adds, 1 = RF.FanOut / RF.SplitRows only (round 4), 2 (default) = + both halves of the encoder output summed straight into
# their rows (RF.SplitFan) and the skip-connection gradients added inside the max-pool backward (RF.PoolSkip) (A/B switch)
_FANIN = 2
# A/B switches of two launch / traffic savings (both on by default): all operand packs of a forward in one launch, and
# no fp32 output for conv_block's first layer when its consumer reads fp16 planes
_PREPACK = True
_ZSKIP = True
# f16x2 mode: encoder input pixels per call from which the fp16 planes are used (below: three bf16 planes; see
# RF.set_f16_active).  262144 = batch 2 at 256^2, where the two arithmetics are level.
_F16_MIN_PIXELS = int(os.environ.get("RPNET_F16_MIN_PIXELS", "262144"))
# the same threshold for eval-mode calls, where the fp16 tensor scales have to be MEASURED (one split pass per layer on top
# of the convolution): batch 2 at 256^2 (the reference driver's call) is small-grid bound and 5 % slower with it (5.33 vs
# 5.07 ms), batch 8 is 20 % faster (9.8 vs 11.8 ms), batch 32 37 % (32.5 vs 44.5 ms).  0 = as in training.
_F16_MIN_PIXELS_EVAL = int(os.environ.get("RPNET_F16_MIN_PIXELS_EVAL", "524288"))
# eval-mode CRE: w_q on a second HIP stream beside w_k while a call has at most this many feature pixels (B h w)
_CRE_STREAMS = True
_CRE_STREAMS_MAX_PIXELS = 16384
# train-mode CRE: w_q (convolution, BatchNorm + ReLU and, through autograd, their backward) on its own HIP stream beside
# w_k: each branch's HBM-bound BatchNorm passes run beside the other branch's convolution — 18.11 -> 17.87 ms per batch-8
# step, configs[4] 33.65 -> 33.34 ms (two alternations on one box, round 3); same kernels, same bits.
# RPNET_CRE_STREAMS_TRAIN=0: both branches on the caller's stream (A/B switch)
_CRE_STREAMS_TRAIN = os.environ.get("RPNET_CRE_STREAMS_TRAIN", "1") == "1"
# train-mode encoder: the support and the query call as two chains on two streams: 0 = never, 1 (default) = when they are
# separate calls anyway (multi-shot / multi-way) AND of comparable length (support images <= 2 x query images), 2 = always,
# also for 1-way 1-shot (two half-size launches per layer instead of one).  Measured, two alternations on one box
# (round 3): configs[4] (2-way 512^2: 8 + 4 images) 33.81 -> 33.02 ms; configs[2] (5-shot: 80 + 16 images, the
# query chain ends after a fifth of the support chain) 84.3 -> 85.0 ms; configs[1] in two half-size launches per layer
# 17.95 -> 18.54 ms (the 16^2 / 32^2 levels no longer fill the machine) — hence the default
# (mode 3 of round 4 — only the 256^2 .. 64^2 levels as two chains — measured 17.73 against 17.55 ms and was removed in round 5)
_ENC_STREAMS = int(os.environ.get("RPNET_ENC_STREAMS", "1"))
_NET_SERIAL = itertools.count(1)      # a never-reused number per RP_Net instance (id() is reused after garbage collection)


def _to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _to_nchw(x):
    return x.permute(0, 3, 1, 2)


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(ch) as the reference builds it — getattr(nn, normalization_type)(ch_out), net/modules.py:48,51,68 with
    unet_normalize_type: InstanceNorm2d — i.e. affine=False, track_running_stats=False: per-image, per-channel statistics in
    train AND eval mode, no parameters and no state_dict entries.  Runs on the BatchNorm kernels with ONE STATISTIC GROUP PER
    IMAGE, gamma = 1, beta = 0 (non-persistent buffers; the running-statistic buffers exist only because the kernels
    update them and are never read)."""
    instance = True

    def __init__(self, ch):
        super().__init__()
        self.num_features = ch
        for name, val in (("weight", torch.ones(ch)), ("bias", torch.zeros(ch)), ("running_mean", torch.zeros(ch)),
                          ("running_var", torch.ones(ch)), ("num_batches_tracked", torch.zeros((), dtype=torch.long))):
            self.register_buffer(name, val, persistent=False)


def _norm(normalization_type, ch):
    if normalization_type == "InstanceNorm2d":
        return InstanceNorm2d(ch)
    if normalization_type != "BatchNorm2d":
        raise NotImplementedError(f"unet_normalize_type={normalization_type!r}: BatchNorm2d (yamls/example.yml:41) and "
                                  "InstanceNorm2d have HIP kernels")
    return nn.BatchNorm2d(ch)


def _norm_mode(norm, training, groups, x):
    """(training flag, statistic groups) a conv + norm + ReLU layer runs with: an instance norm always takes the statistics of
    its own input, one group per image"""
    if getattr(norm, "instance", False):
        return True, x.shape[0]
    return training, groups


class conv_block(nn.Module):
    """net/modules.py:42-58 — 2 x [Conv3x3(s1,p1,bias) -> BatchNorm2d -> ReLU]."""

    def __init__(self, ch_in, ch_out, normalization_type, kernel=3, padding=1):
        super().__init__()
        assert kernel == 3 and padding == 1
        self.ch_in, self.ch_out = ch_in, ch_out
        self.conv = nn.Sequential(
            nn.Conv2d(ch_in, ch_out, kernel_size=kernel, stride=1, padding=padding, bias=True),
            _norm(normalization_type, ch_out), nn.ReLU(inplace=True),
            nn.Conv2d(ch_out, ch_out, kernel_size=kernel, stride=1, padding=padding, bias=True),
            _norm(normalization_type, ch_out), nn.ReLU(inplace=True))

    def forward_nhwc(self, x0, cache, x1=None, groups=1, out_split=True, split=None, pool=False):
        """RF.Operand (or tensor) in, RF.Operand out; out_split: whether a 3x3 convolution reads the block's output as is
        (RF.conv_bn_relu_op); split: channel ranges of the first layer's packed weight when its sources are padded
        (RF.PackedWeight); pool: return MaxPool2d(2, 2) of the block's output instead of the output — the caller needs
        nothing else of it, and the single consumer of the pooled tensor is an unmasked single-source 3x3 convolution"""
        t, groups = _norm_mode(self.conv[1], self.training, groups, x0)
        # the first layer's output feeds the second convolution and nothing else: on fp16 planes its fp32 form is not written
        x = RF.conv_bn_relu_op(x0, self.conv[0], self.conv[1], cache, t, x1=x1, groups=groups, z_unused=_ZSKIP, split=split)
        if pool:
            return RF.conv_bn_relu_op(x, self.conv[3], self.conv[4], cache, t, groups=groups, out_split=True, z_unused=_ZSKIP,
                                      pool=True)
        return RF.conv_bn_relu_op(x, self.conv[3], self.conv[4], cache, t, groups=groups, out_split=out_split)

    def forward(self, x):
        return _to_nchw(self.forward_nhwc(_to_nhwc(x), RF.WeightCache()).x)


class up_conv(nn.Module):
    """net/modules.py:61-75 — nearest x2 -> Conv3x3 -> BatchNorm2d -> ReLU (up-sampling fused into the gather)."""

    def __init__(self, ch_in, ch_out, normalization_type, kernel=3, padding=1):
        super().__init__()
        assert kernel == 3 and padding == 1
        self.ch_in, self.ch_out = ch_in, ch_out
        self.up = nn.Sequential(
            nn.Upsample(scale_factor=2),
            nn.Conv2d(ch_in, ch_out, kernel_size=kernel, stride=1, padding=padding, bias=True),
            _norm(normalization_type, ch_out), nn.ReLU(inplace=True))

    def forward_nhwc(self, x, cache, groups=1, out_split=True):
        t, groups = _norm_mode(self.up[2], self.training, groups, x)
        return RF.conv_bn_relu_op(x, self.up[1], self.up[2], cache, t, groups=groups, upsample=True, out_split=out_split)

    def forward(self, x):
        return _to_nchw(self.forward_nhwc(_to_nhwc(x), RF.WeightCache()).x)


class Unet_2D(nn.Module):
    """net/unet.py:351-360 (constructor state only; the losses there belong to LGCANet)."""

    def __init__(self, cfg, img_ch=5, output_ch=6, t=2, pretrained=True, resnet_type="resnet18"):
        super().__init__()
        self.cfg = cfg
        self.img_ch, self.t, self.pretrained, self.resnet_type = img_ch, t, pretrained, resnet_type
        self.final_activation = cfg["final_activation"]
        self.output_ch = output_ch

    def set_mode(self, mode):
        assert mode in ["train", "valid", "eval", "test"]
        self.mode = mode
        self.train(mode == "train")


class U_Net(Unet_2D):
    """net/unet.py:393-467 — encoder + half decoder, returns the 1/4-resolution 256-channel `d4`."""

    def __init__(self, cfg, img_ch=1, output_ch=6, resnet_type=None):
        super().__init__(cfg, img_ch, output_ch)
        mfm = cfg["mask_feature_map"]
        if mfm not in (False, None, "no", "x", "x2", "x3"):
            # 'x4' / 'x5' widen Conv4 / Conv5 in the reference's constructor (net/unet.py:416-424) but its forward never
            # concatenates a mask there (:451-455): unreachable in the reference too
            raise NotImplementedError(f"mask_feature_map={mfm!r}: the reference's forward only implements 'x', 'x2', 'x3' "
                                      "(net/unet.py:437-449)")
        self.mask_feature_map = mfm if mfm in ("x", "x2", "x3") else False
        nt = cfg["unet_normalize_type"]
        self.Maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        f = [64, 128, 256, 512, 1024]
        # the mask as one more input channel of Conv1 / Conv2 / Conv3 (net/unet.py:401-414)
        self.Conv1 = conv_block(self.img_ch + (mfm == "x"), f[0], nt)
        self.Conv2 = conv_block(f[0] + (mfm == "x2"), f[1], nt)
        self.Conv3 = conv_block(f[1] + (mfm == "x3"), f[2], nt)
        self.Conv4 = conv_block(f[2], f[3], nt)
        self.Conv5 = conv_block(f[3], f[4], nt)
        self.Up5 = up_conv(f[4], f[3], nt)
        self.Up_conv5 = conv_block(f[3] * 2, f[3], nt)
        self.Up4 = up_conv(f[3], f[2], nt)
        self.Up_conv4 = conv_block(f[2] * 2, f[2], nt)

    def _mask_source(self, mask, scale):
        """the mask channel of mask_feature_map 'x2' / 'x3' as a second, 64-channel conv source (channel 0 = the mask
        average-pooled by `scale`, net/unet.py:444,449; the other 63 channels and their weight rows are zero); |mask| <= 1
        bounds it: fp16 tensor scale 2^-15"""
        m = RF.mask_avgpool(mask, scale) if scale > 1 else mask
        src = torch.zeros(*m.shape, 64, device=m.device, dtype=torch.float32)
        src[..., 0] = m
        return RF.Operand(src, scale=torch.full((1,), 2.0 ** -15, device=m.device, dtype=torch.float32))

    def forward_nhwc(self, x, cache, groups=1, mask=None):
        """x [N,H,W,1]; `groups` consecutive image groups keep separate BatchNorm statistics
        (= that many reference calls, in order); mask [N,H,W]: used only with mask_feature_map 'x' / 'x2' / 'x3'.
        Returns the RF.Operand of d4 (tensor `.x`, fp16 tensor scale `.scale`)."""
        if x.shape[1] % 16 or x.shape[2] % 16:
            raise ValueError(f"U_Net needs H, W multiples of 16, got {tuple(x.shape[1:3])}")
        mfm = self.mask_feature_map
        if mfm and mask is None:
            raise ValueError(f"mask_feature_map={mfm!r} needs the mask")
        pool = RF.maxpool2
        # f16x2 training: pooled, concatenated and masked consumers split the fp32 tensor themselves (one joint tensor
        # scale per convolution), so those producers skip their own operand planes
        # (eval mode too: the measured-scale branch of rpnet_bn... emits only the scale — the planes of x1..x4, d5 and the
        # up-convolution outputs would be written and never read)
        sk = "scale" if RF.f16_mode() else True
        # x1 and x2 feed nothing but their pool (:442-448; x3 and x4 are also skip connections): in training the pooled
        # tensor comes out of the block's last BatchNorm + ReLU pass (p1 / p2 are then already pooled) unless a mask
        # channel is concatenated behind the pool
        f1, f2 = self.training and mfm != "x2", self.training and mfm != "x3"
        if mfm == "x":      # cat([x, mask], 1) (net/unet.py:437-438): image and mask as channels 0 / 1 of a 64-channel source
            xin = torch.zeros(*x.shape[:3], 64, device=x.device, dtype=torch.float32)
            xin[..., 0], xin[..., 1] = x[..., 0], mask.float()
            p1 = self.Conv1.forward_nhwc(xin, cache, groups=groups, out_split=sk, split=(2, 64, 64), pool=f1)
        else:
            p1 = self.Conv1.forward_nhwc(x, cache, groups=groups, out_split=sk, pool=f1)
        if not f1:
            p1 = pool(p1)
        if mfm == "x2":     # cat([pool(x1), avg_pool2d(mask, 2)], 1) (:443-444) as two sources
            p2 = self.Conv2.forward_nhwc(p1, cache, x1=self._mask_source(mask.float(), 2), groups=groups, out_split=sk,
                                         split=(64, 64, 128), pool=f2)
        else:
            p2 = self.Conv2.forward_nhwc(p1, cache, groups=groups, out_split=sk, pool=f2)
        if not f2:
            p2 = pool(p2)
        if mfm == "x3":     # cat([pool(x2), avg_pool2d(mask, 4)], 1) (:448-449)
            x3 = self.Conv3.forward_nhwc(p2, cache, x1=self._mask_source(mask.float(), 4), groups=groups, out_split=sk,
                                         split=(128, 128, 192))
        else:
            x3 = self.Conv3.forward_nhwc(p2, cache, groups=groups, out_split=sk)
        # x3 and x4 feed their pool AND a skip connection: one backward pass for both gradients (RF.PoolSkip)
        sch = getattr(self, "schedule", None)      # (the owning RP_Net's options; a stand-alone encoder follows the process defaults)
        fan = (_FANIN if sch is None else sch.get("fanin", _FANIN)) >= 2 and torch.is_grad_enabled() and x3.x.requires_grad
        x3, p3 = RF.pool_skip(x3) if fan else (x3, pool(x3))
        x4 = self.Conv4.forward_nhwc(p3, cache, groups=groups, out_split=sk)
        x4, p4 = RF.pool_skip(x4) if fan else (x4, pool(x4))
        x5 = self.Conv5.forward_nhwc(p4, cache, groups=groups)
        d5 = self.Up5.forward_nhwc(x5, cache, groups=groups, out_split=sk)
        d5 = self.Up_conv5.forward_nhwc(x4, cache, x1=d5, groups=groups)      # cat((x4, d5), 1) as two sources
        d4 = self.Up4.forward_nhwc(d5, cache, groups=groups, out_split=sk)
        return self.Up_conv4.forward_nhwc(x3, cache, x1=d4, groups=groups, out_split=sk)    # cat((x3, d4), 1)

    def forward(self, x, mask=None, do_last_conv=True):
        n, c, h, w = x.shape
        assert c == 1
        m = None if mask is None else mask.reshape(n, h, w)
        return {"d4": _to_nchw(self.forward_nhwc(x.reshape(n, h, w, 1), RF.WeightCache(), mask=m).x)}


class Encoder(nn.Module):
    """net/vgg.py:8-74 — VGG16-style (conv3x3, ReLU) stack, stride 8, last block dilated by 2 and
    without its final ReLU.  Same constructor / forward signature / `features.*` state_dict keys.
    (RP_Net cannot use it in the reference either: net/rp_net.py:248-249 indexes its tensor
    output with ['d4'] — kept as a standalone module on the shared conv kernels.)"""

    def __init__(self, in_channels=3, pretrained_path=None):
        super().__init__()
        self.pretrained_path = pretrained_path
        self.features = nn.Sequential(
            self._make_layer(2, in_channels, 64), nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            self._make_layer(2, 64, 128), nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            self._make_layer(3, 128, 256), nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            self._make_layer(3, 256, 512), nn.MaxPool2d(kernel_size=3, stride=1, padding=1),
            self._make_layer(3, 512, 512, dilation=2, lastRelu=False))
        self._init_weights()

    def _make_layer(self, n_convs, in_channels, out_channels, dilation=1, lastRelu=True):
        layer = []
        for i in range(n_convs):
            layer.append(nn.Conv2d(in_channels, out_channels, kernel_size=3, dilation=dilation, padding=dilation))
            if i != n_convs - 1 or lastRelu:
                layer.append(nn.ReLU(inplace=True))
            in_channels = out_channels
        return nn.Sequential(*layer)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
        if self.pretrained_path is not None:   # net/vgg.py:65-74: first 26 tensors of a VGG16 checkpoint
            dic = torch.load(self.pretrained_path, map_location="cpu")
            keys, new_dic = list(dic.keys()), self.state_dict()
            new_keys = list(new_dic.keys())
            for i in range(26):
                new_dic[new_keys[i]] = dic[keys[i]]
            self.load_state_dict(new_dic)

    def forward(self, x, mask=None):
        cache = RF.WeightCache()
        n, c, h, w = x.shape
        cg = 64                                     # packed Cin of the first conv: 3 real + zero channels
        a = torch.zeros(n, h, w, cg, device=x.device, dtype=torch.float32)
        a[..., :c] = x.permute(0, 2, 3, 1)
        for block in self.features:
            if isinstance(block, nn.MaxPool2d):
                a = RF.MaxPool3.apply(a, block.stride)
                continue
            mods = list(block)
            for i, m in enumerate(mods):
                if not isinstance(m, nn.Conv2d):
                    continue
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                split = (m.in_channels, cg) if m.in_channels < 32 else None
                a = RF.ConvRelu.apply(a, m.weight, m.bias, cache.get(m.weight, split), relu, m.dilation[0])
        return a.permute(0, 3, 1, 2)


class ContextCorrelationEncoder(nn.Module):
    """net/rp_net.py:45-84.  w_context / out are constructed (state_dict parity) and never used."""

    def __init__(self, cfg, in_channels=3, radius=5):
        super().__init__()
        self.radius = cfg["mask_refinement_correlation_radius"]
        num_feat = 64
        cbr = lambda ci, co, k: nn.Sequential(nn.Conv2d(ci, co, k, padding=k // 2), nn.BatchNorm2d(co),  # noqa: E731
                                              nn.ReLU(inplace=True))
        self.w_k = cbr(in_channels, in_channels, 3)
        self.w_q = cbr(in_channels, in_channels, 3)
        self.w_context = cbr(in_channels * 2, in_channels, 1)
        self.q = cbr(in_channels + (self.radius * 2 + 1) ** 2, num_feat, 1)
        self.out = cbr(2 * in_channels, num_feat, 1)
        if not 1 <= self.radius <= 7:
            raise NotImplementedError("mask_refinement_correlation_radius outside 1..7 (the window kernels hold 2 r + 1 <= 15 "
                                      "vertical offsets per block; the yaml ships 5)")

    def forward_masked(self, fts, mask, cache, fts_scale=None, defer_act=False, pre=None):
        """cre(fts*mask, fts*(1-mask)) with the mask multiply fused into the conv gather
        (net/rp_net.py:275,283).  fts [B,h,w,C] NHWC (or a pair of aliases of it, one per convolution: RF.FanOut),
        mask [B,h,w] or None.
        defer_act: return the RF.Operand of the output with cre.q's BatchNorm + ReLU still to be applied by the consumer's
        fused launch (train mode; RF.conv_bn_relu_op(defer_act)) instead of the output tensor.
        pre: (xk, xq) — the operand planes of fts * mask and fts * (1 - mask), already made by the launch that made `mask`
        (RF.CosineMatchUp's fused glue); the two convolutions then take them as they are."""
        t = self.training
        fk, fq = fts if isinstance(fts, tuple) else (fts, fts)
        m1, m2 = (1, 2) if mask is not None else (0, 0)
        sp = "corr" if self.radius == 5 else False       # the correlation then takes the split planes of fm1 / fm2
        # fts_scale: the fp16 tensor scale of the features (their producer's bound; slicing / fan-out keeps it)
        # fm1 / fm2 feed the correlation and (fm1) the 1x1 convolution, both of which read fp16 planes in f16x2 / f16
        # training: their fp32 form is then never written (RF.conv_bn_relu_op(z_unused); a consumer that wanted the
        # values would raise)
        zu = _ZSKIP and sp == "corr" and RF._CORR16 and RF._CONV1X1_SPLIT and self.w_k[0].weight.shape[0] % 128 == 0

        mk, mq = ((mask, 1, pre[0]), (mask, 2, pre[1])) if (pre is not None and mask is not None) else (None, None)

        def w_k():
            return RF.conv_bn_relu_op(RF.Operand(fk, scale=fts_scale, masked=mk), self.w_k[0], self.w_k[1], cache, t, in_scale=mask,
                                      in_mode=m1, out_split=sp, z_unused=zu)

        def w_q():
            return RF.conv_bn_relu_op(RF.Operand(fq, scale=fts_scale, masked=mq), self.w_q[0], self.w_q[1], cache, t, in_scale=mask,
                                      in_mode=m2, out_split=sp, z_unused=zu)

        # inference on a few slices (test_rpnet.py: 2 per call): either convolution is 256 four-wave blocks of a machine
        # that holds 512 — the two are independent, so w_q runs on the side stream beside w_k (one block of each per CU)
        pixels = fk.shape[0] * fk.shape[1] * fk.shape[2]
        sch = getattr(self, "schedule", None)          # (set by the owning RP_Net; a stand-alone encoder follows the process defaults)
        cre_eval = _CRE_STREAMS if sch is None else sch.get("cre_streams_eval", _CRE_STREAMS)
        cre_train = _CRE_STREAMS_TRAIN if sch is None else sch.get("cre_streams_train", _CRE_STREAMS_TRAIN)
        two_eval = cre_eval and not t and not torch.is_grad_enabled() and fk.is_cuda and pixels <= _CRE_STREAMS_MAX_PIXELS
        two_train = cre_train and t and fk.is_cuda
        if two_eval or two_train:
            main = torch.cuda.current_stream(fk.device)
            side = RF._cre_stream(fk.device) if two_train else RF._side_stream(fk.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fm2 = w_q()
            if two_train or pre is not None:
                # read by the side stream (the caching allocator must not hand their blocks out before it is done).  The operand
                # planes `pre` come from the glue launch on the MAIN stream; w_q reads pre[1] on the side stream in forward and,
                # through the saved operand of its weight gradient, in backward — with nothing else keeping it alive once the
                # backward node has returned (round-5 advisor finding)
                for tns in (fq, mask, fts_scale) + (tuple(pre) if pre is not None else ()):
                    if tns is not None:
                        tns.record_stream(side)
            fm1 = w_k()
            main.wait_stream(side)
            for tns in (fm2.x, fm2.p16, fm2.pbf, fm2.scale):     # allocated on the side stream, read on the main one from here on
                if tns is not None:
                    tns.record_stream(main)
        else:
            fm1, fm2 = w_k(), w_q()
        out = self._tail(fm1, fm2, cache, defer_act)
        return out if defer_act else out.x

    def _tail(self, fm1, fm2, cache, defer_act=False):
        corr, fm1b = RF.local_corr(fm1, fm2, self.radius)     # fm1b: alias of fm1, gradient fan-in fused (RF.LocalCorr)
        kk = (2 * self.radius + 1) ** 2
        return RF.conv_bn_relu_op(corr, self.q[0], self.q[1], cache, self.training, x1=fm1b, split=(kk, RF.corr_stride(self.radius)),
                                  out_split=False, defer_act=defer_act)

    def forward(self, fm1, fm2):
        cache = RF.WeightCache()
        t = self.training
        sp = "corr" if self.radius == 5 else False
        a = RF.conv_bn_relu_op(_to_nhwc(fm1), self.w_k[0], self.w_k[1], cache, t, out_split=sp)
        b = RF.conv_bn_relu_op(_to_nhwc(fm2), self.w_q[0], self.w_q[1], cache, t, out_split=sp)
        return _to_nchw(self._tail(a, b, cache).x)


class RP_Net(nn.Module):
    """net/rp_net.py:184-440 — few-shot segmentation with recurrent mask refinement.

    cfg: {'align': bool, 'backbone': 'UNet'}; backbone_cfg: the whole yaml dict.
    """

    def __init__(self, in_channels=3, pretrained_path=None, cfg=None, backbone_cfg=None):
        super().__init__()
        self.pretrained_path = pretrained_path
        self.config = cfg or {"align": False}
        self.backbone_cfg = backbone_cfg
        # serving switch (off by default): in eval mode keep the packed weights and folded BatchNorm affines between
        # calls; the caller promises not to change parameters / running statistics meanwhile (rpnet_amd.graph sets it:
        # a captured graph assumes static weights anyway).  `net._cache.clear()` drops the packs.
        self.freeze_packs = False
        # test / diagnostic hook (off by default): teacher forcing of the refinement loop — {i: mask [B,h,w]} replaces the
        # mask fed INTO iteration i (the loop's own thresholded prediction of iteration i-1, net/rp_net.py:308-311), so
        # that one flipped pixel at the 0.5 threshold cannot compound across iterations when two arithmetics are compared
        self.forced_masks = None
        # test hook (off by default): a dict that forward fills with detached NCHW views of the stage-boundary tensors
        # of SURVEY.md §3.2 — supp_d4, qry_d4, supp_fts[wa][s], protos, inter_i — for comparison with the golden fixtures
        self.taps = None
        self.scale = backbone_cfg.get("scale", 4)
        self.num_iter = backbone_cfg["n_iter_refinement"]
        self.use_relation_enc = backbone_cfg.get("use_relation_enc", "relation")
        backbone = self.config.get("backbone")
        if backbone == "UNet":
            self.encoder = U_Net(backbone_cfg)
            num_feat = 256
            if pretrained_path:
                dic = torch.load(self.pretrained_path, map_location="cpu")["state_dict"]
                self.load_state_dict(dic, strict=False)  # the reference loads before `cre` exists (:212-214)
        elif backbone in ("vgg", "resnet"):
            raise NotImplementedError(f"backbone={backbone!r}: unreachable in the reference too "
                                      "(vgg.Encoder returns a tensor but RP_Net indexes ['d4'], net/rp_net.py:248-249)")
        else:
            raise NotImplementedError
        if self.use_relation_enc != "relation":
            raise NotImplementedError("use_relation_enc='concat' needs SimpleConcat, which the reference never defines "
                                      "(net/rp_net.py:224)")
        if self.scale != 4:
            raise NotImplementedError("scale != 4: the UNet d4 map is 1/4 resolution")
        self.cre = ContextCorrelationEncoder(backbone_cfg, in_channels=num_feat)
        # this model's own execution options (arithmetic, stream layout, fan-in, zero-tile skip); every field None = the
        # process-wide default at call time.  See rpnet_amd/schedule.py
        self.schedule = Schedule()
        for m in (self.cre, self.encoder):
            object.__setattr__(m, "schedule", self.schedule)         # (shared object, not a registered sub-module / buffer)
        self._cache = RF.WeightCache()
        self._serial = next(_NET_SERIAL)      # part of the key of this module's eval-mode fp16 scale history (RF.pred_*)

    # ---------------------------------------------------------------- forward
    def forward(self, supp_imgs, fore_mask, back_mask, qry_imgs, registration_field=None, grid=None,
                query_labels=None, appr_query_labels=None):
        sch = self.schedule
        with RF.scope(conv_math=sch.conv_math, async_wgrad=sch.async_wgrad, mask_skip=sch.mask_skip):
            return self._forward(supp_imgs, fore_mask, back_mask, qry_imgs, registration_field, grid, query_labels, appr_query_labels)

    def _forward(self, supp_imgs, fore_mask, back_mask, qry_imgs, registration_field=None, grid=None,
                 query_labels=None, appr_query_labels=None):
        sch = self.schedule
        n_ways, n_shots, n_queries = len(supp_imgs), len(supp_imgs[0]), len(qry_imgs)
        if n_queries != 1:
            raise NotImplementedError("n_queries != 1 (the reference's reader only produces one, few_shot_reader.py:80)")
        if self.num_iter < 1:
            raise ValueError("n_iter_refinement must be >= 1")
        B = supp_imgs[0][0].shape[0]
        H, W = qry_imgs[0].shape[-2:]
        h, w = H // self.scale, W // self.scale
        cache = self._cache
        RF.reset_async()
        if self.training or not self.freeze_packs:
            cache.clear()  # packed weights live for one forward only (never reused across optimizer steps)

        # ---- features: support and query through the encoder, separate BN statistics (:245-258)
        # (one way, one shot: the support tensor itself — torch.cat of a single tensor is a copy launch each)
        supp = (supp_imgs[0][0] if n_ways * n_shots == 1 else torch.cat([torch.cat(way, 0) for way in supp_imgs], 0)).float()
        qry = qry_imgs[0].float()
        ns = supp.shape[0]
        # eval mode: with predicted scales (RF.pred_*: the previous call's maxima) the fp16 planes cost no extra pass, so they
        # pay from the training threshold on; RF._EVAL_PREDICT = False keeps round 2's measured scales and their higher threshold
        f16_min, f16_min_eval = sch.get("f16_min_pixels", _F16_MIN_PIXELS), sch.get("f16_min_pixels_eval", _F16_MIN_PIXELS_EVAL)
        thr = f16_min if (self.training or not f16_min_eval or not f16_min or RF._EVAL_PREDICT) else f16_min_eval
        RF.set_f16_active((ns + B) * H * W >= thr)      # f16x2 mode: fp16 planes only where they pay
        pred_key = None
        if RF.f16_mode():
            RF.reset_absmax_pool(supp.device)       # measured fp16 scales (eval-mode layers, the correlation): one fill per forward
            if not self.training and not torch.is_grad_enabled() and RF._EVAL_PREDICT:
                pred_key = (self._serial, RF.conv_math(), ns, B, H, W, self.num_iter, n_ways, n_shots, self.forced_masks is not None)
                RF.pred_begin(supp.device, pred_key, allow=not getattr(self, "_pred_redo", False))
        planes = RF.pack_planes()
        if _PREPACK and planes and (self.training or not self.freeze_packs):
            # every 3x3 layer's operand pack of this forward in one launch per kernel instead of two launches per layer
            # (the two up_conv layers on their collapsed four-tap packs, RF._UP4)
            # (round 6: eval mode too — the collapsed form's epilogue carries the folded BatchNorm affine)
            ups = ()
            if self.training or RF.f16_mode():
                # only where every encoder call of this forward takes the collapsed form (the calls see ns + B images, or ns and B)
                enc_n = (ns + B,) if ns == B else (ns, B)
                ups = tuple(m.up[1].weight for m, f in ((self.encoder.Up5, 8), (self.encoder.Up4, 4))
                            if RF.up4_layer_ok(m.up[1].weight, planes, [(n, H // f, W // f) for n in enc_n + ((ns, B) if ns == B else ())]))
            if self.training and supp.is_cuda:      # the packing on its own stream beside the first-layer convolution
                cache.prepack_async(self._pack_weights(), planes, supp.device, ups)
            else:
                cache.prepack(self._pack_weights(), planes, ups)
        # both encoder calls of the reference get the SUPPORT foreground mask of way 0 / shot 0 (net/rp_net.py:248,257)
        enc_mask = fore_mask[0][0].float() if self.encoder.mask_feature_map else None
        if enc_mask is not None and ns != B:
            raise NotImplementedError("mask_feature_map with more than one support image per episode: the reference "
                                      "concatenates B masks onto Wa*Sh*B images (net/unet.py:438) and fails")
        # training: the support call and the query call of the encoder (two calls in the reference, net/rp_net.py:248,257) as
        # two chains on two HIP streams — each chain's statistics-finalize and BatchNorm + ReLU passes run beside the other
        # chain's convolution, in backward likewise; the launches that touch a BatchNorm module's running statistics or
        # parameter gradients keep the order of the two calls (RF.order_begin).  Policy and measurements: _ENC_STREAMS
        enc_streams, fanin = sch.get("enc_streams", _ENC_STREAMS), sch.get("fanin", _FANIN)
        two_chains = (self.training and supp.is_cuda and enc_mask is None and
                      (enc_streams == 2 or (enc_streams == 1 and ns != B and ns <= 2 * B)))
        RF.order_begin(two_chains)
        split_fan = False
        if ns == B and not two_chains:
            d4 = self.encoder.forward_nhwc(torch.cat([supp, qry], 0).reshape(ns + B, H, W, 1), cache, groups=2,
                                           mask=None if enc_mask is None else torch.cat([enc_mask, enc_mask], 0))
            s_supp = s_qry = d4.scale      # fp16 tensor scale of the features (f16x2 / f16 training): both halves keep it
            d4 = d4.x
            # with the gradient fan-in of both halves (RF.SplitFan): the two 3x3 convolutions of the support CRE call, the
            # 2 T of the refinement loop (1-way 1-shot: ns == B, one CRE call on the support features)
            split_fan = d4.requires_grad and fanin >= 2 and n_ways * n_shots == 1
            if split_fan:
                uses = RF.SplitFan.apply(d4, ns, 2, 2 * self.num_iter)
                supp_d4, qry_d4, fan_supp, fan_qry = uses[0], uses[2], uses[:2], uses[2:]
            else:
                supp_d4, qry_d4 = RF.SplitRows.apply(d4, ns) if (d4.requires_grad and fanin) else (d4[:ns], d4[ns:])
        elif two_chains:
            main, side = torch.cuda.current_stream(supp.device), RF._cre_stream(supp.device)
            qin = qry.reshape(B, H, W, 1)
            # both chains read the SAME cached packs: whatever the prepack above did not make (fp32 arithmetic, _PREPACK = False)
            # is made here, on the main stream, in front of the fork
            cache.materialize([w for w in self._pack_weights() if w is not self.cre.w_k[0].weight and w is not self.cre.w_q[0].weight],
                              planes)
            side.wait_stream(main)
            o_s = self.encoder.forward_nhwc(supp.reshape(ns, H, W, 1), cache)
            with torch.cuda.stream(side):
                o_q = self.encoder.forward_nhwc(qin, cache)
            qin.record_stream(side)
            main.wait_stream(side)
            for tns in (o_q.x, o_q.p16, o_q.pbf, o_q.scale):
                if tns is not None:
                    tns.record_stream(main)
            (supp_d4, s_supp), (qry_d4, s_qry) = (o_s.x, o_s.scale), (o_q.x, o_q.scale)
        else:
            o_s = self.encoder.forward_nhwc(supp.reshape(ns, H, W, 1), cache)
            o_q = self.encoder.forward_nhwc(qry.reshape(B, H, W, 1), cache)
            (supp_d4, s_supp), (qry_d4, s_qry) = (o_s.x, o_s.scale), (o_q.x, o_q.scale)
        taps = self.taps
        if taps is not None:
            taps["supp_d4"] = supp_d4.detach().reshape(n_ways, n_shots, B, h, w, -1).permute(0, 1, 2, 5, 3, 4)
            taps["qry_d4"] = _to_nchw(qry_d4.detach())
        # per (way, shot) views: unbind's backward is ONE stack (indexing [wa, s] would zero-fill and copy per slice)
        per = (supp_d4.reshape(B, h, w, -1),) if ns == B else supp_d4.reshape(n_ways * n_shots, B, h, w, -1).unbind(0)
        supp_d4 = [[per[wa * n_shots + s] for s in range(n_shots)] for wa in range(n_ways)]

        # ---- support relation features, per (way, shot) with that shot's own mask (:269-275)
        fore = [[m.float().contiguous() for m in way] for way in fore_mask]
        back = [[m.float().contiguous() for m in way] for way in back_mask]
        supp_fts = [[self.cre.forward_masked(tuple(fan_supp) if split_fan else supp_d4[wa][s],
                                             RF.mask_avgpool(fore[wa][s], self.scale), cache, s_supp)
                     for s in range(n_shots)] for wa in range(n_ways)]

        # ---- prototypes: constant across iterations, computed once (:288-300)
        fg_protos, bg_sum = [], 0
        for wa in range(n_ways):
            fg_w, bg_w = 0, 0
            for s in range(n_shots):
                am, msum = RF.mask_adjoint(torch.stack([back[wa][s], fore[wa][s]], 0), h, w)
                p = RF.MaskedPool.apply(supp_fts[wa][s], am, msum)          # [B,2,C]: bg, fg
                if n_ways * n_shots > 1:
                    bg_w, fg_w = bg_w + p[:, 0], fg_w + p[:, 1]
            if n_ways * n_shots > 1:
                fg_protos.append(fg_w / n_shots)
                bg_sum = bg_sum + bg_w / n_shots
        # 1-way 1-shot: the means over one shot / one way are the pooled rows themselves ([bg, fg]): no slicing, no stack
        protos = p if n_ways * n_shots == 1 else torch.stack([bg_sum / n_ways] + fg_protos, 1).contiguous()  # [B,1+Wa,C]
        if taps is not None:
            taps["supp_fts"] = [[_to_nchw(f.detach()) for f in way] for way in supp_fts]
            taps["protos"] = protos.detach()

        # ---- refinement loop (:281-312)
        soft = self.backbone_cfg["soft_mask"] != False  # noqa: E712 (the reference compares with ==)
        qry_mask = RF.mask_avgpool(appr_query_labels.float(), self.scale)
        refinement = {}
        inter = pred = None
        # the query features feed 2 T convolutions: one-pass gradient fan-in instead of autograd's chain of adds
        T = self.num_iter
        if split_fan:
            qry_uses = fan_qry
        else:
            qry_uses = RF.FanOut.apply(qry_d4, 2 * T) if (qry_d4.requires_grad and fanin) else (qry_d4,) * (2 * T)
        # the loop's glue — cre.q's BatchNorm + ReLU, the cosine match, the bilinear x4, softmax / threshold / 4x4 average and the
        # operand planes of qry * mask, qry * (1 - mask) for the next iteration — is ONE launch per iteration where the shapes
        # fit (RF.CosineMatchUp / rpnet_refine_glue_fwd); a differentiable mask (soft_mask in training) keeps the separate path
        K = 1 + n_ways
        # the prototypes feed the T matches of the loop: one-pass gradient fan-in as for the query features
        proto_uses = RF.FanOut.apply(protos, T) if (T > 1 and protos.requires_grad and fanin) else (protos,) * T
        soft_grad = soft and torch.is_grad_enabled()
        cq = self.cre.q[0].out_channels           # width of the relation features the glue matches against the prototypes
        fuse = qry_d4.is_cuda and RF.glue_supported(K, h, w, H, W, cq) and not soft_grad
        # planes of the masked query features the two 3x3 convolutions of the NEXT call read: fp16 (the feature scale is known)
        # or three bf16; 0 = they gather fp32 values with the mask factor themselves
        xplanes = 0
        if fuse and RF.pack_planes() and qry_d4.shape[-1] % 64 == 0:
            xplanes = RF.pack_planes() if (RF.f16_mode() and s_qry is not None) else RF._MATH["planes"]
            if not RF.glue_supported(K, h, w, H, W, cq, qry_d4.shape[-1], xplanes):
                xplanes = 0
        pre = None
        for i in range(T):
            if self.forced_masks is not None and i in self.forced_masks:
                qry_mask, pre = self.forced_masks[i].float().contiguous(), None
            if fuse:
                io = self.cre.forward_masked((qry_uses[2 * i], qry_uses[2 * i + 1]), qry_mask, cache, s_qry, defer_act=True, pre=pre)
                inter = io.x
                last = i + 1 == T
                forced_next = self.forced_masks is not None and (i + 1) in self.forced_masks
                ex = {"deferred": io.deferred, "mask": not last, "soft": bool(soft)}
                if not last and xplanes and not forced_next:
                    ex.update(x=qry_d4, x_scale=s_qry if xplanes <= 2 else None, planes=xplanes)
                logits, pred = RF.CosineMatchUp.apply(inter, proto_uses[i], H, W, 20.0, ex)
                if not last:
                    qry_mask = ex["mask_out"]
                    pre = (ex["xk"], ex["xq"]) if ex.get("xk") is not None else None
            else:
                inter = self.cre.forward_masked((qry_uses[2 * i], qry_uses[2 * i + 1]), qry_mask, cache, s_qry)
                logits, pred = RF.CosineMatchUp.apply(inter, proto_uses[i], H, W, 20.0)
            if taps is not None:
                taps[f"inter_{i}"] = _to_nchw(inter.detach())
            refinement[i] = logits
            if i + 1 == T:
                break              # the mask of the last iteration's output feeds nothing (net/rp_net.py:308-311 computes it; unused)
            if fuse:
                continue
            if soft_grad:   # soft_mask: the gradient flows through the fed-back mask
                qry_mask = RF.SoftmaxPool.apply(logits, self.scale)
            else:
                qry_mask = RF.softmax_thresh_pool(logits, self.scale, soft)
        # final pass (:314-337) recomputes refinement[T-1] bit for bit: alias it
        output = refinement[self.num_iter - 1]

        align_loss = 0
        if self.config["align"] and self.training:
            align_loss = self.alignLoss(inter, pred, supp_fts, fore, back)
        if pred_key is not None and RF.pred_end(supp.device, pred_key):
            # a layer's maximum exceeded the bound predicted from the previous call: redo this call on measured scales
            self._pred_redo = True
            try:
                return self.forward(supp_imgs, fore_mask, back_mask, qry_imgs, registration_field, grid, query_labels,
                                    appr_query_labels)
            finally:
                self._pred_redo = False
        return {"output": output, "align_loss": align_loss, "refinement": refinement}

    def _pack_weights(self):
        ws = getattr(self, "_pack_list", None)
        if ws is None:
            enc = self.encoder
            convs = [m for blk in (enc.Conv1, enc.Conv2, enc.Conv3, enc.Conv4, enc.Conv5, enc.Up_conv5, enc.Up_conv4)
                     for m in (blk.conv[0], blk.conv[3])] + [enc.Up5.up[1], enc.Up4.up[1], self.cre.w_k[0], self.cre.w_q[0]]
            # (layers whose channel ranges are padded — mask_feature_map — pack on first use with their own ranges)
            ws = self._pack_list = [m.weight for m in convs if m.weight.shape[1] >= 32 and m.weight.shape[1] % 32 == 0]
        return ws

    def alignLoss(self, qry_fts, pred, supp_fts, fore_mask, back_mask):
        """net/rp_net.py:394-440, all B episodes at once; returns sum_epi(loss_epi) / B (:349).
        qry_fts [B,h,w,C]; pred [B,1+Wa,h,w]; supp_fts[wa][s] [B,h,w,C]; masks[wa][s] [B,H,W]."""
        n_ways, n_shots = len(fore_mask), len(fore_mask[0])
        H, W = fore_mask[0][0].shape[-2:]
        masks, counts, keep_all = RF.argmax_masks(pred, want_keep=True)      # :412-415; keep [K,B]: skip_ways (:414,421), per episode
        qp = RF.MaskedPool.apply(qry_fts, masks, counts)                     # :416-417  [B,1+Wa,C]
        loss = 0
        for wa in range(n_ways):
            keep = keep_all[wa + 1]
            protos = qp if n_ways == 1 else torch.stack([qp[:, 0], qp[:, wa + 1]], 1).contiguous()
            for s in range(n_shots):
                logits, _ = RF.CosineMatchUp.apply(supp_fts[wa][s], protos, H, W, 20.0)   # :425-431
                lab = RF.align_labels(fore_mask[wa][s], back_mask[wa][s])                 # :433-436
                term = RF.DiceCE.apply(logits, lab, False, 255, True, keep)
                # (one way, one shot: x / 1 / 1 and 0 + x are x — three scalar launches forward, two backward)
                loss = term if n_ways * n_shots == 1 else loss + term / n_shots / n_ways
        return loss

    # reference-named helpers kept for API parity (NCHW tensors, single episode)
    def calDist(self, fts, prototype, scaler=20):
        f = _to_nhwc(fts)
        n, h, w, _ = f.shape
        logits, _ = RF.CosineMatchUp.apply(f, prototype.reshape(1, 1, -1).expand(n, 1, -1).contiguous(), h, w, scaler)
        return logits[:, 0]  # H == h: the bilinear stage is the identity

    def getFeatures(self, fts, mask):
        f = _to_nhwc(fts)
        am, msum = RF.mask_adjoint(mask.float().reshape(1, *mask.shape), f.shape[1], f.shape[2])
        return RF.MaskedPool.apply(f, am, msum)[:, 0]

    def getPrototype(self, fg_fts, bg_fts):
        n_ways, n_shots = len(fg_fts), len(fg_fts[0])
        fg = [sum(way) / n_shots for way in fg_fts]
        bg = sum([sum(way) / n_shots for way in bg_fts]) / n_ways
        return fg, bg


model_factory = {"RP_Net": RP_Net}

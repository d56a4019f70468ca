"""FPN / ResNet backbone, 2D and 3D.  Architecture and state_dict keys follow the reference's
models/backbone.py (FPN :22-179, ResBlock :183-206, Interpolate :209-217) so its checkpoints load;
the convolutions run on MIOpen through torch (the north star leaves the conv path on MIOpen).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils import fused_epilogue
from ..utils.fused_epilogue import ConvBias


def _lateral_up(conv, x, coarse):
    """P_conv1(x) + F.interpolate(coarse, scale_factor=2) (backbone.py:147-153): add AND nearest up-sampling inside the conv's epilogue"""
    if isinstance(conv, ConvBias):
        return fused_epilogue.conv_bias_add_upsampled(conv, x, coarse, 2)
    return conv(x) + fused_epilogue.upsample_nearest(coarse, 2)


def _lateral(conv, x, top_down):
    """P_conv1(x) + upsampled coarser level (backbone.py:147-153): the add rides the conv's fused epilogue"""
    if isinstance(conv, ConvBias):
        return conv(x, residual=top_down)
    return conv(x) + top_down



class ResBlock(nn.Module):
    def __init__(self, start_filts, planes, conv, stride=1, downsample=None, norm=None, relu="relu"):
        super(ResBlock, self).__init__()
        self.conv1 = conv(start_filts, planes, ks=1, stride=stride, norm=norm, relu=relu)
        self.conv2 = conv(planes, planes, ks=3, pad=1, norm=norm, relu=relu)
        self.conv3 = conv(planes, planes * 4, ks=1, norm=norm, relu=None)
        self.relu = nn.ReLU(inplace=True) if relu == "relu" else nn.LeakyReLU(inplace=True)
        if downsample is not None:
            self.downsample = conv(downsample[0], downsample[0] * downsample[1], ks=1, stride=downsample[2], norm=norm, relu=None)
        else:
            self.downsample = None
        self.stride = stride

    def forward_subsampled(self, x_s):
        """the block on an input that was sub-sampled by this block's stride already (fused_epilogue.stride_tap): conv1 and downsample
        -- both 1x1, same stride -- run at unit stride on x_s; everything else as in forward"""
        residual = fused_epilogue.conv1x1_unit_stride_bias_act(self.downsample, x_s)
        out = self.conv2(fused_epilogue.conv1x1_unit_stride_bias_act(self.conv1[0], x_s, relu=True))
        return self.conv3(out, residual=residual, relu=True)

    def forward(self, x):
        if self.downsample is None and fused_epilogue.res_tap_applies(self.conv1, x):
            # identity block: x feeds conv1 AND the residual add -- one autograd node owns both paths, so that its backward produces
            # the input gradient already added to the residual gradient (utils/fused_epilogue._Conv1x1ResTap)
            h, residual = fused_epilogue.conv_bias_relu_with_res_tap(self.conv1, x)
            out = self.conv2(h)
        else:
            residual = x if self.downsample is None else self.downsample(x)
            out = self.conv2(self.conv1(x))
        if isinstance(self.conv3, ConvBias) and isinstance(self.relu, nn.ReLU):
            # bias + residual + ReLU of backbone.py:203-205 in one pass
            return self.conv3(out, residual=residual, relu=True)
        out = self.conv3(out) + residual
        return self.relu(out)


class Interpolate(nn.Module):
    def __init__(self, scale_factor, mode):
        super(Interpolate, self).__init__()
        self.scale_factor = scale_factor
        self.mode = mode

    def forward(self, x):
        y = fused_epilogue.upsample2x_yx(x, self.scale_factor, self.mode)       # channels-last x2 (y, x) kernel (csrc/upsample.hip)
        if y is not None:
            return y
        return F.interpolate(x, scale_factor=self.scale_factor, mode=self.mode, align_corners=False)


class FPN(nn.Module):
    def __init__(self, cf, conv, operate_stride1=False):
        super(FPN, self).__init__()
        sf = cf.start_filts
        self.start_filts = sf
        self.n_blocks = [3, 4, {"resnet50": 6, "resnet101": 23}[cf.res_architecture], 3]
        self.block_expansion = 4
        self.operate_stride1 = operate_stride1
        self.sixth_pooling = cf.sixth_pooling
        self.dim = conv.dim
        s1 = (2, 2, 1) if conv.dim == 3 else 2
        if operate_stride1:
            self.C0 = nn.Sequential(conv(cf.n_channels, sf, ks=3, pad=1, norm=cf.norm, relu=cf.relu),
                                    conv(sf, sf, ks=3, pad=1, norm=cf.norm, relu=cf.relu))
            self.C1 = conv(sf, sf, ks=7, stride=s1, pad=3, norm=cf.norm, relu=cf.relu)
        else:
            self.C1 = conv(cf.n_channels, sf, ks=7, stride=s1, pad=3, norm=cf.norm, relu=cf.relu)
        sfe = sf * self.block_expansion

        def stage(c_in, planes, n, stride, first_ds):
            layers = [ResBlock(c_in, planes, conv=conv, stride=stride, norm=cf.norm, relu=cf.relu, downsample=first_ds)]
            for _ in range(1, n):
                layers.append(ResBlock(planes * 4, planes, conv=conv, norm=cf.norm, relu=cf.relu))
            return layers

        pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1) if conv.dim == 2 else \
            fused_epilogue.MaxPool3dStem(kernel_size=3, stride=(2, 2, 1), padding=1)
        self.C2 = nn.Sequential(pool, *stage(sf, sf, self.n_blocks[0], 1, (sf, self.block_expansion, 1)))
        self.C3 = nn.Sequential(*stage(sfe, sf * 2, self.n_blocks[1], 2, (sfe, 2, 2)))
        self.C4 = nn.Sequential(*stage(sfe * 2, sf * 4, self.n_blocks[2], 2, (sfe * 2, 2, 2)))
        self.C5 = nn.Sequential(*stage(sfe * 4, sf * 8, self.n_blocks[3], 2, (sfe * 4, 2, 2)))
        if self.sixth_pooling:
            self.C6 = nn.Sequential(*stage(sfe * 8, sf * 16, self.n_blocks[3], 2, (sfe * 8, 2, 2)))
        if conv.dim == 2:
            self.P1_upsample = Interpolate(scale_factor=2, mode="bilinear")
            self.P2_upsample = Interpolate(scale_factor=2, mode="bilinear")
        else:
            self.P1_upsample = Interpolate(scale_factor=(2, 2, 1), mode="trilinear")
            self.P2_upsample = Interpolate(scale_factor=(2, 2, 1), mode="trilinear")
        oc = cf.end_filts
        self.out_channels = oc
        self.P5_conv1 = conv(sf * 32 + cf.n_latent_dims, oc, ks=1, stride=1, relu=None)
        self.P4_conv1 = conv(sf * 16, oc, ks=1, stride=1, relu=None)
        self.P3_conv1 = conv(sf * 8, oc, ks=1, stride=1, relu=None)
        self.P2_conv1 = conv(sf * 4, oc, ks=1, stride=1, relu=None)
        self.P1_conv1 = conv(sf, oc, ks=1, stride=1, relu=None)       # constructed unconditionally, like :112
        if operate_stride1:
            self.P0_conv1 = conv(sf, oc, ks=1, stride=1, relu=None)
            self.P0_conv2 = conv(oc, oc, ks=3, stride=1, pad=1, relu=None)
        self.P1_conv2 = conv(oc, oc, ks=3, stride=1, pad=1, relu=None)  # never used (:175 is commented out there)
        self.P2_conv2 = conv(oc, oc, ks=3, stride=1, pad=1, relu=None)
        self.P3_conv2 = conv(oc, oc, ks=3, stride=1, pad=1, relu=None)
        self.P4_conv2 = conv(oc, oc, ks=3, stride=1, pad=1, relu=None)
        self.P5_conv2 = conv(oc, oc, ks=3, stride=1, pad=1, relu=None)
        if self.sixth_pooling:
            self.P6_conv1 = conv(sf * 64, oc, ks=1, stride=1, relu=None)
            self.P6_conv2 = conv(oc, oc, ks=3, stride=1, pad=1, relu=None)

    @staticmethod
    def _stage(seq, x):
        """One of C3..C6 on the previous stage's output x.  Returns (stage output, the tensor the lateral of x's level must read): with
        the stride tap the first block's two strided 1x1 layers share ONE sub-sampled copy of x and the lateral reads the tap's alias,
        so that the three gradients of x meet in one place (fused_epilogue._StrideTap)."""
        blk = seq[0]
        if isinstance(blk, ResBlock) and isinstance(blk.conv3, ConvBias) and isinstance(blk.relu, nn.ReLU) and fused_epilogue.stride_tap_applies(blk, x):
            x_lat, x_s = fused_epilogue.stride_tap(x, blk.conv1[0].stride)
            out = blk.forward_subsampled(x_s)
            for b in list(seq)[1:]:
                out = b(out)
            return out, x_lat
        return seq(x), x

    def forward(self, x):
        if self.operate_stride1:
            # a ONE-channel image is "contiguous" in both layouts, so MIOpen returns C0[0]'s 18-channel output row-major even in a
            # channels-last net -- and C0[1] (18 -> 18 on the full-resolution volume, the most expensive layer of the Retina U-Net step)
            # then misses this repo's few-channel MFMA kernel (channels-last only) and runs on MIOpen's row-major path at 7 TF/s
            # (39.6 ms forward, 57.9 + 45.8 ms backward at 8 x 128^3; profiles/r04/r04_step_launch_by_launch_retina_unet.txt)
            h = fused_epilogue.conv_c0_bias_relu(self.C0[0], x) if fused_epilogue.conv_c0_applies(self.C0[0], x) else self.C0[0](x)
            w1 = self.C0[1][0].weight if isinstance(self.C0[1], nn.Sequential) else self.C0[1].weight
            mf = torch.channels_last_3d if h.dim() == 5 else torch.channels_last
            if w1.is_contiguous(memory_format=mf) and not w1.is_contiguous() and not h.is_contiguous(memory_format=mf):
                h = h.contiguous(memory_format=mf)
            c0_out = self.C0[1](h)
        else:
            c0_out = x
        if not self.operate_stride1 and fused_epilogue.stem_pool_fused_applies(self.C1, self.C2[0], c0_out):
            # the stem output has ONE consumer here (the pooling in front of C2): one autograd node for stem + bias + ReLU + pooling, whose backward
            # applies the ReLU mask / bias gradient at the pooled resolution (utils/fused_epilogue._ConvStemBiasReLUPool)
            c1_out = None
            c2_out = fused_epilogue.conv_stem_bias_relu_pool(self.C1, c0_out)
            for blk in list(self.C2)[1:]:
                c2_out = blk(c2_out)
        else:
            c1_out = self.C1(c0_out)
            c2_out = self.C2(c1_out)
        c3_out, c2_out = self._stage(self.C3, c2_out)
        c4_out, c3_out = self._stage(self.C4, c3_out)
        c5_out, c4_out = self._stage(self.C5, c4_out)
        if self.sixth_pooling:
            c6_out, c5_out = self._stage(self.C6, c5_out)
            p6_pre_out = self.P6_conv1(c6_out)
            p5_pre_out = _lateral_up(self.P5_conv1, c5_out, p6_pre_out)
        else:
            p5_pre_out = self.P5_conv1(c5_out)
        p4_pre_out = _lateral_up(self.P4_conv1, c4_out, p5_pre_out)
        p3_pre_out = _lateral_up(self.P3_conv1, c3_out, p4_pre_out)
        p2_pre_out = _lateral_up(self.P2_conv1, c2_out, p3_pre_out)
        out_list = [self.P2_conv2(p2_pre_out), self.P3_conv2(p3_pre_out), self.P4_conv2(p4_pre_out), self.P5_conv2(p5_pre_out)]
        if self.sixth_pooling:
            out_list.append(self.P6_conv2(p6_pre_out))
        if self.operate_stride1:
            p1_pre_out = _lateral(self.P1_conv1, c1_out, self.P2_upsample(p2_pre_out))
            p0_pre_out = _lateral(self.P0_conv1, c0_out, self.P1_upsample(p1_pre_out))
            # defer_p0_conv2 (set per call by the Retina U-Net): the caller composes P0_conv2 with its segmentation layer (fused_epilogue.seg_head_composed)
            # and wants P0_conv2's INPUT in the first slot
            out_list = [p0_pre_out if getattr(self, "defer_p0_conv2", False) else self.P0_conv2(p0_pre_out)] + out_list
        return out_list

"""Drop-in for learning3d/models/pcn.py on MI355X (reference: models/pcn.py:8-153).

Inference path: every Conv1d(k=1) is a matrix-core GEMM launch (f16x2 on the fp16 MFMAs by default, chained through fp16
planes; bf16x3 / fp32 MFMA under L3D_GEMM_ARITH).  Two algebraic fusions remove the reference's largest temporaries:
  * encoder (:117-119) concatenates the per-cloud global feature to every point before conv3; here
    W3 is split and the global half becomes a per-cloud shift:  W3 [h ; g] = W3a h + (W3b g);
  * fine decoder (:84-101) materialises a [B, 16384, 1029] feature (4.3 GB at B=64) of which 1024
    channels are the same per-cloud vector; conv5 is applied to the 5 varying channels (grid 2 +
    coarse point 3) and W5[:,5:] global_feature goes into the per-cloud shift.
The FC decoder (3 Linear layers on [B, emb]) runs on the same conv kernels with the clouds as the points of one row block.
With autograd live, forward is still these kernels (_fused.checkpointed); the backward recomputes layer by layer on the HIP
conv / dgrad / wgrad kernels (_train.py) in the reference's op order."""
import torch

from . import _fused
from .pooling import Pooling


FOLD_FUSED = True       # folding decoder as one kernel (l3d_fold_mlp); False: three 1x1-conv launches
FOLD_FACTORED_TRAIN = True   # autograd live: conv5's global-feature columns as a per-cloud shift (PCN._fine_factored)


class PCN(torch.nn.Module):
    def __init__(self, emb_dims=1024, input_shape="bnc", num_coarse=1024, grid_size=4, detailed_output=False):
        super(PCN, self).__init__()
        if input_shape not in ["bcn", "bnc"]:
            raise ValueError("Allowed shapes are 'bcn' (batch * channels * num_in_points), 'bnc' ")
        self.input_shape = input_shape
        self.emb_dims = emb_dims
        self.num_coarse = num_coarse
        self.detailed_output = detailed_output
        self.grid_size = grid_size
        self.num_fine = self.grid_size ** 2 * self.num_coarse
        self.pooling = Pooling('max')
        self.relu = torch.nn.ReLU()
        # encoder (pcn.py:26-50)
        self.conv1 = torch.nn.Conv1d(3, 128, 1)
        self.conv2 = torch.nn.Conv1d(128, 256, 1)
        self.conv3 = torch.nn.Conv1d(2 * 256, 512, 1)
        self.conv4 = torch.nn.Conv1d(512, self.emb_dims, 1)
        # decoder (pcn.py:56-70)
        self.linear1 = torch.nn.Linear(self.emb_dims, 1024)
        self.linear2 = torch.nn.Linear(1024, 1024)
        self.linear3 = torch.nn.Linear(1024, self.num_coarse * 3)
        if detailed_output:        # folding (pcn.py:72-82)
            self.conv5 = torch.nn.Conv1d(1029, 512, 1)
            self.conv6 = torch.nn.Conv1d(512, 512, 1)
            self.conv7 = torch.nn.Conv1d(512, 3, 1)

    # -- reference-order differentiable path (autograd live: train_pcn.py's loop, or the backward of _fused.checkpointed) ----
    def _layer(self, conv, x, relu):
        """Conv1d(k=1) (+ ReLU) on the HIP conv / dgrad / wgrad kernels (_train.py); torch only off the GPU"""
        from ._train import conv_bn_act, hip_layers_ok
        if hip_layers_ok(x):
            return conv_bn_act(x, conv, None, relu=relu)
        y = conv(x)
        return self.relu(y) if relu else y

    def _fc(self, lin, x, relu):
        from ._train import hip_layers_ok, linear_act
        if hip_layers_ok(x):
            return linear_act(x, lin, relu=relu)
        y = lin(x)
        return self.relu(y) if relu else y

    def _encode_torch(self, x):
        out = self._layer(self.conv2, self._layer(self.conv1, x, True), False)
        g = self.pooling(out).unsqueeze(2).repeat(1, 1, self.num_points)
        out = torch.cat([out, g], dim=1)
        out = self._layer(self.conv4, self._layer(self.conv3, out, True), False)
        return self.pooling(out)

    def _grid_center(self, coarse):
        B = coarse.shape[0]
        linspace = torch.linspace(-0.05, 0.05, steps=self.grid_size, device=coarse.device)
        grid = torch.meshgrid(linspace, linspace, indexing="ij")
        grid = torch.reshape(torch.stack(grid, dim=2), (-1, 2)).unsqueeze(0)         # 1 x g^2 x 2
        grid_feature = grid.repeat([B, self.num_coarse, 1])                           # B x fine x 2
        center = coarse.unsqueeze(2).repeat([1, 1, self.grid_size ** 2, 1]).reshape(-1, self.num_fine, 3)
        return grid_feature, center

    def _fine_factored(self, coarse, gfeat):
        """The folding decoder (pcn.py:84-101) with autograd live, conv5 factored as the fused forward has it: 1024 of its 1029
        input channels are the cloud's global feature repeated over all fine points, so their product is a per-cloud vector
        (one addmm, differentiable) and the per-point GEMM is over 5 channels -- not 553 GFLOP forward, as much again in the
        dgrad and a 1029-column wgrad per step (what made a PCN training step 39 ms).  Same function, fp32 rounding aside."""
        from ._train import _ConvAffineAct
        grid_feature, center = self._grid_center(coarse)
        x5 = torch.cat([grid_feature, center], dim=2).permute(0, 2, 1)                   # [B,5,fine]
        w5 = self.conv5.weight.reshape(512, 1029)
        shift = torch.addmm(self.conv5.bias, gfeat, w5[:, 5:].t())                       # [B,512]: W5[:, 5:] g + b5
        z = _ConvAffineAct.apply(x5, w5[:, :5], None, None, None, None, False, False, False, 0)
        out = self.relu(z + shift.unsqueeze(2))
        out = self._layer(self.conv7, self._layer(self.conv6, out, True), False)
        return out.permute(0, 2, 1) + center

    def _fine_torch(self, coarse, gfeat):
        from ._train import hip_layers_ok
        if FOLD_FACTORED_TRAIN and hip_layers_ok(coarse) and self.conv5.in_channels == 1029 and self.conv5.out_channels == 512:
            return self._fine_factored(coarse, gfeat)
        grid_feature, center = self._grid_center(coarse)
        global_feature = gfeat.unsqueeze(1).repeat([1, self.num_fine, 1])
        feature = torch.cat([grid_feature, center, global_feature], dim=2).permute(0, 2, 1)
        out = self._layer(self.conv7, self._layer(self.conv6, self._layer(self.conv5, feature, True), True), False)
        return out.permute(0, 2, 1) + center

    # -- fused inference path ----------------------------------------------------------------------
    def _w_f16(self, name, w):
        """f16x2 weight image of a conv's (slice of a) weight, rebuilt when the parameter changes"""
        cache = self.__dict__.setdefault("_w_f16_cache", {})
        key = (w.data_ptr(), getattr(self, name).weight._version, str(w.device), tuple(w.shape))
        if cache.get(name, (None,))[0] != key:
            cache[name] = (key, _fused.split_weights_f16(w.float().contiguous()))
        return cache[name][1]

    def _encode_f16(self, x, channel_last):
        """The encoder (pcn.py:110-124) as a chain on the fp16 matrix cores (f16x2 arithmetic, conv_f16.hip): conv1 writes fp16
        planes, conv2 reads them and writes planes AND its per-cloud maximum, conv3 takes the pooled half of W3 as a per-cloud
        shift, conv4's epilogue pools -- no fp32 [B,C,N] activation is written or read anywhere."""
        B = x.shape[0]
        N = x.shape[1] if channel_last else x.shape[2]
        img = _fused.first_layer_f16_planes(x, self.conv1.weight.detach().reshape(128, 3), self.conv1.bias.detach(), True, channel_last)
        w2 = self._w_f16("conv2", self.conv2.weight.detach().reshape(256, 128))
        img, g = _fused.pointwise_conv_f16_pool(img, B, N, w2, 128, 256, None, self.conv2.bias.detach(), relu=False, out_planes=True)
        w3 = self.conv3.weight.detach().reshape(512, 512)
        shift = torch.addmm(self.conv3.bias.detach(), g, w3[:, 256:].t())  # per-cloud [B,512]
        img, _ = _fused.pointwise_conv_f16_pool(img, B, N, self._w_f16("conv3", w3[:, :256]), 256, 512, None, shift, relu=True,
                                                out_planes=True, pool=False)
        w4 = self._w_f16("conv4", self.conv4.weight.detach().reshape(self.emb_dims, 512))
        return _fused.pointwise_conv_f16_pool(img, B, N, w4, 512, self.emb_dims, None, self.conv4.bias.detach(), relu=False)[1]

    def _encode_fused(self, x, channel_last):
        N = x.shape[1] if channel_last else x.shape[2]
        if (_fused.gemm_arith() == "f16x2" and _fused.SPLIT_BF16 and N % 256 == 0 and self.emb_dims % 256 == 0
                and self.pooling.pool_type == 'max'):
            return self._encode_f16(x, channel_last)
        pc = _fused.pointwise_conv
        w, _, b = _fused.fold_conv_bn(self.conv1)
        h = pc(x, w, None, b, relu=True, channel_last=channel_last)
        w, _, b = _fused.fold_conv_bn(self.conv2)
        h = pc(h, w, None, b, relu=False)
        g = self.pooling(h)                                                # [B,256]
        w3 = self.conv3.weight.detach().reshape(512, 512)
        shift = torch.addmm(self.conv3.bias.detach(), g, w3[:, 256:].t())  # per-cloud [B,512]
        h = pc(h, w3[:, :256], None, shift, relu=True)
        w, _, b = _fused.fold_conv_bn(self.conv4)
        if self.pooling.pool_type == 'max':
            return _fused.conv_global_max(h, w, None, b, False)             # conv4 + global max, no [B,1024,N] tensor
        h = pc(h, w, None, b, relu=False)
        return self.pooling(h)

    def _fine_fused(self, coarse, gfeat):
        pc = _fused.pointwise_conv
        grid_feature, center = self._grid_center(coarse)
        x5 = torch.cat([grid_feature, center], dim=2)                      # [B, fine, 5] channel-last
        w5 = self.conv5.weight.detach().reshape(512, 1029)
        shift = torch.addmm(self.conv5.bias.detach(), gfeat, w5[:, 5:].t())
        if _fused.SPLIT_BF16 and FOLD_FUSED:
            # conv5 -> conv6 -> conv7 (+ centre) as one kernel: the two [B,512,fine] activations never exist
            from .._lib import check, f32c, lib, ptr, stream_ptr
            w6 = self.conv6.weight.detach().reshape(512, 512)
            f16 = _fused.gemm_arith() == "f16x2"                     # conv6 as f16x2 (3 fp16 products) or bf16x3 (6 bf16)
            key = (w6.data_ptr(), self.conv6.weight._version, str(w6.device), f16)
            if getattr(self, "_w6_split", (None,))[0] != key:
                w6c = w6.float().contiguous()
                self._w6_split = (key, _fused.split_weights_f16(w6c) if f16 else _fused.split_rows(w6c))
            g_, ce = f32c(x5), f32c(center)
            B, Nf, _ = g_.shape
            out = torch.empty((B, Nf, 3), dtype=torch.float32, device=g_.device)
            fn, name = (lib().l3d_fold_mlp_f16, "l3d_fold_mlp_f16") if f16 else (lib().l3d_fold_mlp, "l3d_fold_mlp")
            check(fn(ptr(g_), 5, ptr(f32c(w5[:, :5])), ptr(f32c(shift)), ptr(self._w6_split[1]),
                     ptr(f32c(self.conv6.bias.detach())), ptr(f32c(self.conv7.weight.detach().reshape(3, 512))),
                     ptr(f32c(self.conv7.bias.detach())), ptr(ce), B, Nf, ptr(out), stream_ptr()), name)
            return out
        h = pc(x5, w5[:, :5], None, shift, relu=True, channel_last=True)
        w, _, b = _fused.fold_conv_bn(self.conv6)
        h = pc(h, w, None, b, relu=True)
        w, _, b = _fused.fold_conv_bn(self.conv7)
        h = pc(h, w, None, b, relu=False)
        return h.permute(0, 2, 1) + center

    def forward(self, input_data):
        """reference: models/pcn.py:127-153.  The fused kernels serve the forward in every grad mode (PCN has no BatchNorm or
        dropout, so also under .train(): examples/train_pcn.py:70-91); a backward recomputes through the per-layer HIP route."""
        outs = _fused.checkpointed(self, self._forward, input_data)
        self.global_feature_v, self.coarse_output = outs[0], outs[1]
        result = {'coarse_output': self.coarse_output}
        if self.detailed_output:
            result['fine_output'] = outs[2]
        return result

    def _forward(self, input_data):
        if self.input_shape == "bnc":
            self.num_points = input_data.shape[1]
            input_data = input_data.permute(0, 2, 1)
        else:
            self.num_points = input_data.shape[2]
        if input_data.shape[1] != 3:
            raise RuntimeError("shape of x must be of [Batch x 3 x NumInPoints]")

        if _fused.can_fuse(self, input_data) and input_data.is_cuda:
            cl = self.input_shape == "bnc"
            x = input_data.permute(0, 2, 1) if cl else input_data

            def run():
                g = self._encode_fused(x, cl)
                lr = _fused.linear_rows                                       # FC decoder (pcn.py:132-137): rows = clouds
                c = lr(lr(lr(g, self.linear1, True), self.linear2, True), self.linear3).contiguous().view(g.shape[0], self.num_coarse, 3)
                return (g, c, self._fine_fused(c, g)) if self.detailed_output else (g, c)
            return _fused.run_guarded(input_data.device, run)
        g = self._encode_torch(input_data)
        c = self._fc(self.linear3, self._fc(self.linear2, self._fc(self.linear1, g, True), True), False)
        c = c.view(g.shape[0], self.num_coarse, 3)
        return (g, c, self._fine_torch(c, g)) if self.detailed_output else (g, c)

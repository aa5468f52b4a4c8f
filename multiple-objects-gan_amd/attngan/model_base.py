"""Parameter holders and the fused nn.Sequential used by model.py / inception.py."""
import torch
import torch.nn as nn

from ..hip import ops


# --------------------------------------------------------------------------- parameter holders
class HipConv2d(nn.Conv2d):
    def forward(self, x, up=False):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding, up)


class HipLinear(nn.Linear):
    def forward(self, x, up=False):
        return ops.linear(x, self.weight, self.bias)


class BNCallCounter:
    """nn.BatchNorm bumps `num_batches_tracked` with one tiny kernel per call (96 per train step here).
    The trainer instead makes every counter a view of one flat int64 tensor, counts the calls of a step on
    the host and adds them with ONE launch per step (same buffer values, hipGraph-capturable)."""

    def __init__(self, modules):
        self.mods = [m for net in modules for m in net.modules()
                     if isinstance(m, _HipBNMixin) and m.num_batches_tracked is not None]
        self.index = {id(m): i for i, m in enumerate(self.mods)}
        dev = self.mods[0].num_batches_tracked.device
        self.flat = torch.zeros(len(self.mods), dtype=torch.long, device=dev)
        for i, m in enumerate(self.mods):
            self.flat[i] = m.num_batches_tracked
            m._buffers["num_batches_tracked"] = self.flat[i]
            m._call_counter = self
        self.calls = [0] * len(self.mods)
        self._cached, self._cached_dev = None, None

    def hit(self, m):
        self.calls[self.index[id(m)]] += 1

    def flush(self):
        if self._cached != self.calls:                  # first step (or a changed call pattern): eager only
            self._cached = list(self.calls)
            self._cached_dev = torch.tensor(self._cached, dtype=torch.long, device=self.flat.device)
        self.flat.add_(self._cached_dev)
        self.calls = [0] * len(self.mods)


class _HipBNMixin:
    _call_counter = None

    def _count_call(self):
        if self.num_batches_tracked is not None:
            if self._call_counter is not None:
                self._call_counter.hit(self)
            else:
                self.num_batches_tracked += 1

    def fused_with_conv(self, x, conv, act=ops.ACT_NONE, slope=0.2, groups=1):
        """conv -> this BatchNorm (training statistics) -> act as ONE deep block (hip/ops.DeepConvBNActFn); the caller has
        checked ops.deep_block_eligible.  groups: BatchNorm calls the batch stands for (see fused)."""
        for _ in range(groups):
            self._count_call()
        ph, pw = conv.padding if isinstance(conv.padding, tuple) else (conv.padding, conv.padding)
        return ops.deep_conv_bn_act(x, conv.weight, self.weight, self.bias, self.running_mean, self.running_var, act, slope,
                                    self.eps, self.momentum, conv.stride[0], ph, pw, groups)

    def fused(self, x, act=ops.ACT_NONE, slope=0.2, residual=None, groups=1):
        """groups > 1: x holds `groups` batches one behind the other, each of which the reference passes through this layer in a
        call of its own (one BatchNorm call per object; D(real) and D(fake) of a discriminator update): own batch
        statistics per group, running statistics and the call counter updated group after group -- one launch where the maps
        are small, else the large-map kernels once per group on the group's slice (hip/ops.BNActGroupedFn)."""
        if self.training and groups > 1:
            assert residual is None
            for _ in range(groups):
                self._count_call()
            return ops.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, act, slope, None, self.eps,
                              self.momentum, groups=groups)
        if self.training:
            self._count_call()
            return ops.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, act, slope,
                              residual, self.eps, self.momentum)
        # eval mode: running statistics folded into a per-channel affine
        scale = (self.weight.detach() / torch.sqrt(self.running_var + self.eps)).contiguous()
        shift = (self.bias.detach() - self.running_mean * scale).contiguous()
        fusable = act in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LRELU)
        y = ops.affine_act(x, scale, shift, act if fusable else ops.ACT_NONE, slope)
        if not fusable:
            y = ops.act(y, act, slope)
        return y if residual is None else ops.add(y, residual)

    def forward(self, x):
        return self.fused(x)


class HipBatchNorm2d(_HipBNMixin, nn.BatchNorm2d):
    pass


class HipBatchNorm1d(_HipBNMixin, nn.BatchNorm1d):
    pass


class GLU(nn.Module):
    def forward(self, x):
        assert x.size(1) % 2 == 0, 'channels dont divide 2!'
        return ops.glu(x)


_ACT_CODE = ((GLU, ops.ACT_GLU), (nn.LeakyReLU, ops.ACT_LRELU), (nn.ReLU, ops.ACT_RELU),
             (nn.Tanh, ops.ACT_TANH), (nn.Sigmoid, ops.ACT_SIGMOID))


def _act_of(m):
    for cls, code in _ACT_CODE:
        if isinstance(m, cls):
            return code, float(getattr(m, "negative_slope", 0.0))
    return None, 0.0


class FusedSeq(nn.Sequential):
    """nn.Sequential whose forward pattern-matches its children into fused launches:
    [Upsample] conv|linear [BN [GLU|LeakyReLU|ReLU]] , a trailing BN may take a residual."""

    @staticmethod
    def _deep_block(mods, i, x, up, residual, groups=1):
        """conv (no bias) -> BatchNorm2d (training) [-> LeakyReLU / ReLU] on a small map with packed weights: one fused deep
        block (model.py:575-613, 616-642); returns (output, index behind the matched children) or None"""
        m, n = mods[i], len(mods)
        if up or not isinstance(m, HipConv2d) or m.bias is not None or i + 1 >= n or m.stride[0] != m.stride[1]:
            return None
        bn = mods[i + 1]
        if not isinstance(bn, HipBatchNorm2d) or not bn.training or not torch.is_grad_enabled():
            return None
        j = i + 2
        code, slope = _act_of(mods[j]) if j < n else (None, 0.0)
        if code in (ops.ACT_LRELU, ops.ACT_RELU):
            j += 1
        elif code is None or code in (ops.ACT_TANH, ops.ACT_SIGMOID):
            code, slope = ops.ACT_NONE, 0.0
        else:
            return None                                   # GLU: the generator's blocks, not a deep block
        if j == n and residual is not None:
            return None
        ph, pw = m.padding if isinstance(m.padding, tuple) else (m.padding, m.padding)
        if not ops.deep_block_eligible(x, m.weight, m.stride[0], ph, pw, code, groups):
            return None                                   # (more than two groups: the grouped BatchNorm kernels)
        return bn.fused_with_conv(x, m, code, slope, groups), j

    def forward(self, x, residual=None, groups=1):
        """groups: see _HipBNMixin.fused (the convolutions / linears see one batch of groups*B samples)"""
        mods = list(self)
        i, n, up = 0, len(mods), False
        while i < n:
            m = mods[i]
            if isinstance(m, nn.Upsample):
                up, i = True, i + 1
                continue
            if isinstance(m, (HipConv2d, HipLinear)):
                assert not (up and isinstance(m, HipLinear)), "nn.Upsample in front of a Linear"
                deep = self._deep_block(mods, i, x, up, residual, groups)
                if deep is not None:
                    x, i = deep
                    continue
                if (isinstance(m, HipConv2d) and not up and groups == 1 and i + 1 < n and isinstance(mods[i + 1], nn.Sigmoid)
                        and ops.logits_head_ok(x, m)):
                    # the logits head: full-map convolution + bias + sigmoid in one launch (model.py:626-627, 640-641)
                    x = ops.logits_head(x, m.weight, m.bias)
                    i += 2
                    continue
                if (isinstance(m, HipConv2d) and not up and m.bias is None and i + 1 < n
                        and _act_of(mods[i + 1])[0] == ops.ACT_LRELU and x.dim() == 4 and x.shape[1] <= 16):
                    # conv -> LeakyReLU with no BatchNorm in between (the first layer of a discriminator, model.py:597-598):
                    # the activation rides in the convolution's epilogue
                    x = ops.conv2d_lrelu(x, m.weight, m.stride[0], m.padding, _act_of(mods[i + 1])[1])
                    i += 2
                    continue
                x = m(x, up=up)
                up, i = False, i + 1
                if i < n and isinstance(mods[i], _HipBNMixin):
                    bn = mods[i]
                    i += 1
                    code, slope = _act_of(mods[i]) if i < n else (None, 0.0)
                    if code in (ops.ACT_GLU, ops.ACT_LRELU, ops.ACT_RELU):
                        i += 1
                    else:
                        code = ops.ACT_NONE
                    x = bn.fused(x, code, slope, residual if i == n else None, groups)
                continue
            code, slope = _act_of(m)
            if code is not None:
                x = ops.act(x, code, slope)
            else:
                assert not up, "nn.Upsample must be followed by a convolution"
                x = m(x)                      # anything else (nested containers included) runs as it is
            i += 1
        return x



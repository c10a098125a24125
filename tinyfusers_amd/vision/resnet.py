"""ResBlock -- mirrors tinyfusers/vision/resnet.py:6-31.  Same attribute tree (weight names), fused execution:
GN+SiLU (2 launches) -> conv3x3 (+bias +emb) -> GN+SiLU -> conv3x3 (+bias +skip), skip = 1x1 conv or identity.
``x`` may be the pair (x, skip) of the UNet's output path: the concat of vision/unet.py:72 is never built."""
from ..ff.group_norm import GroupNorm
from ..ff.linear import Linear
from .. import config
from ..storage.tensor import Branch, Tensor
from .conv2d import Conv2d


class ResBlock:
    def __init__(self, channels, emb_channels, out_channels, init=True):
        self.in_layers = [
            GroupNorm(32, channels, init=init),
            Tensor.silu,
            Conv2d(channels, out_channels, kernel_size=[3, 3], padding=[1, 1], init=init),
        ]
        self.emb_layers = [
            Tensor.silu,
            Linear(emb_channels, out_channels, init=init),
        ]
        self.out_layers = [
            GroupNorm(32, out_channels, init=init),
            Tensor.silu,
            lambda x: x,  # dropout slot (vision/resnet.py:20)
            Conv2d(out_channels, out_channels, kernel_size=[3, 3], padding=[1, 1], init=init),
        ]
        self.skip_connection = Conv2d(channels, out_channels, kernel_size=[1, 1], init=init) if channels != out_channels else lambda x: x
        self.in_layers[2]._fp8_ok = self.out_layers[3]._fp8_ok = True     # config 5: both read GroupNorm + SiLU outputs (ff/fp8.py)

    def __call__(self, x, emb, emb_out=None, out_gn=0, out_norm=None):
        """out_gn = G: the block's output is read next by a GroupNorm(G); its statistics ride on the last conv.
        out_norm = (that GroupNorm, silu): where the last conv runs split-K, its reduce applies the norm as well."""
        br = None
        if config.parallel_branches and isinstance(self.skip_connection, Conv2d):
            br = Branch()                                              # the 1x1 skip projection only needs x: parallel branch
            with br:
                skip = self.skip_connection(x)
        if emb_out is None:
            emb_out = self.emb_layers[1](emb, silu_input=True)         # Linear(SiLU(emb)): (rows, Cout)
        g2 = self.out_layers[0].num_groups
        # GroupNorm -> SiLU -> conv as one launch (gn_in: the conv normalises its input patches in LDS) + bias + emb[:, :, None, None]
        # (+ the statistics for norm2)
        h = self.in_layers[2](x, bias_nc=emb_out, gn=g2, gn_in=(self.in_layers[0], True), out_norm=(self.out_layers[0], True))
        n2 = (self.out_layers[0], True)
        if br is not None:
            br.join()
        elif config.fold_skip_projection and config.dtype != "fp8" and isinstance(self.skip_connection, Conv2d) and self.out_layers[3].weight.shape[0] % 8 == 0 \
                and self.skip_connection.weight.shape[1] % 8 == 0:
            # skip_connection(x) + h in ONE GEMM: the 1x1 projection rides as extra K columns of the last conv
            return self.out_layers[3](h, gn=out_gn, extra=(self.skip_connection, x), gn_in=n2, out_norm=out_norm)
        else:
            skip = self.skip_connection(x)
        assert not isinstance(skip, (tuple, list)), "identity skip needs a single tensor (cin == cout)"
        return self.out_layers[3](h, residual=skip, gn=out_gn, gn_in=n2, out_norm=out_norm)


class ResnetBlock:
    """vision/resnet.py:33-45 (VAE; next-row f1)."""

    def __init__(self, in_channels, out_channels=None, init=True):
        self.norm1 = GroupNorm(32, in_channels, init=init)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=[3, 3], padding=[1, 1], init=init)
        self.norm2 = GroupNorm(32, out_channels, init=init)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=[3, 3], padding=[1, 1], init=init)
        self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=[1, 1], init=init) if in_channels != out_channels else lambda x: x

    def __call__(self, x, out_gn=0, out_norm=None):
        h = self.conv1(x, gn=self.norm2.num_groups, gn_in=(self.norm1, True), out_norm=(self.norm2, True))
        return self.conv2(h, residual=self.nin_shortcut(x), gn=out_gn, gn_in=(self.norm2, True), out_norm=out_norm)

"""Decoder of the AutoencoderKL -- the module tinyfusers/vae/decoder.py:8-34 builds (runs once after the sampling loop).

What is contractual is the attribute tree, because attribute paths are checkpoint keys (storage/state.py):
``conv_in``, ``mid``, ``up.<level>.block.<0..2>``, ``up.<level>.upsample.conv`` (levels 1..3), ``norm_out``, ``conv_out``.
Level l works at width WIDTHS[l]; it is entered from the level above (or from the 512-wide mid block) and, except for level 0,
ends by doubling the resolution.  Execution is this package's own: the nearest-2x upsample is resolved inside the following conv's
gather (never written to HBM) and swish(norm_out(x)) is the fused GroupNorm + SiLU apply."""
from ..ff.group_norm import GroupNorm
from ..vision.conv2d import Conv2d
from ..vision.resnet import ResnetBlock
from .mid import Mid

WIDTHS = (128, 256, 512, 512)          # channels of decoder level 0 (full resolution) .. 3 (latent resolution)
BLOCKS_PER_LEVEL = 3


def _conv3(cin, cout, init):
    return Conv2d(cin, cout, kernel_size=[3, 3], padding=[1, 1], init=init)


def _level(index, init):
    width = WIDTHS[index]
    entry = WIDTHS[index + 1] if index + 1 < len(WIDTHS) else WIDTHS[-1]      # what the level above (or mid) hands down
    level = {"block": [ResnetBlock(entry if b == 0 else width, width, init=init) for b in range(BLOCKS_PER_LEVEL)]}
    if index > 0:
        level["upsample"] = {"conv": _conv3(width, width, init)}
    return level


class Decoder:
    def __init__(self, init=True):
        top = WIDTHS[-1]
        self.conv_in = _conv3(4, top, init)
        self.mid = Mid(top, init=init)
        self.up = [_level(i, init) for i in range(len(WIDTHS))]
        self.norm_out = GroupNorm(32, WIDTHS[0], init=init)
        self.conv_out = _conv3(WIDTHS[0], 3, init)

    def __call__(self, x):
        x = self.mid(self.conv_in(x))
        for level in reversed(self.up):                    # latent resolution first
            for block in level["block"]:
                x = block(x)
            up = level.get("upsample")
            if up is not None:
                x = up["conv"](x, upsample=True)           # decoder.py:29-31: nearest 2x, then the 3x3 conv -- one launch here
        return self.conv_out(self.norm_out(x, silu=True))

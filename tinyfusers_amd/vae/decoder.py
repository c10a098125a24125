"""Decoder -- mirrors tinyfusers/vae/decoder.py:8-34 (runs once after the sampling loop).
Same attribute tree (checkpoint keys); the nearest-2x upsample of :29-30 is folded into the conv gather and
swish(norm_out(x)) is the fused GroupNorm+SiLU kernel."""
from ..ff.group_norm import GroupNorm
from ..vision.conv2d import Conv2d
from ..vision.resnet import ResnetBlock
from .mid import Mid


class Decoder:
    def __init__(self, init=True):
        sz = [(128, 256), (256, 512), (512, 512), (512, 512)]
        self.conv_in = Conv2d(4, 512, kernel_size=[3, 3], padding=[1, 1], init=init)
        self.mid = Mid(512, init=init)
        arr = []
        for i, s in enumerate(sz):
            arr.append({"block": [ResnetBlock(s[1], s[0], init=init), ResnetBlock(s[0], s[0], init=init), ResnetBlock(s[0], s[0], init=init)]})
            if i != 0:
                arr[-1]['upsample'] = {"conv": Conv2d(s[0], s[0], kernel_size=[3, 3], padding=[1, 1], init=init)}
        self.up = arr
        self.norm_out = GroupNorm(32, 128, init=init)
        self.conv_out = Conv2d(128, 3, kernel_size=[3, 3], padding=[1, 1], init=init)

    def __call__(self, x):
        x = self.conv_in(x)
        x = self.mid(x)
        for l in self.up[::-1]:
            for b in l['block']:
                x = b(x)
            if 'upsample' in l:
                x = l['upsample']['conv'](x, upsample=True)
        return self.conv_out(self.norm_out(x, silu=True))

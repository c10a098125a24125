"""AutoencoderKL -- mirrors tinyfusers/vae/vae.py:5-18.  Only the decode side (post_quant_conv + Decoder) is on the
sampler's path (variants/sd.py:48-54); the Encoder is not built (not used by example/sd1.py)."""
from ..vision.conv2d import Conv2d
from .decoder import Decoder


class AutoencoderKL:
    def __init__(self, init=True):
        self.decoder = Decoder(init=init)
        self.post_quant_conv = Conv2d(4, 4, kernel_size=[1, 1], init=init)

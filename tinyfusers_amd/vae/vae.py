"""AutoencoderKL -- mirrors tinyfusers/vae/vae.py:5-18: ``encoder``, ``decoder``, ``quant_conv`` (8 -> 8, 1x1), ``post_quant_conv`` (4 -> 4, 1x1).
The sampler uses the decode side only (variants/sd.py:48-54); ``__call__`` is the reference's round trip image -> latent means -> image."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh
from ..vision.conv2d import Conv2d
from .decoder import Decoder
from .encoder import Encoder


class AutoencoderKL:
    def __init__(self, init=True, init_encoder=None):
        """init_encoder: draw random weights for the encoder side too (default: as ``init``).  The sampler never uses the encoder, so
        StableDiffusion builds it empty (init_encoder=False: the module tree update_state walks, no 34 M parameters drawn and uploaded)."""
        self.encoder = Encoder(init=init if init_encoder is None else bool(init_encoder))
        self.decoder = Decoder(init=init)
        self.quant_conv = Conv2d(8, 8, kernel_size=[1, 1], init=init)
        self.post_quant_conv = Conv2d(4, 4, kernel_size=[1, 1], init=init)

    def encode(self, x):
        """image (n, 3, H, W) -> the latent's means (n, 4, H/8, W/8): vae.py:13-15 (``latent[:, 0:4]  # only the means``)."""
        moments = self.quant_conv(self.encoder(x))
        n, c, h, w = moments.shape
        means = DeviceArray.empty((n, 4, h, w), np.float16, "nhwc")
        hip.tf_memcpy_2d_async(means.ptr, 4 * 2, moments.ptr, c * 2, 4 * 2, n * h * w, _sh())     # channels 0..3 of every NHWC pixel
        return means

    def __call__(self, x):
        """vae.py:12-18."""
        return self.decoder(self.post_quant_conv(self.encode(x)))

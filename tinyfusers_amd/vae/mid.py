"""Mid -- mirrors tinyfusers/vae/mid.py:5-12."""
from ..attention.attention import AttnBlock
from ..storage.tensor import Tensor
from ..vision.resnet import ResnetBlock


class Mid:
    def __init__(self, block_in, init=True):
        self.block_1 = ResnetBlock(block_in, block_in, init=init)
        self.attn_1 = AttnBlock(block_in, init=init)
        self.block_2 = ResnetBlock(block_in, block_in, init=init)

    def __call__(self, x):
        return Tensor.sequential([self.block_1, self.attn_1, self.block_2], x)

"""Per-op CPU restatement of the reference's op surface (fp32, torch-CPU / numpy).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Arrays are torch CPU fp32 tensors in the
reference's logical layouts: NCHW for images, (B, T, C) for tokens, (B, NH, T, HS) for heads.
The dense ops go through the very torch functions the reference's own tests use as ground
truth (tests/conv2d.py:27, tests/layer_norm.py:38, tests/sdpa.py:71, tests/linear.py) because
the reference's arithmetic for them lives in closed third-party libraries (cuDNN frontend,
unpinned git HEAD; cuBLAS through cupy-cuda12x==12.3.0) that are absent from /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "as_t", "conv_2d", "conv2d_bias", "linear", "group_norm", "group_norm_affine", "layer_norm",
    "sigmoid", "silu", "gelu", "quick_gelu", "softmax_rows", "scaled_dot_product_attention",
    "geglu", "upsample_nearest2x", "timestep_embedding", "get_alphas_cumprod",
    "get_x_prev_and_pred_x0", "cfg_combine",
]


def as_t(x):
    if isinstance(x, torch.Tensor):
        return x.to(torch.float32)
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))


def conv_2d(x, w, padding, stride, dilation):
    """vision/conv2d.py:9-28 -- cuDNN conv_fprop, fp32, NCHW cross-correlation, no bias.
    Net effect asserted by tests/conv2d.py:27-33 == torch.nn.functional.conv2d."""
    return F.conv2d(as_t(x), as_t(w), None, tuple(stride), tuple(padding), tuple(dilation))


def conv2d_bias(x, w, b, padding=(0, 0), stride=(1, 1), dilation=(1, 1)):
    """vision/conv2d.py:55-58 -- Conv2d.__call__: conv_2d(x.astype(f32)) + bias[None,:,None,None]."""
    y = conv_2d(x, w, padding, stride, dilation)
    return y if b is None else y + as_t(b).reshape(1, -1, 1, 1)


def linear(x, w, b=None):
    """ff/linear.py:119-120 (the live branch, D3) -- cp.dot(x, W^T) + b."""
    y = as_t(x) @ as_t(w).t()
    return y if b is None else y + as_t(b)


def group_norm(x, num_groups, eps):
    """ff/group_norm.py:3-11 -- reshape (N,G,-1); mean; centre; 1/sqrt(mean(sq)+eps) (biased var)."""
    x = as_t(x)
    n, c, h, w = x.shape
    xg = x.reshape(n, num_groups, -1)
    mean = xg.mean(dim=-1, keepdim=True)
    yn = xg - mean
    den = torch.sqrt((yn * yn).mean(dim=-1, keepdim=True) + eps)
    return (yn * (1.0 / den)).reshape(n, c, h, w)


def group_norm_affine(x, num_groups, weight, bias, eps=1e-5):
    """ff/group_norm.py:18-21 -- GroupNorm.__call__: group_norm(x) * gamma[c] + beta[c]."""
    y = group_norm(x, num_groups, eps)
    return y * as_t(weight).reshape(1, -1, 1, 1) + as_t(bias).reshape(1, -1, 1, 1)


def layer_norm(x, weight, bias, eps=1e-5):
    """ff/layer_norm.py:8-32, :34-49 -- LayerNorm over the last dim, eps 1e-5, affine.
    The reference hands cuDNN NHWC-style strides that are only right for B==1; its own
    test pins torch.nn.functional.layer_norm as the truth (tests/layer_norm.py:38-41), so the
    standard last-dim LayerNorm is the semantics restated here."""
    x = as_t(x)
    if weight is not None and as_t(weight).ndim > 1:
        # tests/layer_norm.py:22-41: a (1, C, H, W) scale normalises over [C, H, W]; :44-71: a (1, 1, 1, W) scale over [W]
        w, b = as_t(weight), as_t(bias)
        shape = tuple(w.shape)
        while len(shape) > 1 and shape[0] == 1:
            shape = shape[1:]
        return F.layer_norm(x, shape, w.reshape(shape), b.reshape(shape), float(eps))
    return F.layer_norm(x, (x.shape[-1],), as_t(weight), as_t(bias), float(eps))


def sigmoid(x):
    """storage/tensor.py:64-66 -- 1/(1+exp(-x))."""
    x = as_t(x)
    return 1.0 / (1.0 + torch.exp(-x))


def silu(x):
    """storage/tensor.py:68-70 (== swish :84-86) -- x * sigmoid(x)."""
    x = as_t(x)
    return x * sigmoid(x)


def quick_gelu(x):
    """storage/tensor.py:76-78 -- x * sigmoid(1.702 x)."""
    x = as_t(x)
    return x * sigmoid(x * 1.702)


def gelu(x):
    """storage/tensor.py:80-82 -- 0.5 x (1 + tanh(0.7978845608 x (1 + 0.044715 x^2)))."""
    x = as_t(x)
    return 0.5 * x * (1.0 + torch.tanh(x * 0.7978845608 * (1.0 + 0.044715 * x * x)))


def softmax_rows(x):
    """native/cuda/softmax.cu:24-112 -- numerically stable row softmax over (N, C)."""
    x = as_t(x)
    m = x.max(dim=-1, keepdim=True).values
    e = torch.exp(x - m)
    return e / e.sum(dim=-1, keepdim=True)


def scaled_dot_product_attention(q, k, v, attn_mask=None):
    """attention/sdpa.py:53-77 -- softmax(scale * q k^T (+mask)) v, scale = 1/sqrt(HS),
    q,k,v (B,NH,T,HS); bool mask: 0 -> -inf, else additive (:67-68)."""
    q, k, v = as_t(q), as_t(k), as_t(v)
    scale = np.float32(1.0 / math.sqrt(q.shape[-1]))
    pre = float(scale) * (q @ k.transpose(-1, -2))
    if attn_mask is not None:
        m = attn_mask if isinstance(attn_mask, torch.Tensor) else torch.from_numpy(np.asarray(attn_mask))
        if m.dtype == torch.bool:
            pre = pre + torch.where(m == 0, torch.tensor(-float("inf")), torch.tensor(0.0))
        else:
            pre = pre + m.to(torch.float32)
    b, nh, tq, tk = pre.shape
    att = softmax_rows(pre.reshape(b * nh * tq, tk)).reshape(b, nh, tq, tk)
    return att @ v


def geglu(x, w, b):
    """ff/nn.py:10-12 -- proj(x) -> split last dim in halves (a, gate) -> a * gelu(gate)."""
    y = linear(x, w, b)
    a, gate = torch.chunk(y, 2, dim=-1)
    return a * gelu(gate)


def upsample_nearest2x(x):
    """vision/unet.py:81-83 -- broadcast (b,c,h,1,w,1)->(b,c,h,2,w,2) then reshape: nearest 2x."""
    x = as_t(x)
    b, c, h, w = x.shape
    return x.reshape(b, c, h, 1, w, 1).expand(b, c, h, 2, w, 2).reshape(b, c, h * 2, w * 2)


def timestep_embedding(timesteps, dim, max_period=10000):
    """vision/unet.py:92-97 -- [cos(t f), sin(t f)], f = exp(-ln(max_period) i/half), fp32, (1, dim)."""
    half = dim // 2
    freqs = np.exp(-np.log(max_period) * np.arange(half, dtype=np.float32) / half)
    args = np.asarray(timesteps, dtype=np.float32) * freqs.astype(np.float32)
    out = np.concatenate((np.cos(args), np.sin(args))).reshape(1, -1).astype(np.float32)
    return torch.from_numpy(out)


def get_alphas_cumprod(beta_start=0.00085, beta_end=0.0120, n_training_steps=1000):
    """variants/sd.py:61-65 -- cumprod(1 - linspace(sqrt(b0), sqrt(b1), n, f32)**2), fp32."""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_training_steps, dtype=np.float32) ** 2
    alphas = 1.0 - betas
    return np.cumprod(alphas, axis=0)


def get_x_prev_and_pred_x0(x, e_t, a_t, a_prev):
    """variants/sd.py:14-25 -- DDIM update with sigma_t = 0."""
    x, e_t = as_t(x), as_t(e_t)
    a_t = as_t(np.asarray(a_t, dtype=np.float32))
    a_prev = as_t(np.asarray(a_prev, dtype=np.float32))
    sigma_t = 0
    sqrt_one_minus_at = torch.sqrt(1 - a_t)
    pred_x0 = (x - sqrt_one_minus_at * e_t) / torch.sqrt(a_t)
    dir_xt = torch.sqrt(1.0 - a_prev - sigma_t ** 2) * e_t
    x_prev = torch.sqrt(a_prev) * pred_x0 + dir_xt
    return x_prev, pred_x0


def cfg_combine(latents, guidance):
    """variants/sd.py:44-45 -- e_t = uncond + g * (cond - uncond); latents = [uncond x B ; cond x B]."""
    latents = as_t(latents)
    b = latents.shape[0] // 2
    un, co = latents[:b], latents[b:]
    return un + float(np.asarray(guidance).reshape(-1)[0]) * (co - un)

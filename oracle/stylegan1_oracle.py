"""Oracle (test infrastructure, CPU fp32) for the StyleGAN1 synthesis network, state-dict driven.

Restates /root/reference/models/stylegan1.py for the inference path of ``G_style`` (`--stylegan1`):
  my_linear        <- MyLinear.forward        :33-37   (equalised lr: weight * w_mul, bias * b_mul)
  conv_layer       <- MyConv2d.forward        :73-103  (>= 128 px: conv_transpose2d with the 4-shift-summed kernel :83-93;
                                                         below: nearest upscale + conv; blur, then bias)
  blur             <- BlurLayer.forward       :162-167
  layer_epilogue   <- LayerEpilogue.forward   :289-318 (NoiseLayer :113-123, LeakyReLU 0.2, InstanceNorm2d, StyleMod :131-136)
  synthesis        <- InputBlock :352-362, GSynthesisBlock :405-410, G_synthesis.forward :491-500
  mapping          <- G_mapping               :192-223
  truncate         <- G_style.forward         :593-596
Pinned by tests/golden/stylegan1.npz (outputs of the imported reference classes, tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F


def my_linear(x, weight, bias, gain, lrmul=1.0, use_wscale=True):
    he_std = gain * weight.shape[1] ** (-0.5)
    w_mul = he_std * lrmul if use_wscale else lrmul
    return F.linear(x, weight * w_mul, None if bias is None else bias * lrmul)


def blur(x):
    k = torch.tensor([1.0, 2.0, 1.0])
    k = (k[:, None] * k[None, :]) / 16.0
    return F.conv2d(x, k[None, None].expand(x.shape[1], -1, -1, -1), padding=1, groups=x.shape[1])


def upscale2d(x):
    b, c, h, w = x.shape
    return x.view(b, c, h, 1, w, 1).expand(-1, -1, -1, 2, -1, 2).contiguous().view(b, c, 2 * h, 2 * w)


def conv_layer(x, weight, bias, gain=math.sqrt(2), upscale=False, with_blur=False):
    """MyConv2d.forward (use_wscale=True, lrmul=1)."""
    k = weight.shape[-1]
    w_mul = gain * (weight.shape[1] * k * k) ** (-0.5)
    done = False
    if upscale and min(x.shape[2:]) * 2 >= 128:
        w = (weight * w_mul).permute(1, 0, 2, 3)
        w = F.pad(w, (1, 1, 1, 1))
        w = w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]
        x = F.conv_transpose2d(x, w, stride=2, padding=(w.size(-1) - 1) // 2)
        done = True
    elif upscale:
        x = upscale2d(x)
    if not done:
        x = F.conv2d(x, weight * w_mul, None, padding=k // 2)
    if with_blur:
        x = blur(x)
    if bias is not None:
        x = x + bias.view(1, -1, 1, 1)
    return x


def layer_epilogue(sd, prefix, x, latent, noise):
    """noise [B or 1, 1, H, W] -> + weight[c] * noise, LeakyReLU(0.2), instance norm (biased var, eps 1e-5), style mod."""
    x = x + sd[f"{prefix}.top_epi.noise.weight"].view(1, -1, 1, 1) * noise
    x = F.leaky_relu(x, 0.2)
    x = F.instance_norm(x, eps=1e-5)
    style = my_linear(latent, sd[f"{prefix}.style_mod.lin.weight"], sd[f"{prefix}.style_mod.lin.bias"], gain=1.0)
    style = style.view(-1, 2, x.shape[1], 1, 1)
    return x * (style[:, 0] + 1.0) + style[:, 1]


def synthesis(sd, dlatents, noise, prefix="g_synthesis"):
    """sd: state dict of G_style (or of a bare G_synthesis with prefix ""); noise: one tensor per block."""
    pre = f"{prefix}." if prefix else ""
    names = []
    for key in sd:
        if key.startswith(f"{pre}blocks.") and key.endswith(".epi1.top_epi.noise.weight"):
            names.append(key[len(f"{pre}blocks."):].split(".")[0])
    b = dlatents.shape[0]
    x = None
    for i, name in enumerate(names):
        p = f"{pre}blocks.{name}"
        if i == 0:
            x = sd[f"{p}.const"].expand(b, -1, -1, -1) + sd[f"{p}.bias"].view(1, -1, 1, 1)
            x = layer_epilogue(sd, f"{p}.epi1", x, dlatents[:, 0], noise[0])
            x = conv_layer(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"])
            x = layer_epilogue(sd, f"{p}.epi2", x, dlatents[:, 1], noise[0])
        else:
            x = conv_layer(x, sd[f"{p}.conv0_up.weight"], sd[f"{p}.conv0_up.bias"], upscale=True, with_blur=True)
            x = layer_epilogue(sd, f"{p}.epi1", x, dlatents[:, 2 * i], noise[i])
            x = conv_layer(x, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"])
            x = layer_epilogue(sd, f"{p}.epi2", x, dlatents[:, 2 * i + 1], noise[i])
    return conv_layer(x, sd[f"{pre}torgb.weight"], sd[f"{pre}torgb.bias"], gain=1.0)


def mapping(sd, z, prefix="g_mapping"):
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(8):
        x = F.leaky_relu(my_linear(x, sd[f"{prefix}.dense{i}.weight"], sd[f"{prefix}.dense{i}.bias"], gain=math.sqrt(2), lrmul=0.01), 0.2)
    return x.unsqueeze(1).expand(-1, 18, -1)


def truncate(styles, truncation_latent, truncation):
    interp = torch.lerp(truncation_latent.expand_as(styles), styles, truncation)
    do_trunc = (torch.arange(styles.size(1)) < 8).view(1, -1, 1)
    return torch.where(do_trunc, interp, styles)

"""Oracle (test infrastructure, CPU fp32) for the StyleGAN2 generator forward, state-dict driven.

Restates /root/reference/models/stylegan2.py for the ``input_is_latent=True`` inference path:
  equal_linear      <- EqualLinear.forward        :123-146
  modulated_conv2d  <- ModulatedConv2d.forward    :217-254  (per-sample weights + grouped conv, as the reference)
  styled_conv       <- StyledConv.forward         :338-343  (+ NoiseInjection :262-266, FusedLeakyReLU op/fused_act.py:74-83)
  to_rgb            <- ToRGB.forward              :356-365  (+ Upsample :34-52)
  generator_forward <- Generator.forward          :526-576  (truncation lerp :541-543, layer walk :547-569)
  mapping_network   <- Generator.style / PixelNorm :15-20,388-393
  frames_to_uint8   <- /root/reference/render.py:40-43

Pinned by tests/golden/gen_*.npz / layers_*.npz (see tests/golden/make_golden.py).  This file is also the
``cpu_baseline`` (kind "port") timed by bench.py on the GPU box's host cores.
"""
import math

import torch
import torch.nn.functional as F

from . import ops_oracle as ops


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    y = F.linear(x, weight * scale)
    if activation:
        return ops.fused_leaky_relu(y, bias * lr_mul)
    return y + (bias * lr_mul)


def pixel_norm(x):
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


def mapping_network(sd, z, n_mlp=8, lr_mlp=0.01):
    """z [N,512] -> w [N,512] (the evident intent of generate_latents)."""
    w = pixel_norm(z)
    for i in range(1, n_mlp + 1):
        w = equal_linear(w, sd[f"style.{i}.weight"], sd[f"style.{i}.bias"], lr_mul=lr_mlp, activation=True)
    return w


def latents_from_z(sd, zs, n_latent, inject_index=None):
    """The ``input_is_latent=False`` branch of Generator.forward (:511-526): every z [N,512] goes through the mapping
    network; one z fills all n_latent rows, two are mixed at ``inject_index`` (rows < index from the first)."""
    ws = [mapping_network(sd, z) for z in zs]
    if len(ws) < 2:
        return ws[0][:, None, :].repeat(1, n_latent, 1)
    return torch.cat([ws[0][:, None, :].repeat(1, inject_index, 1), ws[1][:, None, :].repeat(1, n_latent - inject_index, 1)], 1)


def modulated_conv2d(x, style_vec, weight, mod_weight, mod_bias, demodulate=True, upsample=False, blur_kernel=None):
    """x [B,Cin,H,W], style_vec [B,512], weight [1,Cout,Cin,k,k]."""
    b, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    s = equal_linear(style_vec, mod_weight, mod_bias)  # [B,Cin]
    wgt = (1.0 / math.sqrt(cin * k * k)) * weight * s.reshape(b, 1, cin, 1, 1)
    if demodulate:
        d = torch.rsqrt((wgt * wgt).sum(dim=(2, 3, 4)) + 1e-8)
        wgt = wgt * d.reshape(b, cout, 1, 1, 1)
    if upsample:
        wt = wgt.transpose(1, 2).reshape(b * cin, cout, k, k)
        y = F.conv_transpose2d(x.reshape(1, b * cin, h, w), wt, stride=2, padding=0, groups=b)
        y = y.reshape(b, cout, y.shape[-2], y.shape[-1])
        # Blur pad for factor 2 / 4 taps / k=3: p = (4-2)-(3-1) = 0 -> pad0 = 1, pad1 = 1  (:185-191)
        p = (blur_kernel.shape[0] - 2) - (k - 1)
        return ops.upfirdn2d(y, blur_kernel, pad=((p + 1) // 2 + 1, p // 2 + 1))
    y = F.conv2d(x.reshape(1, b * cin, h, w), wgt.reshape(b * cout, cin, k, k), padding=k // 2, groups=b)
    return y.reshape(b, cout, y.shape[-2], y.shape[-1])


def styled_conv(sd, prefix, x, style_vec, noise, upsample):
    y = modulated_conv2d(
        x,
        style_vec,
        sd[f"{prefix}.conv.weight"],
        sd[f"{prefix}.conv.modulation.weight"],
        sd[f"{prefix}.conv.modulation.bias"],
        demodulate=True,
        upsample=upsample,
        blur_kernel=sd.get(f"{prefix}.conv.blur.kernel"),
    )
    y = y + sd[f"{prefix}.noise.weight"] * noise
    return ops.fused_leaky_relu(y, sd[f"{prefix}.activate.bias"])


def to_rgb(sd, prefix, x, style_vec, skip=None):
    y = modulated_conv2d(
        x,
        style_vec,
        sd[f"{prefix}.conv.weight"],
        sd[f"{prefix}.conv.modulation.weight"],
        sd[f"{prefix}.conv.modulation.bias"],
        demodulate=False,
    )
    y = y + sd[f"{prefix}.bias"]
    if skip is not None:
        kern = sd[f"{prefix}.upsample.kernel"]
        p = kern.shape[0] - 2
        y = y + ops.upfirdn2d(skip, kern, up=2, pad=((p + 1) // 2 + 1, p // 2))
    return y


def generator_forward(sd, latents, noise=None, truncation=None, truncation_latent=None, bends=None, return_activations=False,
                      min_rgb_size=4):
    """latents [B,n_latent,512] (or [B,512]); noise: list of [B,1,r,r] / None (None -> checkpoint buffer
    ``noises.noise_i``, i.e. randomize_noise=False, :531-535); truncation: None | float | [B] tensor.
    ``bends``: optional {layer_id: callable} applied where ManipulationLayer sits (:297-307, ids :417-449)."""
    size = sd[[k for k in sd if k.startswith("noises.noise_")][-1]].shape[-1]
    log_size = int(math.log2(size))
    n_latent = log_size * 2 - 2
    num_layers = (log_size - 2) * 2 + 1
    if latents.dim() == 2:
        latents = latents[:, None, :].repeat(1, n_latent, 1)
    b = latents.shape[0]
    noise = list(noise) if noise is not None else [None] * num_layers
    for i in range(num_layers):
        if noise[i] is None:
            noise[i] = sd[f"noises.noise_{i}"]
    if truncation is not None:
        t = torch.as_tensor(truncation, dtype=torch.float32).reshape(-1)
        tl = truncation_latent if truncation_latent is not None else torch.zeros(1, latents.shape[-1])
        latents = tl[None] + t[:, None, None] * (latents - tl[None])
    bends = bends or {}

    def bend(layer_id, t):
        return bends[layer_id](t) if layer_id in bends else t

    acts = []
    if "input.linear.weight" in sd:  # LatentInput (:281-294): activate(fused_lrelu(EqualLinear(latent[:, 0]))) -> [B,C,4,4]
        w_in = sd["input.linear.weight"]
        first = F.linear(latents[:, 0], w_in * (1 / math.sqrt(w_in.shape[1])))
        first = ops.fused_leaky_relu(ops.fused_leaky_relu(first, sd["input.linear.bias"]), sd["input.activate.bias"])
        out = bend(0, first.reshape(b, -1, 4, 4))
    else:
        out = bend(0, sd["input.input"].repeat(b, 1, 1, 1))
    out = bend(1, styled_conv(sd, "conv1", out, latents[:, 0], noise[0], False))
    acts.append(out)
    current = 4
    image = to_rgb(sd, "to_rgb1", out, latents[:, 1]) if min_rgb_size <= current else None  # (:553-556)
    i = 1
    for n in range(log_size - 2):
        current *= 2
        out = bend(2 * n + 2, styled_conv(sd, f"convs.{2 * n}", out, latents[:, i], noise[2 * n + 1], True))
        acts.append(out)
        out = bend(2 * n + 3, styled_conv(sd, f"convs.{2 * n + 1}", out, latents[:, i + 1], noise[2 * n + 2], False))
        acts.append(out)
        if min_rgb_size <= current:  # (:567-568)
            image = to_rgb(sd, f"to_rgbs.{n}", out, latents[:, i + 2], image)
        i += 2
    if return_activations:
        return image, acts
    return image


def frames_to_uint8(images):
    """[B,3,H,W] float -> [B,H,W,3] uint8: clamp(-1,1), (x+1)*127.5, truncating cast (render.py:40-43)."""
    x = (images.clamp(-1, 1) + 1) * 127.5
    return x.permute(0, 2, 3, 1).contiguous().numpy().astype("uint8")

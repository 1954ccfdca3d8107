"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (wetts_amd/).

A CPU float32 restatement of the reference's SynthesizerTrn.infer() hot path (wenet-e2e/wetts,
wetts/vits/model/models.py:228-280) as plain functions over a dict of *folded* weights
(name -> tensor, weight-norm already collapsed).  Because the kernels under test are floating
point, the restatement is written with torch CPU tensor ops (the task's "torch fp32 reference for
a floating-point kernel"), but it is a restatement, not the reference's nn.Modules: the
relative-position attention is the direct banded form, generate_path is an index search, Flip is
index arithmetic, and ConvTranspose1d is not special-cased anywhere else.

Pinning: tests/test_oracle_golden.py checks every function here against golden vectors produced
by the *real* reference run in the build container (tests/golden/make_golden.py, fixtures under
tests/golden/*.npz) and, when /root/reference is present, against the live reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Each function cites the reference file:line it follows (paths relative to wetts/vits/).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # model/modules.py:7


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def sequence_mask(length, max_length=None):
    """utils/commons.py:113-117."""
    if max_length is None:
        max_length = int(length.max())
    ar = torch.arange(max_length, dtype=length.dtype)
    return ar.unsqueeze(0) < length.unsqueeze(1)


def layer_norm_c(x, gamma, beta, eps=1e-5):
    """model/normalization.py:16-19: LayerNorm over the channel dim of [B,C,T]."""
    xt = x.transpose(1, -1)
    xt = F.layer_norm(xt, (x.shape[1],), gamma, beta, eps)
    return xt.transpose(1, -1)


def conv1d(W, name, x, dilation=1, padding=0, groups=1):
    return F.conv1d(x, W[name + ".weight"], W.get(name + ".bias"), dilation=dilation,
                    padding=padding, groups=groups)


# ------------------------------------------------------------------------------------------------
# text encoder  (model/encoders.py:47-57, model/attentions.py:70-87,225-282,403-411)
# ------------------------------------------------------------------------------------------------
def rel_attention(W, pre, x, attn_mask, n_heads, window):
    """MultiHeadAttention.forward/attention (attentions.py:225-282) in banded form:
    scores[i,j] = (q_i/sqrt(dk)).k_j + [|j-i|<=w] (q_i/sqrt(dk)).E_k[j-i+w];
    out_i = sum_j P[i,j] v_j + sum_{|r|<=w} P[i,i+r] E_v[r+w]   (the pad/reshape skew of
    attentions.py:284-358 computes exactly these terms)."""
    q = conv1d(W, pre + ".conv_q", x)
    k = conv1d(W, pre + ".conv_k", x)
    v = conv1d(W, pre + ".conv_v", x)
    b, d, t = q.shape
    dk = d // n_heads
    q = q.view(b, n_heads, dk, t).transpose(2, 3)  # [b,h,t,dk]
    k = k.view(b, n_heads, dk, t).transpose(2, 3)
    v = v.view(b, n_heads, dk, t).transpose(2, 3)
    qs = q / math.sqrt(dk)
    scores = torch.matmul(qs, k.transpose(-2, -1))  # [b,h,t,t]
    idx = torch.arange(t)
    if window is not None:  # window_size=None (the VITS2 flow encoders): plain attention
        Ek = W[pre + ".emb_rel_k"][0]  # [2w+1, dk] shared across heads
        Ev = W[pre + ".emb_rel_v"][0]
        rel = torch.matmul(qs, Ek.t())  # [b,h,t,2w+1]
        for r in range(-window, window + 1):
            i = idx[(idx + r >= 0) & (idx + r < t)]
            if i.numel():
                scores[:, :, i, i + r] += rel[:, :, i, r + window]
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, v)  # [b,h,t,dk]
    if window is not None:
        for r in range(-window, window + 1):
            i = idx[(idx + r >= 0) & (idx + r < t)]
            if i.numel():
                out[:, :, i, :] += p[:, :, i, i + r].unsqueeze(-1) * Ev[r + window]
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return conv1d(W, pre + ".conv_o", out)


def ffn(W, pre, x, x_mask, k):
    """FFN.forward with _same_padding (attentions.py:403-429)."""
    pl, pr = (k - 1) // 2, k // 2
    h = conv1d(W, pre + ".conv_1", F.pad(x * x_mask, (pl, pr)))
    h = torch.relu(h)
    h = conv1d(W, pre + ".conv_2", F.pad(h * x_mask, (pl, pr)))
    return h * x_mask


def encoder_stack(W, pre, x, x_mask, n_layers, n_heads, window, k, g=None):
    """attentions.Encoder.forward (attentions.py:70-87); g [B,gin,1] enables the speaker branch
    at layer cond_layer_idx = 2 (:44-48,74-78)."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for l in range(n_layers):
        if g is not None and l == 2:
            gl = F.linear(g.transpose(1, 2), W[pre + ".spk_emb_linear.weight"],
                          W[pre + ".spk_emb_linear.bias"]).transpose(1, 2)
            x = (x + gl) * x_mask
        y = rel_attention(W, f"{pre}.attn_layers.{l}", x, attn_mask, n_heads, window)
        x = layer_norm_c(x + y, W[f"{pre}.norm_layers_1.{l}.gamma"],
                         W[f"{pre}.norm_layers_1.{l}.beta"])
        y = ffn(W, f"{pre}.ffn_layers.{l}", x, x_mask, k)
        x = layer_norm_c(x + y, W[f"{pre}.norm_layers_2.{l}.gamma"],
                         W[f"{pre}.norm_layers_2.{l}.beta"])
    return x * x_mask


def text_encoder(W, cfg, x_ids, x_lengths, g=None):
    """TextEncoder.forward (encoders.py:47-57) + Encoder.forward (attentions.py:70-87); g is only
    used by speaker-conditioned encoders (models.py:87-101)."""
    H = cfg["hidden_channels"]
    x = F.embedding(x_ids, W["enc_p.emb.weight"]) * math.sqrt(H)
    x = x.transpose(1, -1)
    x_mask = sequence_mask(x_lengths, x.shape[2]).unsqueeze(1).to(x.dtype)
    x = x * x_mask
    x = encoder_stack(W, "enc_p.encoder", x, x_mask, cfg["n_layers"], cfg["n_heads"],
                      cfg["window_size"], cfg["kernel_size"],
                      g if cfg.get("use_spk_conditioned_encoder", 0) else None)
    stats = conv1d(W, "enc_p.proj", x) * x_mask
    m, logs = torch.split(stats, cfg["inter_channels"], dim=1)
    return x, m, logs, x_mask


# ------------------------------------------------------------------------------------------------
# duration predictors
# ------------------------------------------------------------------------------------------------
def dds_conv(W, pre, x, x_mask, g=None, n_layers=3, k=3):
    """DDSConv.forward (duration_predictors.py:45-57)."""
    if g is not None:
        x = x + g
    C = x.shape[1]
    for i in range(n_layers):
        dil = k ** i
        pad = (k * dil - dil) // 2
        y = conv1d(W, f"{pre}.convs_sep.{i}", x * x_mask, dilation=dil, padding=pad, groups=C)
        y = layer_norm_c(y, W[f"{pre}.norms_1.{i}.gamma"], W[f"{pre}.norms_1.{i}.beta"])
        y = F.gelu(y)
        y = conv1d(W, f"{pre}.convs_1x1.{i}", y)
        y = layer_norm_c(y, W[f"{pre}.norms_2.{i}.gamma"], W[f"{pre}.norms_2.{i}.beta"])
        y = F.gelu(y)
        x = x + y
    return x * x_mask


def rq_spline_inverse(x, uw, uh, ud, tail_bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """unconstrained_rational_quadratic_spline(inverse=True, tails='linear') +
    rational_quadratic_spline inverse branch (utils/transforms.py:47-97,100-187), written
    per-element over all positions (inside-interval positions take the spline, the rest are the
    identity) instead of boolean-mask scatter."""
    nb = uw.shape[-1]
    inside = (x >= -tail_bound) & (x <= tail_bound)
    cst = float(np.log(np.exp(1 - min_d) - 1))
    ud = F.pad(ud, (1, 1))
    ud[..., 0] = cst
    ud[..., -1] = cst

    def cum(un, mn):
        p = F.softmax(un, dim=-1)
        p = mn + (1 - mn * nb) * p
        c = torch.cumsum(p, dim=-1)
        c = F.pad(c, (1, 0), value=0.0)
        c = (2 * tail_bound) * c + (-tail_bound)
        c[..., 0] = -tail_bound
        c[..., -1] = tail_bound
        return c, c[..., 1:] - c[..., :-1]

    cw, widths = cum(uw, min_w)
    ch, heights = cum(uh, min_h)
    deriv = min_d + F.softplus(ud)
    # searchsorted (transforms.py:42-44): last edge += eps, count(x >= edge) - 1
    edges = ch.clone()
    edges[..., -1] += 1e-6
    xc = torch.where(inside, x, torch.zeros_like(x))
    b = (torch.sum(xc[..., None] >= edges, dim=-1) - 1).clamp(0, nb - 1)[..., None]
    in_cw = cw.gather(-1, b)[..., 0]
    in_bw = widths.gather(-1, b)[..., 0]
    in_ch = ch.gather(-1, b)[..., 0]
    delta = heights / widths
    in_delta = delta.gather(-1, b)[..., 0]
    d0 = deriv.gather(-1, b)[..., 0]
    d1 = deriv[..., 1:].gather(-1, b)[..., 0]
    in_h = heights.gather(-1, b)[..., 0]
    a_ = (xc - in_ch) * (d0 + d1 - 2 * in_delta) + in_h * (in_delta - d0)
    b_ = in_h * d0 - (xc - in_ch) * (d0 + d1 - 2 * in_delta)
    c_ = -in_delta * (xc - in_ch)
    disc = b_.pow(2) - 4 * a_ * c_
    assert bool((disc[inside] >= 0).all()), "transforms.py:171"
    root = (2 * c_) / (-b_ - torch.sqrt(disc))
    y = root * in_bw + in_cw
    return torch.where(inside, y, x)


def conv_flow_reverse(W, pre, z, x_mask, g, filter_channels, num_bins=10, tail_bound=5.0):
    """ConvFlow.forward(reverse=True) (duration_predictors.py:90-122)."""
    x0, x1 = z[:, :1], z[:, 1:]
    h = conv1d(W, pre + ".pre", x0)
    h = dds_conv(W, pre + ".convs", h, x_mask, g=g)
    h = conv1d(W, pre + ".proj", h) * x_mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    uw = h[..., :num_bins] / math.sqrt(filter_channels)
    uh = h[..., num_bins:2 * num_bins] / math.sqrt(filter_channels)
    ud = h[..., 2 * num_bins:]
    x1 = rq_spline_inverse(x1, uw, uh, ud, tail_bound=tail_bound)
    return torch.cat([x0, x1], 1) * x_mask


def sdp_reverse(W, cfg, x, x_mask, g, eps_w, noise_scale_w):
    """StochasticDurationPredictor.forward(reverse=True) (duration_predictors.py:213-219,254-263).
    eps_w replaces the torch.randn draw at :257."""
    H = cfg["hidden_channels"]
    x = conv1d(W, "dp.pre", x)
    if g is not None:
        x = x + conv1d(W, "dp.cond", g)
    x = dds_conv(W, "dp.convs", x, x_mask)
    x = conv1d(W, "dp.proj", x) * x_mask
    z = eps_w * noise_scale_w
    # reversed(flows) minus the "useless vflow": [Flip, CF_n, ..., Flip, CF_2, Flip, EA]
    for f in range(cfg["sdp_n_flows"] - 1, 0, -1):
        z = torch.flip(z, [1])
        z = conv_flow_reverse(W, f"dp.flows.{2 * f + 1}", z, x_mask, x, H)
    z = torch.flip(z, [1])
    z = (z - W["dp.flows.0.m"]) * torch.exp(-W["dp.flows.0.logs"]) * x_mask  # :139-141
    return z[:, :1]


def dp_forward(W, cfg, x, x_mask, g):
    """DurationPredictor.forward (duration_predictors.py:297-311)."""
    if g is not None:
        x = x + conv1d(W, "dp.cond", g)
    x = conv1d(W, "dp.conv_1", x * x_mask, padding=1)
    x = torch.relu(x)
    x = layer_norm_c(x, W["dp.norm_1.gamma"], W["dp.norm_1.beta"])
    x = conv1d(W, "dp.conv_2", x * x_mask, padding=1)
    x = torch.relu(x)
    x = layer_norm_c(x, W["dp.norm_2.gamma"], W["dp.norm_2.beta"])
    x = conv1d(W, "dp.proj", x * x_mask)
    return x * x_mask


# ------------------------------------------------------------------------------------------------
# length regulation  (models.py:254-267, commons.py:120-136)
# ------------------------------------------------------------------------------------------------
def durations_to_lengths(logw, x_mask, length_scale):
    w = torch.exp(logw) * x_mask * length_scale
    w_ceil = torch.ceil(w)
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    return w_ceil, y_lengths


def generate_path(w_ceil, x_mask, y_mask):
    """commons.generate_path (commons.py:120-136) as an index search: frame ty belongs to the first
    phoneme whose cumulative duration exceeds ty.  Returns attn [B,1,Ty,Tx] and the index map."""
    b, _, tx = w_ceil.shape
    ty = y_mask.shape[-1]
    cum = torch.cumsum(w_ceil[:, 0], -1)  # [b,tx]
    frames = torch.arange(ty, dtype=cum.dtype).view(1, ty, 1)
    below = (cum.unsqueeze(1) > frames)  # [b,ty,tx]
    first = below.to(torch.int64).argmax(-1)
    has = below.any(-1)
    f2p = torch.where(has & (y_mask[:, 0] > 0), first, torch.full_like(first, -1))
    attn = torch.zeros(b, ty, tx)
    bi, ti = torch.nonzero(f2p >= 0, as_tuple=True)
    attn[bi, ti, f2p[bi, ti]] = 1.0
    attn = attn * x_mask[:, 0].unsqueeze(1) * y_mask[:, 0].unsqueeze(2)
    return attn.unsqueeze(1), f2p


# ------------------------------------------------------------------------------------------------
# flow^-1  (model/flows.py:442-449,494-513; model/modules.py:60-106; commons.py:98-105)
# ------------------------------------------------------------------------------------------------
def wn(W, pre, x, x_mask, g, hidden, n_layers, k):
    """WN.forward (modules.py:60-87) with dilation_rate 1."""
    out = torch.zeros_like(x)
    gc = conv1d(W, pre + ".cond_layer", g) if g is not None else None
    for i in range(n_layers):
        x_in = conv1d(W, f"{pre}.in_layers.{i}", x, padding=(k - 1) // 2)
        if gc is not None:
            x_in = x_in + gc[:, i * 2 * hidden:(i + 1) * 2 * hidden]
        acts = torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:])
        rs = conv1d(W, f"{pre}.res_skip_layers.{i}", acts)
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * x_mask


def wn_16bit_sim(W, pre, x, x_mask, g, hidden, n_layers, k, dtype):
    """Numerics spec of the 16-bit WaveNet mode (wetts_set_flow_precision): the graph of `wn` with
    the in_layers / res_skip weights and every activation between the element-wise steps rounded to
    the 16-bit type, f32 accumulation, f32 bias / conditioning adds before the one rounding per
    stored value, the skip sum kept in f32."""
    q = lambda t: t.to(dtype).to(torch.float32)
    out = torch.zeros_like(x)
    gc = conv1d(W, pre + ".cond_layer", g) if g is not None else None
    x = q(x)
    for i in range(n_layers):
        x_in = F.conv1d(x, q(W[f"{pre}.in_layers.{i}.weight"]), W[f"{pre}.in_layers.{i}.bias"],
                        padding=(k - 1) // 2)
        if gc is not None:
            x_in = x_in + gc[:, i * 2 * hidden:(i + 1) * 2 * hidden]
        x_in = q(x_in)
        acts = q(torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:]))
        rs = q(F.conv1d(acts, q(W[f"{pre}.res_skip_layers.{i}.weight"]),
                        W[f"{pre}.res_skip_layers.{i}.bias"]))
        if i < n_layers - 1:
            x = q((x + rs[:, :hidden]) * x_mask)
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * x_mask


def mono_flow_reverse(W, pre, x, y_mask, residual_connection):
    """MonoTransformerFlowLayer.forward(reverse=True) with mean_only=True (flows.py:242-324): a coupling on
    the natural channel halves whose mean comes from a 2-layer, 2-head, window-less Encoder on x0 and a 1x1
    `post`.  residual_connection=True (type "mono_layer_post_residual", :287-300): x0 is halved first and
    kept halved, x1 <- (x1 - m) / (1 + exp(-logs)) with logs = 0; False (":302-324"): the Encoder output gets
    the x0 residual and x1 <- (x1 - m) * exp(-0)."""
    half = x.shape[1] // 2
    x0, x1 = x[:, :half], x[:, half:]
    if residual_connection:
        x0 = x0 / 2
        h = encoder_stack(W, pre + ".pre_transformer", x0, y_mask, 2, 2, None, 3)  # masks its input itself
        m = conv1d(W, pre + ".post", h) * y_mask
        x1 = ((x1 - m) / (1 + torch.exp(-torch.zeros_like(m)))) * y_mask
    else:
        h = encoder_stack(W, pre + ".pre_transformer", x0 * y_mask, y_mask, 2, 2, None, 3) + x0
        m = conv1d(W, pre + ".post", h) * y_mask
        x1 = (x1 - m) * y_mask
    return torch.cat([x0, x1], 1)


def flow_reverse(W, cfg, z_p, y_mask, g, wn_dtype=None):
    """ResidualCouplingTransformersBlock.forward(reverse=True): reversed [(RCL, Flip) x n], or
    [(RCL, Flip, Mono) x n] for the mono_layer types (flows.py:442-449); RCL reverse with mean_only (flows.py:494-513).  wn_dtype (torch.bfloat16 /
    torch.float16): the 16-bit WaveNet numerics spec instead of the f32 graph."""
    H, I = cfg["hidden_channels"], cfg["inter_channels"]
    half = I // 2
    x = z_p
    tf = cfg.get("transformer_flows", 0)
    for f in range(cfg["flow_n_flows"] - 1, -1, -1):
        if tf >= 3:
            # "mono_layer_*": flows = [RCL, Flip, MonoTransformerFlowLayer] x n (flows.py:391-425), so the
            # reversed list applies the mono layer of flow f first
            x = mono_flow_reverse(W, f"flow.flows.{3 * f + 2}", x, y_mask, residual_connection=(tf == 4))
        x = torch.flip(x, [1])
        pre = f"flow.flows.{(3 if tf >= 3 else 2) * f}"
        x0, x1 = x[:, :half], x[:, half:]
        if cfg.get("transformer_flows", 0) == 1:
            # "pre_conv" = ResidualCouplingTransformersLayer (flows.py:95-177): a 2-layer,
            # 2-head, window-less Encoder on x0 with a residual, ahead of `pre`
            x0_ = encoder_stack(W, pre + ".pre_transformer", x0 * y_mask, y_mask, 2, 2, None, 3)
            h = conv1d(W, pre + ".pre", x0_ + x0) * y_mask
        elif cfg.get("transformer_flows", 0) == 2:
            # "pre_conv2" = ResidualCouplingTransformersLayer2 (flows.py:16-92): one Encoder layer
            # (2 heads, relative window 4, FFN kernel = the flow's kernel size) on pre(x0)
            h = conv1d(W, pre + ".pre", x0) * y_mask
            h = h + encoder_stack(W, pre + ".pre_transformer", h * y_mask, y_mask, 1, 2, 4,
                                  cfg["flow_kernel_size"])
        else:
            h = conv1d(W, pre + ".pre", x0) * y_mask
        if wn_dtype is None:
            h = wn(W, pre + ".enc", h, y_mask, g, H, cfg["flow_wn_layers"], cfg["flow_kernel_size"])
        else:
            h = wn_16bit_sim(W, pre + ".enc", h, y_mask, g, H, cfg["flow_wn_layers"],
                             cfg["flow_kernel_size"], wn_dtype)
        m = conv1d(W, pre + ".post", h) * y_mask
        x1 = (x1 - m) * y_mask
        x = torch.cat([x0, x1], 1)
    return x


# ------------------------------------------------------------------------------------------------
# HiFi-GAN generator  (model/decoders.py:63-82,157-170,205-214)
# ------------------------------------------------------------------------------------------------
def hifigan(W, cfg, z, g):
    x = conv1d(W, "dec.conv_pre", z, padding=3)
    if g is not None:
        x = x + conv1d(W, "dec.cond", g)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, uk) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, W[f"dec.ups.{i}.weight"], W[f"dec.ups.{i}.bias"], stride=u,
                               padding=(uk - u) // 2)
        xs = None
        for j, (k, dils) in enumerate(zip(cfg["resblock_kernel_sizes"],
                                          cfg["resblock_dilation_sizes"])):
            n = i * nk + j
            r = x
            if cfg["resblock"] == 1:
                for d, dil in enumerate(dils[:3]):
                    t = F.leaky_relu(r, LRELU_SLOPE)
                    t = conv1d(W, f"dec.resblocks.{n}.convs1.{d}", t, dilation=dil,
                               padding=(k * dil - dil) // 2)
                    t = F.leaky_relu(t, LRELU_SLOPE)
                    t = conv1d(W, f"dec.resblocks.{n}.convs2.{d}", t, padding=(k - 1) // 2)
                    r = t + r
            else:
                for d, dil in enumerate(dils[:2]):
                    t = F.leaky_relu(r, LRELU_SLOPE)
                    t = conv1d(W, f"dec.resblocks.{n}.convs.{d}", t, dilation=dil,
                               padding=(k * dil - dil) // 2)
                    r = t + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)  # default slope 0.01 (decoders.py:78)
    x = F.conv1d(x, W["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


# ------------------------------------------------------------------------------------------------
# Vocos generator  (model/decoders.py:221-308: ConvNeXtLayer :221-248, VocosGenerator :251-305)
# ------------------------------------------------------------------------------------------------
def onnx_stft_inverse_basis(n_fft, hop, win_length):
    """OnnxSTFT.__init__'s `inverse_basis` buffer (utils/stft.py:266-290): pinv(scale * [Re; Im] rows 0..n_fft/2 of
    fft(eye(n_fft))) transposed, as a conv_transpose1d weight [n_fft + 2, 1, n_fft], times the periodic hann window
    (scipy get_window('hann', win_length, fftbins=True), zero-centre-padded to n_fft -- librosa's pad_center, the
    identity for win_length == n_fft)."""
    import numpy as np
    scale = n_fft / hop
    fb = np.fft.fft(np.eye(n_fft))
    cutoff = int(n_fft / 2 + 1)
    fb = np.vstack([np.real(fb[:cutoff, :]), np.imag(fb[:cutoff, :])])
    inv = torch.FloatTensor(np.linalg.pinv(scale * fb).T[:, None, :])
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    lpad = (n_fft - win_length) // 2
    w = np.pad(w, (lpad, n_fft - win_length - lpad))
    return inv * torch.from_numpy(w).float()


def onnx_stft_inverse(mag, phase, n_fft, hop, win_length):
    """OnnxSTFT.inverse (utils/stft.py:325-340): conv_transpose1d of [mag cos; mag sin] with the inverse basis at
    stride hop, n_fft/2 samples cut from each end.  No window-envelope division."""
    x = torch.cat([mag * torch.cos(phase), mag * torch.sin(phase)], dim=1)
    o = F.conv_transpose1d(x, onnx_stft_inverse_basis(n_fft, hop, win_length), stride=hop, padding=0)
    o = o[:, :, int(n_fft / 2):]
    return o[:, :, :-int(n_fft / 2)]


def vocos(W, cfg, z, g):
    """VocosGenerator.forward.  The iSTFT is torch.istft with the arguments
    torchaudio.transforms.InverseSpectrogram(n_fft, hop_length, win_length, center=True) passes
    (hann window, not normalized, one-sided, length=None) -- decoders.py:279,304 -- or, for a model built with
    is_onnx=True (cfg["is_onnx"]; every exported graph, export_onnx.py:59), OnnxSTFT.inverse -- decoders.py:300-301."""
    x = F.pad(z, (1, 0), mode="reflect")               # nn.ReflectionPad1d([1, 0])
    x = conv1d(W, "dec.in_conv", x)
    if g is not None:
        x = x + conv1d(W, "dec.cond", g)
    x = layer_norm_c(x, W["dec.norm_pre.gamma"], W["dec.norm_pre.beta"])
    C = x.shape[1]
    for l in range(cfg["vocos_num_layers"]):
        p = f"dec.layers.{l}"
        res = x
        x = F.conv1d(x, W[p + ".dw_conv.weight"], W[p + ".dw_conv.bias"], padding=1, groups=C)
        x = layer_norm_c(x, W[p + ".norm.gamma"], W[p + ".norm.beta"])
        x = conv1d(W, p + ".pw_conv1", x)
        x = F.gelu(x)
        x = conv1d(W, p + ".pw_conv2", x)
        x = res + W[p + ".scale"] * x
    x = layer_norm_c(x, W["dec.norm_post.gamma"], W["dec.norm_post.beta"])
    x = conv1d(W, "dec.out_conv", x)
    mag, phase = x.chunk(2, dim=1)
    mag = mag.exp().clamp_max(max=1e2)
    n_fft, hop, win = cfg["istft_n_fft"], cfg["istft_hop_length"], cfg["istft_win_length"]
    if cfg.get("is_onnx", 0):
        return onnx_stft_inverse(mag, phase, n_fft, hop, win)
    spec = mag * (phase.cos() + 1j * phase.sin())
    o = torch.istft(spec, n_fft, hop, win, window=torch.hann_window(win), center=True,
                    normalized=False, onesided=True, length=None, return_complex=False)
    return o.unsqueeze(1)


def decoder(W, cfg, z, g):
    """self.dec(z, g): the generator the config selects (models.py:102-128)."""
    return vocos(W, cfg, z, g) if cfg.get("vocoder_type", 0) == 1 else hifigan(W, cfg, z, g)


_QDTYPE = torch.bfloat16


def _q(x):
    """round-to-nearest-even to the 16-bit storage type (bfloat16 / float16), kept in float32"""
    return x.to(_QDTYPE).to(torch.float32)


def hifigan_16bit_sim(W, cfg, z, g, dtype=torch.bfloat16):
    global _QDTYPE
    old, _QDTYPE = _QDTYPE, dtype
    try:
        return hifigan_bf16sim(W, cfg, z, g)
    finally:
        _QDTYPE = old


def hifigan_bf16sim(W, cfg, z, g):
    """Numerics spec of the bf16 decoder mode (wetts_set_decoder_precision(m, 1)): same graph as
    `hifigan`, with conv weights and every inter-conv activation rounded to bfloat16, f32
    accumulation, f32 bias / residual / running-sum adds before the single rounding per output.  conv_pre is a conv of
    the mode like the others (round 6): 16-bit input z, 16-bit weights, f32 bias + speaker conditioning, one rounding."""
    x = F.conv1d(_q(z), _q(W["dec.conv_pre.weight"]), W["dec.conv_pre.bias"], padding=3)
    if g is not None:
        x = x + conv1d(W, "dec.cond", g)
    x = _q(x)
    nk = len(cfg["resblock_kernel_sizes"])

    def cw(name):
        return _q(W[name + ".weight"]), W[name + ".bias"]

    for i, (u, uk) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        w, b = cw(f"dec.ups.{i}")
        x = _q(F.conv_transpose1d(_q(F.leaky_relu(x, LRELU_SLOPE)), w, b, stride=u,
                                  padding=(uk - u) // 2))
        xs = None
        for j, (k, dils) in enumerate(zip(cfg["resblock_kernel_sizes"],
                                          cfg["resblock_dilation_sizes"])):
            n = i * nk + j
            r = x
            nd = 3 if cfg["resblock"] == 1 else 2
            for d, dil in enumerate(dils[:nd]):
                last = d == nd - 1
                if cfg["resblock"] == 1:
                    w1, b1 = cw(f"dec.resblocks.{n}.convs1.{d}")
                    t = _q(F.conv1d(_q(F.leaky_relu(r, LRELU_SLOPE)), w1, b1, dilation=dil,
                                    padding=(k * dil - dil) // 2))
                    w2, b2 = cw(f"dec.resblocks.{n}.convs2.{d}")
                    y = F.conv1d(_q(F.leaky_relu(t, LRELU_SLOPE)), w2, b2, padding=(k - 1) // 2) + r
                else:
                    w1, b1 = cw(f"dec.resblocks.{n}.convs.{d}")
                    y = F.conv1d(_q(F.leaky_relu(r, LRELU_SLOPE)), w1, b1, dilation=dil,
                                 padding=(k * dil - dil) // 2) + r
                if last:
                    if xs is not None:
                        y = y + xs
                    if j == nk - 1:
                        y = y / nk
                    xs = _q(y)
                else:
                    r = _q(y)
        x = xs
    # conv_post like every other conv of this mode: 16-bit input (after the leaky-relu) and weights,
    # f32 accumulation; tanh in f32
    x = _q(F.leaky_relu(x))
    x = F.conv1d(x, _q(W["dec.conv_post.weight"]), None, padding=3)
    return torch.tanh(x)


# ------------------------------------------------------------------------------------------------
# uint8 dynamic quantisation of the decoder's Conv1d nodes (wetts/vits/export_onnx.py:149-157:
# onnxruntime.quantization.quantize_dynamic(..., weight_type=QuantType.QUInt8)).  onnxruntime 1.13.1 is
# fetched by the reference's CMake and is NOT in the reference tree, so this is a restatement of the
# published operator definitions -- PARITY UNPINNED:
#   weights      quant_utils.compute_scale_zp / quantize_nparray: per tensor, asymmetric uint8,
#                range widened to include 0, scale = (max - min) / 255, zp = round(0 - min / scale),
#                w_q = clip(round(w / scale + zp), 0, 255)
#   activations  ONNX DynamicQuantizeLinear: the same formulas over the WHOLE input tensor, per call,
#                rounding half to even, saturating
#   contraction  ONNX ConvInteger: int32 sum (x_q - z_x)(w_q - z_w), zero padding = z_x
#   output       Cast(float) * (s_x * s_w) + bias            (the graph quantize_dynamic emits)
# ConvTranspose nodes are left in float32 by dynamic quantisation.
# ------------------------------------------------------------------------------------------------
def _dq_params(t):
    mn = torch.clamp(t.min(), max=0.0).to(torch.float32)
    mx = torch.clamp(t.max(), min=0.0).to(torch.float32)
    scale = (mx - mn) / torch.tensor(255.0, dtype=torch.float32)
    if not (scale > 0):
        scale = torch.tensor(1.0, dtype=torch.float32)
    zp = torch.clamp(torch.round((0.0 - mn) / scale), 0, 255)  # torch.round: half to even
    return scale, zp


def dynamic_quantize_linear(x):
    """ONNX DynamicQuantizeLinear (opset 11, published operator definition): returns
    (y uint8, y_scale f32 scalar, y_zero_point uint8 scalar).  Pinned by the known-answer examples of the
    operator specification (tests/test_oracle_golden.py::test_onnx_spec_*)."""
    scale, zp = _dq_params(x)
    y = torch.clamp(torch.round(x / scale) + zp, 0, 255)
    return y.to(torch.uint8), scale, zp.to(torch.uint8)


def conv_integer(xq, wq, x_zero_point, w_zero_point, dilation=1, padding=0):
    """ONNX ConvInteger (opset 10) on 1-D data: int32 sum of (x_q - z_x)(w_q - z_w); the zero padding of the
    convolution stands for x_q = z_x (contributes 0).  xq [B,Cin,T] uint8, wq [Cout,Cin,k] uint8 -> int32.
    float64 holds the int32 sums exactly."""
    acc = F.conv1d(xq.double() - float(x_zero_point), wq.double() - float(w_zero_point), None,
                   dilation=dilation, padding=padding)
    return acc.to(torch.int32)


def dynamic_quant_conv1d(x, w, b, dilation=1, padding=0):
    """One quantised Conv node of the graph quantize_dynamic emits: DynamicQuantizeLinear(x) -> ConvInteger
    against the statically quantised weights -> Cast(float) * (s_x * s_w) + bias.  Exact integer arithmetic."""
    xq, sx, zx = dynamic_quantize_linear(x)
    wq, sw, zw = dynamic_quantize_linear(w)  # quant_utils.quantize_nparray: the same per-tensor formulas
    acc = conv_integer(xq, wq, zx, zw, dilation, padding)
    y = acc.to(torch.float32) * (sx * sw)
    if b is not None:
        y = y + b.view(1, -1, 1)
    return y


def hifigan_uint8_dynamic(W, cfg, z, g):
    """Generator.forward as the dynamically quantised ONNX graph computes it."""
    def qc(name, x, dilation=1, padding=0):
        return dynamic_quant_conv1d(x, W[name + ".weight"], W.get(name + ".bias"), dilation, padding)
    x = qc("dec.conv_pre", z, padding=3)
    if g is not None:
        x = x + qc("dec.cond", g)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, uk) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, W[f"dec.ups.{i}.weight"], W[f"dec.ups.{i}.bias"], stride=u,
                               padding=(uk - u) // 2)
        xs = None
        for j, (k, dils) in enumerate(zip(cfg["resblock_kernel_sizes"],
                                          cfg["resblock_dilation_sizes"])):
            n = i * nk + j
            r = x
            nd = 3 if cfg["resblock"] == 1 else 2
            for d, dil in enumerate(dils[:nd]):
                if cfg["resblock"] == 1:
                    t = qc(f"dec.resblocks.{n}.convs1.{d}", F.leaky_relu(r, LRELU_SLOPE), dil,
                           (k * dil - dil) // 2)
                    t = qc(f"dec.resblocks.{n}.convs2.{d}", F.leaky_relu(t, LRELU_SLOPE), 1, (k - 1) // 2)
                else:
                    t = qc(f"dec.resblocks.{n}.convs.{d}", F.leaky_relu(r, LRELU_SLOPE), dil,
                           (k * dil - dil) // 2)
                r = t + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = dynamic_quant_conv1d(x, W["dec.conv_post.weight"], None, 1, 3)
    return torch.tanh(x)


# ------------------------------------------------------------------------------------------------
# infer  (model/models.py:228-280)
# ------------------------------------------------------------------------------------------------
def infer(W, cfg, x_ids, x_lengths, sid=None, noise_scale=1.0, length_scale=1.0,
          noise_scale_w=1.0, max_len=None, eps_w=None, eps_z=None, return_stages=False,
          timers=None):
    """eps_w [B,2,Tx] / eps_z [B,inter,Ty] replace the two torch.randn draws
    (duration_predictors.py:257, models.py:267); when None they are drawn here in the reference's
    order from the global CPU generator.  `timers` (a dict) receives the four stage times the
    reference prints (models.py:273-279): enc, dp, flow, dec seconds."""
    import time as _time
    with torch.no_grad():
        g = None
        if cfg["n_speakers"] > 0:
            g = F.embedding(sid, W["emb_g.weight"]).unsqueeze(-1)
        _t1 = _time.perf_counter()
        x, m_p, logs_p, x_mask = text_encoder(W, cfg, x_ids, x_lengths, g)
        _t2 = _time.perf_counter()
        if cfg["use_sdp"]:
            if eps_w is None:
                eps_w = torch.randn(x.size(0), 2, x.size(2))
            logw = sdp_reverse(W, cfg, x, x_mask, g, eps_w, noise_scale_w)
        else:
            logw = dp_forward(W, cfg, x, x_mask, g)
        _t3 = _time.perf_counter()
        w_ceil, y_lengths = durations_to_lengths(logw, x_mask, length_scale)
        y_mask = sequence_mask(y_lengths, None).unsqueeze(1).to(x_mask.dtype)
        attn, f2p = generate_path(w_ceil, x_mask, y_mask)
        m_e = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
        logs_e = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
        if eps_z is None:
            eps_z = torch.randn_like(m_e)
        z_p = m_e + eps_z * torch.exp(logs_e) * noise_scale
        _t4 = _time.perf_counter()
        z = flow_reverse(W, cfg, z_p, y_mask, g)
        _t5 = _time.perf_counter()
        o = decoder(W, cfg, (z * y_mask)[:, :, :max_len], g)
        _t6 = _time.perf_counter()
        if timers is not None:
            timers.update(enc=_t2 - _t1, dp=_t3 - _t2, flow=_t5 - _t4, dec=_t6 - _t5)
    if return_stages:
        return dict(x=x, m_p=m_p, logs_p=logs_p, x_mask=x_mask, logw=logw, w_ceil=w_ceil,
                    y_lengths=y_lengths, y_mask=y_mask, attn=attn, f2p=f2p, m_p_exp=m_e,
                    logs_p_exp=logs_e, z_p=z_p, z=z, o=o, g=g)
    return o, attn, y_mask, (z, z_p, m_e, logs_e)


def audio_to_int16(audio):
    """inference.py:100-110 output scaling (float32 arithmetic, numpy >= 2 scalar rules)."""
    a = np.asarray(audio, dtype=np.float32).copy()
    a *= np.float32(32767.0) / max(np.float32(0.01), np.max(np.abs(a))) * np.float32(0.6)
    return np.clip(a, -32767.0, 32767.0).astype(np.int16)


# ------------------------------------------------------------------------------------------------
# MAS  (utils/monotonic_align.py:22-57), numpy restatement; the C twin is oracle/mas_oracle.c
# ------------------------------------------------------------------------------------------------
def maximum_path_numpy(neg_cent, t_ys, t_xs):
    values = np.array(neg_cent, dtype=np.float32, copy=True)
    paths = np.zeros(values.shape, dtype=np.int32)
    neg = np.float32(-1e9)
    for i in range(values.shape[0]):
        value, path = values[i], paths[i]
        t_y, t_x = int(t_ys[i]), int(t_xs[i])
        index = t_x - 1
        for y in range(t_y):
            lo, hi = max(0, t_x + y - t_y), min(t_x, y + 1)
            if hi <= lo:
                continue
            xs = np.arange(lo, hi)
            if y == 0:
                v_cur = np.full(xs.shape, neg, np.float32)
                v_prev = np.where(xs == 0, np.float32(0.0), neg).astype(np.float32)
            else:
                prev = value[y - 1]
                v_cur = np.where(xs == y, neg, prev[xs]).astype(np.float32)
                v_prev = np.where(xs == 0, neg, prev[np.maximum(xs - 1, 0)]).astype(np.float32)
            value[y, lo:hi] = value[y, lo:hi] + np.maximum(v_prev, v_cur)
        for y in range(t_y - 1, -1, -1):
            path[y, index] = 1
            if index != 0 and (index == y or value[y - 1, index] < value[y - 1, index - 1]):
                index -= 1
    return paths


# ------------------------------------------------------------------------------------------------
# Philox4x32-10 + Box-Muller: CPU restatement of the library's noise kernel (kernels.hip:randn_kernel).
# The reference itself draws with torch.randn (duration_predictors.py:257, models.py:267); only the
# DISTRIBUTION is part of its contract, so this oracle pins (a) the counter RNG to the published
# Random123 known-answer vectors and (b) the kernel to this restatement element by element.
# ------------------------------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    """ctr uint32[..., 4], key uint32[..., 2] -> uint32[..., 4]  (Salmon et al., SC'11)."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    M0, M1, m32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & m32, p1 >> np.uint64(32), p1 & m32
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, axis=-1).astype(np.uint32)


def philox_randn(n, seed, offset):
    """n standard-normal float32 values: group q of four uses counter (offset + q), key = seed."""
    nq = (n + 3) // 4
    q = np.uint64(offset) + np.arange(nq, dtype=np.uint64)
    ctr = np.zeros((nq, 4), dtype=np.uint32)
    ctr[:, 0] = (q & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    key = np.empty((nq, 2), dtype=np.uint32)
    key[:, 0] = np.uint32(seed & 0xFFFFFFFF)
    key[:, 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32_10(ctr, key)
    out = np.empty((nq, 4), dtype=np.float32)
    for h in range(2):
        u1 = ((r[:, 2 * h] >> 8).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
        u2 = (r[:, 2 * h + 1] >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
        ang = np.float32(6.28318530717958647692) * u2
        out[:, 2 * h] = rad * np.cos(ang)
        out[:, 2 * h + 1] = rad * np.sin(ang)
    return out.reshape(-1)[:n]

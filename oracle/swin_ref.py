"""ORACLE — test infrastructure only.

Plain-PyTorch, CPU, functional restatement of the reference's Swin backbone
(part_distillation/modeling/backbone/swin.py) driven by a flat ``sd`` with the
reference's key names (``patch_embed.*``,
``layers.{s}.blocks.{b}.*``, ``layers.{s}.downsample.*``, ``norm{i}.*``), so the
same weights drive this file and the HIP product path.

Pinned against goldens captured from the REAL reference ``SwinTransformer``
(tests/golden/make_golden.py: ``swin_tiny`` = window 4, ``swin_w12`` = window 12
with head_dim 32, the geometry BASELINE configs 3 / 5 run) in
tests/test_oracle.py.  DropPath is the identity here (parity runs disable it,
SURVEY §8d); absolute position embeddings (``ape``) are not restated — no
shipped YAML enables them.

Each function cites the reference lines it follows.
"""
import torch
import torch.nn.functional as F


def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, x.shape[-1:], sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def relative_position_index(ws):
    """swin.py:110-125: index of the (2*ws-1)^2 bias table for every (query, key) pair of a ws x ws window."""
    r = torch.arange(ws)
    yy, xx = torch.meshgrid(r, r, indexing="ij")
    y, x = yy.reshape(-1), xx.reshape(-1)
    dy = y[:, None] - y[None, :] + ws - 1
    dx = x[:, None] - x[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def shift_region_labels(Hp, Wp, ws, shift):
    """swin.py:417-436: the 3 x 3 region labels of the padded map that SW-MSA masks are built from."""
    lab = torch.zeros((Hp, Wp))
    cuts = lambda n: ((0, n - ws), (n - ws, n - shift), (n - shift, n))
    cnt = 0
    for h0, h1 in cuts(Hp):
        for w0, w1 in cuts(Wp):
            lab[h0:h1, w0:w1] = cnt
            cnt += 1
    return lab


def to_windows(x, ws):
    """swin.py:48-59: [B,Hp,Wp,C] -> [B*nW, ws*ws, C] (row-major windows, row-major tokens)."""
    B, Hp, Wp, C = x.shape
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def from_windows(w, ws, B, Hp, Wp):
    """swin.py:62-75."""
    C = w.shape[-1]
    x = w.reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, Hp, Wp, C)


def window_attention(sd, p, xw, heads, ws, mask):
    """WindowAttention.forward, swin.py:135-175.  xw [B_, N, C]; mask [nW, N, N] of 0 / -100 or None."""
    B_, N, C = xw.shape
    hd = C // heads
    qkv = _lin(sd, p + ".qkv", xw).reshape(B_, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)
    table = sd[p + ".relative_position_bias_table"]
    bias = table[relative_position_index(ws).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    s = s + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        s = (s.reshape(B_ // nW, nW, heads, N, N) + mask[None, :, None]).reshape(-1, heads, N, N)
    a = torch.softmax(s, -1)
    return _lin(sd, p + ".proj", (a @ v).transpose(1, 2).reshape(B_, N, C))


def block(sd, p, x, H, W, heads, ws, shift, mask):
    """SwinTransformerBlock.forward, swin.py:239-299 (drop_path = identity)."""
    B, L, C = x.shape
    y = _ln(sd, p + ".norm1", x).reshape(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    if shift > 0:
        y = torch.roll(y, (-shift, -shift), (1, 2))
    a = window_attention(sd, p + ".attn", to_windows(y, ws), heads, ws, mask if shift > 0 else None)
    y = from_windows(a, ws, B, Hp, Wp)
    if shift > 0:
        y = torch.roll(y, (shift, shift), (1, 2))
    y = y[:, :H, :W].reshape(B, H * W, C)
    x = x + y
    h = _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", _ln(sd, p + ".norm2", x))))
    return x + h


def patch_merging(sd, p, x, H, W):
    """PatchMerging.forward, swin.py:315-341."""
    B, L, C = x.shape
    x = x.reshape(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.reshape(B, -1, 4 * C)
    return F.linear(_ln(sd, p + ".norm", x), sd[p + ".reduction.weight"])


def swin_forward(sd, p, x, *, depths, num_heads, window_size, patch_size=4, out_indices=(0, 1, 2, 3)):
    """SwinTransformer.forward, swin.py:655-682 (PatchEmbed :483-499 with patch_norm, BasicLayer.forward :410-457).
    ``p`` is the key prefix ("" or e.g. "backbone").  Returns {"res2".."res5"} NCHW."""
    pre = (p + ".") if p else ""
    _, _, H, W = x.shape
    if W % patch_size:
        x = F.pad(x, (0, patch_size - W % patch_size))
    if H % patch_size:
        x = F.pad(x, (0, 0, 0, patch_size - H % patch_size))
    x = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=patch_size)
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    if (pre + "patch_embed.norm.weight") in sd:
        x = _ln(sd, pre + "patch_embed.norm", x)
    ws, shift = window_size, window_size // 2
    outs = {}
    for s, depth in enumerate(depths):
        Hp = -(-Wh // ws) * ws
        Wp = -(-Ww // ws) * ws
        mw = to_windows(shift_region_labels(Hp, Wp, ws, shift)[None, :, :, None], ws).squeeze(-1)
        mask = mw[:, None, :] - mw[:, :, None]
        mask = torch.where(mask != 0, torch.full_like(mask, -100.0), torch.zeros_like(mask))
        for b in range(depth):
            x = block(sd, f"{pre}layers.{s}.blocks.{b}", x, Wh, Ww, num_heads[s], ws, 0 if b % 2 == 0 else shift, mask)
        if s in out_indices:
            C = x.shape[-1]
            outs[f"res{s + 2}"] = _ln(sd, f"{pre}norm{s}", x).reshape(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous()
        if (f"{pre}layers.{s}.downsample.reduction.weight") in sd:
            x = patch_merging(sd, f"{pre}layers.{s}.downsample", x, Wh, Ww)
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
    return outs

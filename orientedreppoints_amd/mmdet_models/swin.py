"""Swin Transformer backbone on stock PyTorch-ROCm (interface of mmdet/models/backbones/swin_transformer.py:450-631:
embed_dim, depths, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop rates, ape, patch_norm, out_indices,
frozen_stages, use_checkpoint; configs/dota/orientedrepoints_swin_tiny_demo.py:9-24).  Out of the HIP hot path: BASELINE.json
keeps the backbone on the framework.  Written from the paper (Liu et al., "Swin Transformer: Hierarchical Vision Transformer
using Shifted Windows", 2021): windowed multi-head self-attention with a learned relative-position bias, every second block
on windows shifted by half a window (cyclic roll + a mask that separates the wrapped-around regions), patch merging between
stages.  Own layout of the computation: one `scaled_dot_product_attention` call per block over all windows with the
(bias + shift mask) tensor as its additive mask, the shift masks built once per (resolution, device).

Parameter / buffer names are the released checkpoints' (`swin_tiny_patch4_window7_224.pth`): patch_embed.{proj,norm},
layers.<i>.blocks.<j>.{norm1,attn.{relative_position_bias_table,relative_position_index,qkv,proj},norm2,mlp.{fc1,fc2}},
layers.<i>.downsample.{norm,reduction}, norm<i> for every output stage, absolute_pos_embed with ape=True."""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from .registry import BACKBONES


class StochasticDepth(nn.Module):
    """Drops the residual branch of whole samples with probability p (training only), rescaling the survivors."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = torch.rand((x.size(0),) + (1,) * (x.dim() - 1), device=x.device, dtype=x.dtype) >= self.p
        return x * keep / (1.0 - self.p)


def to_windows(x, ws):
    """[B, H, W, C] (H, W multiples of ws) -> [B * nW, ws * ws, C], windows in row-major order."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def from_windows(w, ws, H, W):
    B = w.size(0) // ((H // ws) * (W // ws))
    x = w.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, -1)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim, self.ws, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        # pair (i, j) of window positions -> row of the table: (dy + ws - 1) * (2 ws - 1) + (dx + ws - 1)
        ys, xs = torch.meshgrid(torch.arange(window_size), torch.arange(window_size), indexing='ij')
        pos = torch.stack([ys.flatten(), xs.flatten()])                       # [2, N]
        rel = pos[:, :, None] - pos[:, None, :] + (window_size - 1)             # [2, N, N]
        self.register_buffer('relative_position_index', rel[0] * (2 * window_size - 1) + rel[1])
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def forward(self, x, shift_mask=None):
        """x [nWB, N, C]; shift_mask None or [nW, N, N] (0 / -100)."""
        BW, N, C = x.shape
        q, k, v = self.qkv(x).view(BW, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1)
        bias = bias.unsqueeze(0).to(q.dtype)                                   # [1, heads, N, N]
        if shift_mask is not None:
            nW = shift_mask.size(0)
            bias = (bias + shift_mask.to(q.dtype).unsqueeze(1)).repeat(BW // nW, 1, 1, 1)
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, scale=self.scale,
                                             dropout_p=self.attn_drop.p if self.training else 0.0)
        return self.proj_drop(self.proj(out.transpose(1, 2).reshape(BW, N, C)))


class Mlp(nn.Module):
    def __init__(self, dim, hidden, drop=0.):
        super().__init__()
        self.fc1, self.act, self.fc2, self.drop = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim), nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class SwinBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path):
        super().__init__()
        self.ws, self.shift = window_size, shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = StochasticDepth(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop)

    def forward(self, x, H, W, shift_mask):
        B, L, C = x.shape
        ws = self.ws
        y = self.norm1(x).view(B, H, W, C)
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws                        # pad to whole windows (bottom / right)
        if pr or pb:
            y = F.pad(y, (0, 0, 0, pr, 0, pb))
        Hp, Wp = H + pb, W + pr
        if self.shift:
            y = torch.roll(y, shifts=(-self.shift, -self.shift), dims=(1, 2))
        y = self.attn(to_windows(y, ws), shift_mask if self.shift else None)
        y = from_windows(y, ws, Hp, Wp)
        if self.shift:
            y = torch.roll(y, shifts=(self.shift, self.shift), dims=(1, 2))
        if pr or pb:
            y = y[:, :H, :W, :]
        x = x + self.drop_path(y.reshape(B, L, C))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class PatchMerging(nn.Module):
    """2 x 2 neighbourhoods concatenated (4C), normalised, projected to 2C."""

    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        # channel order of the released weights: (even row, even col), (odd, even), (even, odd), (odd, odd)
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(x.view(B, -1, 4 * C)))


class SwinStage(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path,
                 downsample, use_checkpoint):
        super().__init__()
        self.ws, self.shift, self.use_checkpoint = window_size, window_size // 2, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias, qk_scale,
                      drop, attn_drop, drop_path[i]) for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None
        self._masks = {}

    def shift_mask(self, H, W, device):
        """[nW, N, N]: 0 where two positions of a shifted window come from the same image region, -100 otherwise."""
        ws, s = self.ws, self.shift
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        key = (Hp, Wp, device)
        m = self._masks.get(key)
        if m is None:
            region = torch.zeros(Hp, Wp, device=device)
            bands = (slice(0, -ws), slice(-ws, -s), slice(-s, None))
            for a, hs in enumerate(bands):
                for b, wsl in enumerate(bands):
                    region[hs, wsl] = 3 * a + b
            r = to_windows(region.view(1, Hp, Wp, 1), ws).squeeze(-1)          # [nW, N]
            m = (r[:, None, :] != r[:, :, None]).float() * -100.0
            if len(self._masks) > 8:
                self._masks.clear()
            self._masks[key] = m
        return m

    def forward(self, x, H, W):
        mask = self.shift_mask(H, W, x.device)
        for blk in self.blocks:
            if self.use_checkpoint and x.requires_grad:
                x = cp.checkpoint(blk, x, H, W, mask, use_reentrant=False)
            else:
                x = blk(x, H, W, mask)
        if self.downsample is not None:
            return x, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm=True):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim) if norm else None

    def forward(self, x):
        p = self.patch_size
        H, W = x.shape[2:]
        if W % p or H % p:
            x = F.pad(x, (0, (p - W % p) % p, 0, (p - H % p) % p))
        x = self.proj(x)                                                       # [B, C, Wh, Ww]
        if self.norm is not None:
            Wh, Ww = x.shape[2:]
            x = self.norm(x.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(x.size(0), -1, Wh, Ww)
        return x


@BACKBONES.register_module
class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        assert norm_layer is nn.LayerNorm
        self.num_layers, self.embed_dim, self.ape = len(depths), embed_dim, ape
        self.out_indices, self.frozen_stages = tuple(out_indices), frozen_stages
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        if ape:
            side = pretrain_img_size if isinstance(pretrain_img_size, int) else pretrain_img_size[0]
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, side // patch_size, side // patch_size))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        rates = torch.linspace(0, drop_path_rate, sum(depths)).tolist()        # stochastic depth grows linearly with depth
        self.layers = nn.ModuleList()
        for i, depth in enumerate(depths):
            self.layers.append(SwinStage(embed_dim * 2 ** i, depth, num_heads[i], window_size, mlp_ratio, qkv_bias, qk_scale,
                                         drop_rate, attn_drop_rate, rates[sum(depths[:i]):sum(depths[:i + 1])],
                                         i < self.num_layers - 1, use_checkpoint))
        self.num_features = [embed_dim * 2 ** i for i in range(self.num_layers)]
        for i in self.out_indices:
            self.add_module('norm%d' % i, nn.LayerNorm(self.num_features[i]))
        self._freeze_stages()

    def _freeze_stages(self):
        frozen = []
        if self.frozen_stages >= 0:
            frozen.append(self.patch_embed)
        if self.frozen_stages >= 1:
            frozen.append(self.pos_drop)
            if self.ape:
                self.absolute_pos_embed.requires_grad = False
            frozen += list(self.layers[:self.frozen_stages - 1])
        for m in frozen:
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        """`pretrained`: a local checkpoint (state dict, or {'model': ...} / {'state_dict': ...}); no network here."""
        if isinstance(pretrained, str):
            import os
            import warnings
            if os.path.isfile(pretrained):
                sd = torch.load(pretrained, map_location='cpu')
                sd = sd.get('model', sd.get('state_dict', sd))
                sd = {k[len('backbone.'):] if k.startswith('backbone.') else k: v for k, v in sd.items()}
                missing, unexpected = self.load_state_dict(sd, strict=False)
                unexpected = [k for k in unexpected if not (k.startswith('head.') or k.startswith('norm.') or 'attn_mask' in k)]
                if missing or unexpected:
                    warnings.warn('SwinTransformer.init_weights(%r): missing %s, unexpected %s'
                                  % (pretrained, list(missing)[:8], unexpected[:8]))
                return
            warnings.warn('SwinTransformer.init_weights: pretrained=%r was NOT loaded (no such file); random initialisation'
                          % pretrained)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        H, W = x.shape[2:]
        if self.ape:
            x = x + F.interpolate(self.absolute_pos_embed, size=(H, W), mode='bicubic')
        x = self.pos_drop(x.flatten(2).transpose(1, 2))
        outs = []
        for i, layer in enumerate(self.layers):
            out, x, Hn, Wn = layer(x, H, W)
            if i in self.out_indices:
                y = getattr(self, 'norm%d' % i)(out)
                outs.append(y.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
            H, W = Hn, Wn
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

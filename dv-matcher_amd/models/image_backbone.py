"""The image backbone of the visual-feature injection path (BASELINE configs[4]): what the reference obtains
with `torch.hub.load("mhamilton723/FeatUp", 'dinov2', use_norm=True)` (train.py:72, test.py:87, deform.py:157) and calls
as `upsampler(image_tensor)` on the (3B,3,224,224) depth renderings (models/model.py:693, 964-965):

    DINOv2 ViT-S/14 patch tokens (B,384,16,16)  ->  ChannelNorm  ->  4 x joint-bilateral x2 upsamplers guided by the image
                                                                      ->  (B,384,256,256)

FeatUp and DINOv2 are third-party packages that are NOT in the reference tree (requirements.txt:44-45: an editable local
clone) and their weights cannot be fetched here, so this is a from-the-papers restatement with random initialisation:
**parity unpinned** for the network itself (architecture: Oquab et al. 2023 "DINOv2", ViT-S/14 with LayerScale; Fu et al.
2024 "FeatUp", JBU stack).  What IS kept is the interface — a callable (B,3,H,W) -> (B,384,16*h,16*w) whose `state_dict`
uses the hub model's key layout (`model.0.model.*` = the ViT, `model.1.norm.*`, `upsampler.up{1..4}.*`,
`upsampler.fixup_proj.1.*`), so the real checkpoint drops in with `load_state_dict` — and the throughput-relevant
structure: the ViT runs as PyTorch-ROCm GEMMs / fused attention (as BASELINE.json's north_star prescribes for the
pre-trained backbone), the per-pixel adaptive convolution of every JBU stage is the HIP kernel dvm_adaptive_conv_f32.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from dvm import ops


# ----------------------------------------------------------------------------------------------- DINOv2 ViT-S/14
class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, C = x.shape
        q, k, v = self.qkv(x).view(B, T, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        return self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, C))


class _LayerScale(nn.Module):
    def __init__(self, dim, init=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1, self.attn, self.ls1 = nn.LayerNorm(dim, eps=1e-6), _Attention(dim, heads), _LayerScale(dim)
        self.norm2, self.mlp, self.ls2 = nn.LayerNorm(dim, eps=1e-6), _Mlp(dim, 4 * dim), _LayerScale(dim)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class DinoV2ViT(nn.Module):
    """ViT-S/14 with DINOv2's parameter names: cls_token, pos_embed (1, 1 + 37*37, C) (trained at 518 px, interpolated
    bicubically to the input's patch grid), mask_token, patch_embed.proj, blocks.{i}.{norm1,attn.qkv,attn.proj,ls1.gamma,
    norm2,mlp.fc1,mlp.fc2,ls2.gamma}, norm."""

    def __init__(self, dim=384, depth=12, heads=6, patch=14, train_grid=37):
        super().__init__()
        self.patch_size, self.embed_dim = patch, dim
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(0.02 * torch.randn(1, 1 + train_grid * train_grid, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, dim))
        self.patch_embed = _PatchEmbed(dim, patch)
        self.blocks = nn.ModuleList(_Block(dim, heads) for _ in range(depth))
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def _pos(self, h, w):
        n = self.pos_embed.shape[1] - 1
        g = int(round(math.sqrt(n)))
        if (h, w) == (g, g):
            return self.pos_embed
        # frozen weights (no autograd, eval): the resized table is a constant of (h, w) — ATen's bicubic kernel runs this
        # small resize in ONE workgroup (0.85 ms per forward), so it is made once per version of the parameter
        frozen = not torch.is_grad_enabled() and not self.training
        key = (h, w, self.pos_embed.data_ptr(), self.pos_embed._version, str(self.pos_embed.device))
        if frozen and getattr(self, "_pos_cache", (None, None))[0] == key:
            return self._pos_cache[1]
        grid = self.pos_embed[:, 1:].reshape(1, g, g, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(h, w), mode="bicubic", align_corners=False)
        pos = torch.cat([self.pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, h * w, -1)], dim=1)
        if frozen:
            self._pos_cache = (key, pos.detach())
        return pos

    def forward_patch_tokens(self, img):
        """(B,3,H,W), H and W multiples of 14 -> normalised patch tokens as a map (B,C,H/14,W/14)."""
        B, _, H, W = img.shape
        h, w = H // self.patch_size, W // self.patch_size
        x = self.patch_embed(img)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1) + self._pos(h, w)
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)[:, 1:].reshape(B, h, w, -1).permute(0, 3, 1, 2)


class DinoFeaturizer(nn.Module):
    """FeatUp's featurizer wrapper: holds the ViT as `.model` and returns the patch-token map."""

    def __init__(self):
        super().__init__()
        self.model = DinoV2ViT()
        self.patch_size, self.dim = 14, 384

    def forward(self, img):
        return self.model.forward_patch_tokens(img)


class ChannelNorm(nn.Module):
    """LayerNorm over the channel axis of a (B,C,H,W) map (FeatUp's `use_norm=True`)."""

    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------------- joint bilateral upsampling
def _conv1x1(conv, x, post=None):
    """A 1x1 nn.Conv2d as the GEMM it is, on the library's own matrix-core kernel (MIOpen falls back to a naive fp32
    convolution for these shapes: 15 ms per call at 24 x 256^2 pixels); anything else goes through the module.
    post = (scale, r): r + scale * conv(x), in the GEMM's epilogue on the library path."""
    if x.is_cuda and conv.kernel_size == (1, 1) and not torch.is_grad_enabled():
        B, C, H, W = x.shape
        if post is not None:
            post = (post[0], post[1].reshape(B, -1, H * W))
        return ops.linear(x.reshape(B, C, H * W), conv.weight, bias=conv.bias, channel_major=True, post=post).view(B, -1, H, W)
    y = conv(x)
    return y if post is None else y * post[0] + post[1]


def _seq1x1(seq, x, post=None):
    """A Sequential of 1x1 convs / activations / dropouts; `post` applies to its LAST conv (which must end it)."""
    mods = list(seq)
    for i, m in enumerate(mods):
        last = post is not None and i == len(mods) - 1
        if last and not isinstance(m, nn.Conv2d):
            raise ValueError("_seq1x1: `post` needs the sequence to end in a convolution")
        x = _conv1x1(m, x, post if last else None) if isinstance(m, nn.Conv2d) else m(x)
    return x


class JBULearnedRange(nn.Module):
    """One x2 joint-bilateral upsampling stage: every high-resolution pixel filters the bicubically upsampled features
    with its own d x d kernel = softmax range kernel on a learned projection of the guidance image x Gaussian spatial
    kernel, plus a learned correction."""

    def __init__(self, guidance_dim, feat_dim, key_dim, scale=2, radius=3):
        super().__init__()
        self.scale, self.radius, self.diameter = scale, radius, 2 * radius + 1
        self.guidance_dim, self.key_dim, self.feat_dim = guidance_dim, key_dim, feat_dim
        d2 = self.diameter ** 2
        self.range_temp = nn.Parameter(torch.tensor(0.0))
        self.range_proj = nn.Sequential(nn.Conv2d(guidance_dim, key_dim, 1, 1), nn.GELU(), nn.Dropout2d(0.1), nn.Conv2d(key_dim, key_dim, 1, 1))
        self.fixup_proj = nn.Sequential(nn.Conv2d(guidance_dim + d2, d2, 1, 1), nn.GELU(), nn.Dropout2d(0.1), nn.Conv2d(d2, d2, 1, 1))
        self.sigma_spatial = nn.Parameter(torch.tensor(1.0))

    def range_kernel(self, guidance):
        B, _, H, W = guidance.shape
        proj = self.range_proj(guidance)
        padded = F.pad(proj, [self.radius] * 4, mode="reflect")
        temp = self.range_temp.exp().clamp(1e-4, 1e4)
        # the unfolded windows take key_dim * d*d floats per pixel (400 MB per 256 x 256 image): bound them to ~1 GB at a time
        step = max(1, int(2 ** 28 // (self.key_dim * self.diameter ** 2 * H * W)))
        out = []
        for b0 in range(0, B, step):
            win = F.unfold(padded[b0:b0 + step], self.diameter).view(-1, self.key_dim, self.diameter ** 2, H, W)
            out.append(F.softmax(temp * (win * proj[b0:b0 + step].unsqueeze(2)).sum(1), dim=1))
        return torch.cat(out) if len(out) > 1 else out[0]                                            # (B,d*d,H,W)

    def spatial_kernel(self, device):
        r = torch.linspace(-1, 1, self.diameter, device=device)
        d2 = r[:, None] ** 2 + r[None, :] ** 2
        return torch.exp(-d2 / (2 * self.sigma_spatial ** 2)).reshape(1, self.diameter ** 2, 1, 1)

    def combined_kernel_torch(self, guidance):
        """The same kernel in plain torch ops (the formulation of the FeatUp paper; the checker of dvm_jbu_kernel_f32)."""
        k = self.range_kernel(guidance) * self.spatial_kernel(guidance.device)
        return k / k.sum(1, keepdim=True).clamp_min(1e-7)

    def forward(self, source, guidance):
        B, _, H, W = guidance.shape
        if guidance.is_cuda and self.key_dim == 32 and self.diameter == 7 and not self.training:
            k = ops.jbu_kernel(_seq1x1(self.range_proj, guidance), self.range_temp, self.sigma_spatial, self.diameter)   # (B,49,H,W)
            k = _seq1x1(self.fixup_proj, torch.cat([k, guidance], dim=1), post=(0.1, k))     # k + 0.1 * fixup(...)
            return ops.adaptive_conv(ops.bicubic_resize_pad(source, (H, W), self.radius), k, tap_major=True)
        k = self.combined_kernel_torch(guidance)
        k = k + 0.1 * self.fixup_proj(torch.cat([k, guidance], dim=1))
        hr = F.pad(F.interpolate(source, size=(H, W), mode="bicubic", align_corners=False), [self.radius] * 4, mode="reflect")
        return ops.adaptive_conv(hr, k, tap_major=True)


class JBUStack(nn.Module):
    def __init__(self, feat_dim):
        super().__init__()
        self.up1, self.up2, self.up3, self.up4 = (JBULearnedRange(3, feat_dim, 32, radius=3) for _ in range(4))
        self.fixup_proj = nn.Sequential(nn.Dropout2d(0.2), nn.Conv2d(feat_dim, feat_dim, kernel_size=1))

    def forward(self, source, guidance):
        for up in (self.up1, self.up2, self.up3, self.up4):
            h, w = source.shape[2] * 2, source.shape[3] * 2
            source = up(source, F.adaptive_avg_pool2d(guidance, (h, w)))
        return _seq1x1(self.fixup_proj, source, post=(0.1, source))     # fixup(source) * 0.1 + source


class UpsampledBackbone(nn.Module):
    """`upsampler(image)`: (B,3,224,224) -> (B,384,256,256)."""

    def __init__(self, use_norm=True):
        super().__init__()
        feat = DinoFeaturizer()
        self.model = nn.Sequential(feat, ChannelNorm(feat.dim)) if use_norm else feat
        self.upsampler = JBUStack(feat.dim)
        self.patch_size, self.dim = feat.patch_size, feat.dim

    @torch.no_grad()
    def forward(self, image):
        return self.upsampler(self.model(image), image)


def load_upsampler(use_norm=True, weights=None, device="cuda", seed=0):
    """The stand-in for torch.hub.load("mhamilton723/FeatUp", 'dinov2', use_norm=use_norm): random initialisation (seeded)
    unless `weights` names a state_dict file in the hub model's layout."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        net = UpsampledBackbone(use_norm=use_norm)
    finally:
        torch.random.set_rng_state(gen_state)
    if weights:
        net.load_state_dict(torch.load(weights, map_location="cpu", weights_only=True))
    return net.to(device).eval()

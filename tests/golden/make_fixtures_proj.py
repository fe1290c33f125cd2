"""Golden vectors for the visual-feature injection path (runs in the BUILD container only).

TEST INFRASTRUCTURE.  Runs the *reference's* Uni3FC.proj2img / I2P / forward(dino_feat=None) projection branch
(models/model.py:584-710) on CPU — stub-imported through tests/golden/ref_import.py, with torch_scatter's published
`scatter(src, index, dim, reduce='sum')` semantics supplied by a few lines of torch and matplotlib (installed here)
providing the PiYG colour map — and records inputs + outputs as tests/golden/proj_*.npz.  The image backbone
(`upsampler`: FeatUp's DINOv2, not available offline) is replaced ON BOTH SIDES by a small seeded convolution whose
weights are part of the fixture.

    cd /tmp && python /root/repo/tests/golden/make_fixtures_proj.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import  # noqa: E402


def scatter(src, index, dim=1, reduce='sum'):
    """torch_scatter.scatter for the one call site (models/model.py:627): sum over dim 1, output size = max index + 1."""
    assert dim == 1 and reduce == 'sum'
    B, M, C = src.shape
    out = torch.zeros(B, int(index.max()) + 1, C, dtype=src.dtype)
    return out.scatter_add_(1, index.unsqueeze(-1).expand(-1, -1, C), src)


def stand_in_backbone(seed, C, stride):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(C, 3, 5, 5, generator=g) * 0.2
    b = torch.randn(C, generator=g) * 0.1
    return w, b, (lambda img: torch.tanh(torch.nn.functional.conv2d(img, w, b, stride=stride, padding=2)))


def clouds(seed, B, N):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, N, generator=g) * torch.tensor([0.25, 0.45, 0.15]).view(1, 3, 1)
    x[:, 1] += 0.3
    if B > 1:
        x[1] = x[1] * 1.6 + 0.2          # a second shape with another extent
    return x


def main():
    rmodel, _, _ = ref_import.import_reference()
    rmodel.scatter = scatter
    rmodel.device = 'cpu'
    torch.manual_seed(0)
    net = rmodel.Uni3FC(k=8)
    from oracle import torch_ref
    import matplotlib.cm as cm
    cmap = cm.get_cmap('PiYG')
    cmap._init()
    assert np.array_equal(cmap._lut[:256, :3].astype(np.float32), torch_ref.piyg_lut().numpy()), "PiYG table restatement is off"
    for tag, seed, B, N, C, stride in (("small", 1, 2, 400, 12, 4), ("dense", 2, 1, 3000, 8, 1), ("up", 3, 1, 500, 6, 7)):
        x = clouds(seed, B, N)
        w, b, up = stand_in_backbone(10 + seed, C, stride)
        pts_1 = rmodel.rotate_point_cloud_batch_torch(x, -np.pi / 2, axis='z')
        img, pc_min, grid, offs = net.proj2img(pts_1)
        feats = up(img)
        gathered = net.I2P(pts_1, feats, pc_min, grid, offs)
        # the whole branch: replicate forward()'s first half (models/model.py:683-710)
        pts_2 = torch.cat((pts_1[..., 2:3], pts_1[..., 0:2]), dim=-1)
        pts_3 = torch.cat((pts_1[..., 1:3], pts_1[..., 0:1]), dim=-1)
        pr = [net.proj2img(p) for p in (pts_1, pts_2, pts_3)]
        f_all = up(torch.cat([p[0] for p in pr], 0))
        clip = torch.cat([torch.nn.functional.normalize(net.I2P(p, f_all[v * B:(v + 1) * B], *pr[v][1:]), dim=-1)
                          for v, p in enumerate((pts_1, pts_2, pts_3))], dim=-1)
        lut = torch_ref.piyg_lut()
        # images are stored as colour-table indices (0..255, 255 + 1 = background) to keep the fixture small
        flat = img.permute(0, 2, 3, 1).reshape(-1, 3)
        code = torch.full((flat.shape[0],), 256, dtype=torch.int16)
        live = flat[:, 0] != -1
        d = (flat[live][:, None, :] - lut[None]).abs().sum(-1)
        assert float(d.min(1)[0].max()) == 0.0
        code[live] = d.argmin(1).to(torch.int16)
        out = dict(x=x.numpy(), conv_w=w.numpy(), conv_b=b.numpy(), stride=stride, img_code=code.view(B, 224, 224).numpy(),
                   pc_min=pc_min.numpy(), grid_size=grid.numpy(), offset_x=offs[0].numpy(), offset_y=offs[1].numpy(),
                   i2p_view1=gathered.numpy(), clip_feats=clip.numpy(), feat_hw=np.array(feats.shape[2:]))
        path = os.path.join(HERE, "proj_%s.npz" % tag)
        np.savez_compressed(path, **out)
        print("wrote %s (%.1f KB), feature map %s" % (path, os.path.getsize(path) / 1024, tuple(feats.shape)))


if __name__ == "__main__":
    main()

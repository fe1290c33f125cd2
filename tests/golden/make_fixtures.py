"""Golden-vector generator (runs in the BUILD container only).

TEST INFRASTRUCTURE.  Imports the reference (/root/reference) on CPU through
tests/golden/ref_import.py, runs its hot-path functions on
seeded inputs and writes inputs + expected outputs as small .npz fixtures into
tests/golden/.  Only the fixtures (data) are committed and travel to the GPU
box; the reference's code never does.

    cd /tmp && python /root/repo/tests/golden/make_fixtures.py [group ...]

Every RNG draw of the reference on this path is captured in the fixture
(FPS start index = nodes_idx[0]; random.sample anchors; random.randint
suffixes are irrelevant to outputs).
"""
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

REF = ref_import.REF
torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def read_off_verts(path):
    with open(path) as f:
        assert f.readline().strip() == "OFF"
        nv, nf, _ = map(int, f.readline().split())
        v = np.array([list(map(float, f.readline().split())) for _ in range(nv)], dtype=np.float32)
    return v


def scape_verts(idx, n, seed):
    v = read_off_verts(os.path.join(REF, "data/scape_r/shapes_train/mesh%03d.off" % idx))
    g = np.random.RandomState(seed)
    sel = g.permutation(v.shape[0])[:n]
    return v[sel]


def feats(kind, n, d, gen):
    x = torch.randn(n, d, generator=gen)
    if kind == "trained":
        x = 0.3 * torch.relu(x)
    return x


# --------------------------------------------------------------------------
def group_softcorr(rl):
    """rows 6,7,8,9: knnsearch_t_grad + topk_pi + Pi@verts, knnsearch_t."""
    crit = rl.GraphDeformLoss_Neural(save_name="fx")
    cases = [("randn", 256, 256, 100.0, 0), ("randn", 192, 130, 10.0, 1), ("trained", 256, 200, 33.94736842105263, 2),
             ("randn", 1024, 1024, 100.0, 3), ("trained", 700, 1024, 57.89473684210526, 4)]
    for kind, n, m, alpha, seed in cases:
        g = torch.Generator().manual_seed(100 + seed)
        f1 = feats(kind, n, 128, g)[None]
        f2 = feats(kind, m, 128, g)[None]
        v2 = torch.rand(1, m, 3, generator=g)
        alpha_np = np.float64(alpha)
        Pi = rl.knnsearch_t_grad(f1, f2, alpha=alpha_np)
        tv, ti = torch.topk(Pi, 10, dim=-1)
        Pk = crit.topk_pi(Pi)
        v12 = torch.matmul(Pk, v2)
        T12 = rl.knnsearch_t(f1, f2)
        dmm = torch.cdist(f1, f2)
        # strict gaps (so integer parity is well defined)
        dex = torch.cdist(f1, f2, compute_mode='donot_use_mm_for_euclid_dist')
        s2 = torch.topk(dex, 2, dim=-1, largest=False)[0]
        arrs = dict(feat1=f1, feat2=f2, verts2=v2, alpha=alpha_np, topk_val=tv, topk_idx=ti.int(),
                    verts12=v12, T12=T12.int(), exact_gap=(s2[..., 1] - s2[..., 0]).min(),
                    row_max=Pi.max(-1)[0], dist_mm_row0=dmm[0, 0], dist_exact_row0=dex[0, 0])
        if n * m <= 256 * 256:
            arrs["Pi_topk_dense"] = Pk
        save("softcorr_%s_%dx%d_s%d" % (kind, n, m, seed), **arrs)


def group_knn(rl, rm):
    """rows 2,10: knn_grad (xyz k=10), knn_new (feature k=40), knn (k_dist)."""
    for name, n, seed in [("rand", 256, 0), ("scape", 1024, 1), ("rand", 2048, 2)]:
        g = torch.Generator().manual_seed(200 + seed)
        if name == "scape":
            v = torch.from_numpy(scape_verts(seed, n, seed))[None]
        else:
            v = torch.rand(1, n, 3, generator=g)
        idx = rl.knn_grad(v, v, 10)
        f64 = torch.randn(1, n, 64, generator=g)
        idn = rm.knn_new(f64, f64, 40)
        f128 = torch.randn(1, n, 128, generator=g)
        sel = torch.randperm(n, generator=g)[: n // 4]
        idk = rl.knn(f128[:, sel], f128, min(100, n // 2))
        save("knn_%s_%d" % (name, n), verts=v, knn_grad_idx=idx.int(), feat64=f64, knn_new_idx=idn.int(),
             feat128=f128, anchors=sel.int(), knn_idx=idk.int())


def build_graph(rdg, verts, seed):
    """verts (N,3) torch cpu. Returns dg (reference object)."""
    torch.manual_seed(seed)
    dg = rdg.DeformationGraph_geod()
    geod = torch.cdist(verts, verts, p=2.0).cpu().numpy()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dg.construct_graph_euclidean(verts.cpu(), geod, torch.device("cpu"))
    return dg


def group_dg(rl, rdg):
    """rows 12,13,14: graph build, rot6d, warp/ARAP."""
    for name, n, seed in [("rand", 256, 0), ("scape", 512, 1), ("scape", 1024, 2), ("rand", 2048, 3)]:
        g = torch.Generator().manual_seed(300 + seed)
        if name == "scape":
            v = torch.from_numpy(scape_verts(seed, n, seed))
        else:
            v = torch.rand(n, 3, generator=g)
        dg = build_graph(rdg, v, 1234 + seed)
        nn_ = n // 2
        d6 = 0.3 * torch.randn(1, nn_, 6, generator=g)
        iden = torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float32).view(1, 1, 6)
        R = rl.rotation_6d_to_matrix(d6 + iden)
        T = 0.05 * torch.randn(1, nn_, 3, generator=g)
        warped, arap, sr = dg(v, R, T)
        save("dg_%s_%d" % (name, n), verts=v, fps_start=np.int64(dg.nodes_idx[0]), nodes_idx=dg.nodes_idx.astype(np.int32),
             one_ring=np.asarray(dg.one_ring_neigh).astype(np.int32), infl_idx=dg.influence_nodes_idx.int(),
             dists=dg.dists, weights=dg.weights, sigma=np.float64(dg.sigma), d6=d6, R=R, T=T, warped=warped,
             arap=arap, sr=sr)


def group_dg_grad(rl, rdg):
    """Gradients of the reference's OWN rot6d -> warp + ARAP chain (its autograd, fp32) w.r.t. the Deformer's outputs:
    the pin for the HIP backward kernels and for oracle/torch_ref.py's fp64 restatement (tests/test_gpu_backward.py)."""
    for name, n, seed in [("scape", 512, 1), ("rand", 256, 0)]:
        g = torch.Generator().manual_seed(300 + seed)
        v = torch.from_numpy(scape_verts(seed, n, seed)) if name == "scape" else torch.rand(n, 3, generator=g)
        dg = build_graph(rdg, v, 1234 + seed)
        nn_ = n // 2
        iden = torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float32).view(1, 1, 6)
        d6 = (0.3 * torch.randn(1, nn_, 6, generator=g) + iden).requires_grad_(True)
        T = (0.05 * torch.randn(1, nn_, 3, generator=g)).requires_grad_(True)
        gw = torch.randn(n, 3, generator=g)
        ga = torch.randn((), generator=g)
        warped, arap, _sr = dg(v, rl.rotation_6d_to_matrix(d6), T)
        ((warped * gw).sum() + arap * ga).backward()
        save("graddg_%s_%d" % (name, n), verts=v, fps_start=np.int64(dg.nodes_idx[0]), nodes_idx=dg.nodes_idx.astype(np.int32),
             one_ring=np.asarray(dg.one_ring_neigh).astype(np.int32), infl_idx=dg.influence_nodes_idx.int(), weights=dg.weights,
             d6=d6, T=T, gw=gw, ga=ga, warped=warped, arap=arap, d6_grad=d6.grad, T_grad=T.grad)


def deformer_weights():
    sd = torch.load(os.path.join(REF, "ckpt/dvmatcher_scape_r/ep_deformer_val_best.pth"), weights_only=True,
                    map_location="cpu")
    return sd


def group_deformer(rl, rm):
    """row 11 (+ real shipped weights as a data fixture) and chamfer row 15."""
    sd = deformer_weights()
    save("deformer_scape_r_weights", **{k.replace(".", "__"): v for k, v in sd.items()})
    dfm = rm.Deformer(10)
    dfm.load_state_dict(sd)
    dfm.eval()
    crit = rl.GraphDeformLoss_Neural(save_name="fx")
    for n, m, seed in [(256, 256, 0), (300, 200, 1)]:
        g = torch.Generator().manual_seed(400 + seed)
        B = 2
        f1 = 0.3 * torch.relu(torch.randn(B, n, 128, generator=g))
        f2 = 0.3 * torch.relu(torch.randn(B, m, 128, generator=g))
        v1 = torch.rand(B, n, 3, generator=g)
        v2 = torch.rand(B, m, 3, generator=g)
        Pi = crit.topk_pi(rl.knnsearch_t_grad(f1, f2, alpha=np.float64(40.0)))
        v12 = torch.matmul(Pi, v2)
        idx11 = rl.knn_grad(v1, v1, 10)
        idx22 = rl.knn_grad(v2, v2, 10)
        f1c = rl.index_points(f1, idx11)
        f2c = rl.index_points(f2, idx22)
        fps1 = torch.stack([torch.randperm(n, generator=g)[: n // 2] for _ in range(B)])
        with torch.no_grad():
            out = dfm(f1c, f2c, v1, v12, Pi, fps1)
        d1, d2, i1, i2 = crit.chamfer_dist_3d(v12, v2)
        save("deformer_%dx%d" % (n, m), feat1=f1, feat2=f2, verts1=v1, verts2=v2, alpha=np.float64(40.0), fps1=fps1.int(),
             deformations=out, verts12=v12, ch_d1=d1, ch_d2=d2, ch_i1=i1, ch_i2=i2,
             chamfer_loss=crit.chamfer_loss(v12, v2))


def run_loss(rl, rm, crit_cls, kw, B, n, m, seed, alpha, with_grad=True, scape=False, unit=False):
    sd = deformer_weights()
    dfm = rm.Deformer(10)
    dfm.load_state_dict(sd)
    dfm.train()
    g = torch.Generator().manual_seed(500 + seed)
    if unit:  # unit-scale features: |f| ~ 11 is BELOW the smallest pairwise distance (~14), sizes that do not tile
        f1 = torch.randn(B, n, 128, generator=g).requires_grad_(with_grad)
        f2 = torch.randn(B, m, 128, generator=g).requires_grad_(with_grad)
    else:
        f1 = (0.3 * torch.relu(torch.randn(B, n, 128, generator=g))).requires_grad_(with_grad)
        f2 = (0.3 * torch.relu(torch.randn(B, m, 128, generator=g))).requires_grad_(with_grad)
    if scape:
        v1 = torch.from_numpy(np.stack([scape_verts(2 * b, n, seed + b) for b in range(B)]))
        v2 = torch.from_numpy(np.stack([scape_verts(2 * b + 1, m, seed + 10 + b) for b in range(B)]))
    else:
        v1 = torch.rand(B, n, 3, generator=g)
        v2 = torch.rand(B, m, 3, generator=g)
    dist1 = torch.cdist(v1, v1)
    dist2 = torch.cdist(v2, v2)
    crit = crit_cls(save_name="fx", **kw)
    crit.device = "cpu"   # only read by the rank term's torch.eye(device=self.device) (models/loss.py:1429)
    random.seed(7000 + seed)
    torch.manual_seed(8000 + seed)
    st_py = random.getstate()
    loss, dl, dfl, ml, sl = crit(f1, f2, dist1, dist2, v1, v2, np.float64(alpha), dfm)
    out = dict(feat1=f1, feat2=f2, verts1=v1, verts2=v2, alpha=np.float64(alpha), py_seed=np.int64(7000 + seed),
               torch_seed=np.int64(8000 + seed), loss=loss, dist_loss=torch.as_tensor(dl), deform_loss=torch.as_tensor(dfl),
               map_loss=torch.as_tensor(ml), self_rec_loss=torch.as_tensor(sl))
    # captured draws: replay the python RNG to record the anchors
    if kw.get("w_dist", 1) > 0:
        random.setstate(st_py)
        a1 = random.sample(range(n), kw["N_dist"])
        a2 = random.sample(range(m), kw["N_dist"])
        out.update(anchors1=np.array(a1, np.int32), anchors2=np.array(a2, np.int32))
    if with_grad:
        loss.backward()
        out.update(g_feat1=f1.grad, g_feat2=f2.grad)
        for k, p in dfm.named_parameters():
            out["g_" + k.replace(".", "__")] = p.grad
    return out


def group_loss_rank(rl, rm):
    """The rank term (models/loss.py:1427-1433 / 1065-1071): ||Pi Pi^T - I||_F on the top-10 correspondence, w_rank > 0
    (off in the shipped configs, but it is the constructor default)."""
    kw = dict(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=30, N_dist=60, partial=False, w_deform=0.5, w_img=0,
              w_rank=0.3, w_self_rec=0.5, w_cd=0.1, w_arap=0.01)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            save("loss_full_rank_160", **run_loss(rl, rm, rl.GraphDeformLoss_Neural, kw, 2, 160, 160, 5, 30.0))
            kwp = dict(kw, w_deform=1000, w_self_rec=1000, partial=True)
            save("loss_partial_rank_144", **run_loss(rl, rm, rl.GraphDeformLoss_Neural_Partial, kwp, 2, 144, 144, 6, 45.0))
        finally:
            os.chdir(cwd)


def group_loss(rl, rm, only_unit=False):
    """rows 16,17,18 + everything composed: the criterion forward (+backward)."""
    kw = dict(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=50, N_dist=100, partial=False, w_deform=0.5, w_img=0,
              w_rank=0, w_self_rec=0.5, w_cd=0.1, w_arap=0.01)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            kwp = dict(kw, w_deform=1000, w_self_rec=1000, k_dist=30, N_dist=60, partial=True)
            if not only_unit:
                save("loss_full_256", **run_loss(rl, rm, rl.GraphDeformLoss_Neural, kw, 2, 256, 256, 0, 40.0))
                save("loss_full_scape_384", **run_loss(rl, rm, rl.GraphDeformLoss_Neural, kw, 2, 384, 384, 1, 85.6, scape=True))
                save("loss_partial_256x120", **run_loss(rl, rm, rl.GraphDeformLoss_Neural_Partial, kwp, 2, 256, 120, 2, 25.0))
            save("loss_full_300_unit", **run_loss(rl, rm, rl.GraphDeformLoss_Neural, kw, 2, 300, 300, 3, 80.0, unit=True))
            save("loss_partial_300x170_unit", **run_loss(rl, rm, rl.GraphDeformLoss_Neural_Partial, kwp, 2, 300, 170, 4, 60.0, unit=True))
        finally:
            os.chdir(cwd)


GROUPS = ["softcorr", "knn", "dg", "deformer", "loss"]


def main():
    want = sys.argv[1:] or GROUPS
    rm, rl, rdg = ref_import.import_reference()
    import builtins
    for gname in want:
        print("== group", gname)
        if gname == "softcorr":
            group_softcorr(rl)
        elif gname == "knn":
            group_knn(rl, rm)
        elif gname == "dg":
            group_dg(rl, rdg)
        elif gname == "dg_grad":
            group_dg_grad(rl, rdg)
        elif gname == "deformer":
            group_deformer(rl, rm)
        elif gname == "loss":
            group_loss(rl, rm)
        elif gname == "loss_rank":
            group_loss_rank(rl, rm)
        elif gname == "loss_unit":
            group_loss(rl, rm, only_unit=True)
        elif gname == "backbone" or gname.startswith("backbone:"):   # backbone:<fixture>,<fixture> regenerates a subset
            import make_fixtures_backbone
            make_fixtures_backbone.run(rm, save, only=gname.split(":", 1)[1].split(",") if ":" in gname else None)
        else:
            raise SystemExit("unknown group " + gname)


if __name__ == "__main__":
    main()

"""Backbone golden vectors (rows 1-5 of SURVEY §8a): pos-encoding, SA_Layer, N2PAttention[_DIM],
Uni3FC — generated from the reference (see make_fixtures.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weights_init import reinit  # noqa: E402


def run(rm, save):
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(600)
    # positional encoding (chaotic in the input: store inputs and outputs)
    net = reinit(rm.Uni3FC(k=40), salt=1)
    x = torch.rand(2, 3, 200, generator=g) * 1.7 - 0.6
    save("bb_posenc", x=x, pos=net.pos_encoding_sin_wave(x))
    # SA_Layer
    for mode in ("eval", "train"):
        sa = reinit(rm.SA_Layer(64), salt=2)
        getattr(sa, mode)()
        xs = torch.randn(2, 64, 256, generator=g)
        with torch.no_grad():
            save("bb_sa_" + mode, x=xs, out=sa(xs))
    # N2P attention blocks
    for name, cls, C in (("n2p64", rm.N2PAttention, 64), ("n2p128", rm.N2PAttention_DIM, 128)):
        for mode in ("eval", "train"):
            blk = reinit(cls(40), salt=3)
            getattr(blk, mode)()
            xs = torch.randn(2, C, 256, generator=g)
            with torch.no_grad():
                xt = xs.permute(0, 2, 1)
                save("bb_%s_%s" % (name, mode), x=xs, out=blk(xs), knn_idx=rm.knn_new(xt, xt, 40).int())
    # whole LG-Net
    for mode, B, N in (("eval", 1, 256), ("train", 2, 192)):
        net = reinit(rm.Uni3FC(k=40), salt=4)
        getattr(net, mode)()
        xyz = torch.rand(B, 3, N, generator=g)
        dino = torch.randn(B, N, 1152, generator=g).half().float()  # exactly representable in fp16 (stored as such)
        with torch.no_grad():
            feat, cf = net(xyz, dino, None)
        save("bb_uni3fc_" + mode, xyz=xyz, dino=dino.half(), feat=feat, cfeats=cf)
    # one full training step: backbone x2 -> criterion -> backward
    import random
    import tempfile
    import ref_import
    _, rl, _ = ref_import.import_reference() if False else (None, sys.modules["models.loss"], None)
    net = reinit(rm.Uni3FC(k=40), salt=5).train()
    dfm = rm.Deformer(10)
    dfm.load_state_dict(torch.load(os.path.join(ref_import.REF, "ckpt/dvmatcher_scape_r/ep_deformer_val_best.pth"),
                                   weights_only=True, map_location="cpu"))
    dfm.train()
    B, N = 2, 192
    v1, v2 = torch.rand(B, N, 3, generator=g), torch.rand(B, N, 3, generator=g)
    d1 = torch.randn(B, N, 1152, generator=g).half().float()
    d2 = torch.randn(B, N, 1152, generator=g).half().float()
    crit = rl.GraphDeformLoss_Neural(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=40, N_dist=64, partial=False, w_deform=0.5,
                                     w_img=0, w_rank=0, w_self_rec=0.5, w_cd=0.1, w_arap=0.01, save_name="fx")
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            random.seed(9001)
            torch.manual_seed(9002)
            f1, _ = net(v1.permute(0, 2, 1), d1, None)
            f2, _ = net(v2.permute(0, 2, 1), d2, None)
            out = crit(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, np.float64(57.5), dfm)
            out[0].backward()
        finally:
            os.chdir(cwd)
    arrs = dict(verts1=v1, verts2=v2, dino1=d1.half(), dino2=d2.half(), alpha=np.float64(57.5), feat1=f1, feat2=f2,
                losses=torch.stack([torch.as_tensor(o).detach().float() for o in out]))
    keep = ["conv6.0.weight", "bn6.weight", "n2p_attention7.q_conv.weight", "n2p_attention7.v_conv.weight",
            "n2p_attention1.k_conv.weight", "n2p_attention1.ff.0.weight", "sa1.q_conv.weight", "sa1.v_conv.weight",
            "sa4.trans_conv.bias", "conv0.0.weight", "bn0.bias", "conv5.0.weight"]
    named = dict(net.named_parameters())
    for k in keep:
        arrs["g_" + k.replace(".", "__")] = named[k].grad
    arrs["gnorm_backbone"] = torch.sqrt(sum((p.grad ** 2).sum() for p in net.parameters() if p.grad is not None))
    arrs["n_params_without_grad"] = np.int64(sum(1 for p in net.parameters() if p.grad is None))
    for k, p in dfm.named_parameters():
        arrs["gd_" + k.replace(".", "__")] = p.grad
    save("bb_trainstep", **arrs)
    # state_dict contract (names + shapes)
    net = rm.Uni3FC(k=40)
    sd = net.state_dict()
    save("bb_state_dict_keys", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
         dkeys=np.array(list(rm.Deformer(10).state_dict().keys())))

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
    # state_dict contract (names + shapes)
    net = rm.Uni3FC(k=40)
    sd = net.state_dict()
    save("bb_state_dict_keys", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
         dkeys=np.array(list(rm.Deformer(10).state_dict().keys())))

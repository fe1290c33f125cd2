"""Backbone golden vectors (rows 1-5, 18 of SURVEY §8a): pos-encoding, SA_Layer, N2PAttention[_DIM], Uni3FC, the
whole training step — generated from the reference (see make_fixtures.py).

The CANONICAL reference is the one evaluated with `torch.set_num_threads(1)`: oneDNN's multi-threaded Conv1d splits
the reduction across threads, so the reference's own output depends on its thread count (1 vs 8 threads: first conv
differs by 2e-6, and because the feature-space kNNs are discrete and the 7 attention layers amplify, points whose
neighbour set flipped end up with visibly different features — the reference does not reproduce itself).  With one
thread every conv is the K-blocked fp32 fma chain that oracle/dvm_oracle.c::dvo_linear restates and that
dvm_linear_f32 evaluates bit for bit.  Every whole-network fixture therefore records

  feat / cfeats        the canonical (1-thread) outputs,
  knn_idx, knn_margin  the 7 neighbour sets the reference used and each row's score gap between its 40th and 41st
                       neighbour (teacher forcing / near-tie attribution in the tests),
  feat_t8, feat_f64    the same network evaluated with 8 threads and in float64: the reference's own noise
                       envelope, against which a free-running implementation is judged.

Weights: weights_init.reinit.  With its unit gain the 11 residual attention layers of LG-Net are CHAOTIC on real shapes
(a flipped neighbour cascades; 1-thread vs 8-thread reference: half of the points of a 1024-point SCAPE shape differ by
more than 2e-3, hard maps agree on 35 % of 5000 vertices, training-step gradients differ by 10-20 %): no implementation
other than a bit-exact one can be compared with such a reference, and the reference cannot be compared with itself.
`bb_noise_summary` records those numbers.  The SCAPE-shaped fixtures therefore use gain = 0.5 (STABLE: a flip stays
local; reference vs itself: features 1e-5, losses 1e-6, gradients 1e-3), which is the regime a trained network must be
in for its maps to be reproducible at all; the small random-cloud fixtures keep gain = 1.

Inputs that are pure functions of a torch CPU generator seed are not stored (`dino_seed`); the tests regenerate them.
"""
import contextlib
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weights_init import dino_from_seed, reinit  # noqa: E402
import ref_import  # noqa: E402

K = 40
STABLE_GAIN = 0.5


def read_off(path):
    tok = open(path).read().split()
    assert tok[0] == "OFF"
    nv, nf = int(tok[1]), int(tok[2])
    v = np.array(tok[4:4 + 3 * nv], dtype=np.float64).reshape(nv, 3)
    f = np.array(tok[4 + 3 * nv:4 + 3 * nv + 4 * nf], dtype=np.int64).reshape(nf, 4)[:, 1:]
    return v.astype(np.float32), f


def scape(idx):
    return read_off(os.path.join(ref_import.REF, "data/scape_r/shapes_train/mesh%03d.off" % idx))


@contextlib.contextmanager
def record_knn(rm, log):
    """Wraps the reference's knn_new: logs the index set and each row's 40th/41st score gap."""
    orig = rm.knn_new

    def wrapped(a, b, k):
        idx = orig(a, b, k)
        inner = -2 * torch.matmul(a, b.transpose(2, 1))
        s = -torch.sum(a ** 2, dim=2, keepdim=True) - inner - torch.sum(b ** 2, dim=2, keepdim=True).transpose(2, 1)
        top = s.topk(k=k + 1, dim=-1)[0]
        log.append((idx.to(torch.int16).clone(), (top[..., k - 1] - top[..., k]).float().clone()))
        return idx

    rm.knn_new = wrapped
    try:
        yield
    finally:
        rm.knn_new = orig


def f64_twin(rm, net):
    """The same network in float64 (the positional encoding stays the fp32 one: it is chaotic by construction and part
    of the input as far as this comparison goes)."""
    import copy
    n64 = copy.deepcopy(net).double()
    n64.pos_encoding_sin_wave = lambda c: net.pos_encoding_sin_wave(c.float()).double()
    return n64


def uni3fc_case(rm, net, xyz, dino, with_noise=True):
    """Canonical forward + knn log (+ the 8-thread and float64 evaluations)."""
    out = {}
    torch.set_num_threads(1)
    log = []
    with torch.no_grad(), record_knn(rm, log):
        feat, cf = net(xyz, dino, None)
    assert len(log) == 7
    out.update(feat=feat, cfeats=cf, knn_idx=torch.stack([l[0] for l in log]), knn_margin=torch.stack([l[1] for l in log]))
    if with_noise:
        state = {k: v.clone() for k, v in net.state_dict().items()}   # train mode updates running stats: restore
        torch.set_num_threads(8)
        with torch.no_grad():
            out["feat_t8"] = net(xyz, dino, None)[0]
        net.load_state_dict(state)
        with torch.no_grad():
            out["feat_f64"] = f64_twin(rm, net)(xyz.double(), dino.double(), None)[0].float()
        net.load_state_dict(state)
        torch.set_num_threads(1)
    return out


def run(rm, save, only=None):
    torch.set_num_threads(1)
    rl = sys.modules["models.loss"]
    want = lambda name: only is None or name in only  # noqa: E731
    g = torch.Generator().manual_seed(600)
    # nn.Conv1d(k=1) [+ eval BatchNorm + LeakyReLU] evaluated by one thread, at every reduction length LG-Net uses: the
    # vectors that pin the K-blocked fma chain of dvo_linear / dvm_linear_f32 (the live torch result depends on the host
    # CPU's oneDNN kernels, so the GPU box compares against these, not against its own torch)
    if want("linear_chain"):
        arrs = {}
        for i, (Cin, Cout, N, bias) in enumerate([(64, 16, 24, True), (64, 64, 19, True), (128, 24, 24, False), (256, 16, 24, False),
                                                  (384, 16, 24, False), (512, 16, 21, False), (768, 12, 24, False), (1152, 16, 24, False)]):
            conv = torch.nn.Conv1d(Cin, Cout, 1, bias=bias)
            bn = torch.nn.BatchNorm1d(Cout).eval()
            with torch.no_grad():
                bn.weight.copy_(1 + 0.1 * torch.randn(Cout, generator=g))
                bn.bias.copy_(0.1 * torch.randn(Cout, generator=g))
                bn.running_mean.copy_(0.1 * torch.randn(Cout, generator=g))
                bn.running_var.copy_(0.5 + torch.rand(Cout, generator=g))
            xs = torch.randn(2, Cin, N, generator=g)
            with torch.no_grad():
                y = conv(xs)
                z = torch.nn.functional.leaky_relu(bn(y), 0.2)
            arrs.update({"x%d" % i: xs, "w%d" % i: conv.weight.detach(), "y%d" % i: y, "z%d" % i: z, "bn%d" % i: torch.stack(
                [bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var])})
            if bias:
                arrs["b%d" % i] = conv.bias.detach()
        save("linear_chain", n=np.int64(8), **arrs)
    # positional encoding (chaotic in the input: store inputs and outputs)
    net = reinit(rm.Uni3FC(k=40), salt=1)
    x = torch.rand(2, 3, 200, generator=g) * 1.7 - 0.6
    if want("bb_posenc"):
        save("bb_posenc", x=x, pos=net.pos_encoding_sin_wave(x))
    # SA_Layer
    for mode in ("eval", "train"):
        sa = reinit(rm.SA_Layer(64), salt=2)
        getattr(sa, mode)()
        xs = torch.randn(2, 64, 256, generator=g)
        if want("bb_sa_" + mode):
            with torch.no_grad():
                save("bb_sa_" + mode, x=xs, out=sa(xs))
    # N2P attention blocks
    for name, cls, C in (("n2p64", rm.N2PAttention, 64), ("n2p128", rm.N2PAttention_DIM, 128)):
        for mode in ("eval", "train"):
            blk = reinit(cls(40), salt=3)
            getattr(blk, mode)()
            xs = torch.randn(2, C, 256, generator=g)
            if want("bb_%s_%s" % (name, mode)):
                with torch.no_grad():
                    xt = xs.permute(0, 2, 1)
                    save("bb_%s_%s" % (name, mode), x=xs, out=blk(xs), knn_idx=rm.knn_new(xt, xt, 40).int())
    # whole LG-Net, random clouds
    for mode, B, N, seed in (("eval", 1, 256, 611), ("train", 2, 192, 612)):
        name = "bb_uni3fc_" + mode
        xyz = torch.rand(B, 3, N, generator=g)
        if not want(name):
            continue
        net = reinit(rm.Uni3FC(k=40), salt=4)
        getattr(net, mode)()
        save(name, xyz=xyz, dino_seed=np.int64(seed), **uni3fc_case(rm, net, xyz, dino_from_seed(seed, B, N)))
    # whole LG-Net on two SCAPE shapes subsampled to N = 1024 (the shapes train.py feeds it, config 1 of BASELINE.json)
    if want("bb_uni3fc_scape1024_eval"):
        v0, _ = scape(0)
        v1, _ = scape(1)
        p = torch.Generator().manual_seed(77)
        sel0, sel1 = torch.randperm(v0.shape[0], generator=p)[:1024], torch.randperm(v1.shape[0], generator=p)[:1024]
        xyz = torch.from_numpy(np.stack([v0[sel0.numpy()], v1[sel1.numpy()]])).permute(0, 2, 1).contiguous()
        dino = dino_from_seed(613, 2, 1024)
        net = reinit(rm.Uni3FC(k=40), salt=4, gain=STABLE_GAIN).eval()
        case = uni3fc_case(rm, net, xyz, dino)
        with torch.no_grad():   # the hard maps between the two sampled clouds (test.py:103-110), canonical / 8 threads / float64
            for tag in ("", "_t8", "_f64"):
                f = case["feat" + tag]
                case["T12" + tag] = rl.knnsearch_t(f[:1], f[1:])[0, :, 0].to(torch.int16)
                case["T21" + tag] = rl.knnsearch_t(f[1:], f[:1])[0, :, 0].to(torch.int16)
        save("bb_uni3fc_scape1024_eval", xyz=xyz, sel1=sel0.to(torch.int16), sel2=sel1.to(torch.int16), dino_seed=np.int64(613),
             gain=np.float64(STABLE_GAIN), **case)
        # the same shapes with unit-gain weights: the chaotic regime, statistics only
        net = reinit(rm.Uni3FC(k=40), salt=4).eval()
        c1 = uni3fc_case(rm, net, xyz, dino)
        stats = {}
        for tag in ("t8", "f64"):
            e = (c1["feat_" + tag] - c1["feat"]).abs().reshape(-1, 128).max(1)[0]
            stats["rows_gt_1e-4_" + tag] = (e > 1e-4).float().mean()
            stats["rows_gt_2e-3_" + tag] = (e > 2e-3).float().mean()
            stats["max_" + tag] = e.max()
            with torch.no_grad():
                stats["T12_agree_" + tag] = (rl.knnsearch_t(c1["feat"][:1], c1["feat"][1:]) ==
                                             rl.knnsearch_t(c1["feat_" + tag][:1], c1["feat_" + tag][1:])).float().mean()
        save("bb_noise_summary", what=np.array("reference vs itself (1 thread vs 8 threads / float64), unit-gain weights, 2 SCAPE shapes x 1024 points"),
             **stats)
    # end to end (north_star): raw shape -> Uni3FC -> knnsearch_t on two FULL SCAPE meshes, both directions, for the
    # canonical reference, its 8-thread run and its float64 twin; the meshes travel so that the test can price map
    # differences in geodesic distance (eval/geo_mat.py:15-41)
    if want("bb_e2e_scape"):
        (v0, f0), (v1, f1) = scape(0), scape(1)
        net = reinit(rm.Uni3FC(k=40), salt=4, gain=STABLE_GAIN).eval()
        x0, x1 = torch.from_numpy(v0).t()[None].contiguous(), torch.from_numpy(v1).t()[None].contiguous()
        d0, d1 = dino_from_seed(620, 1, v0.shape[0]), dino_from_seed(621, 1, v1.shape[0])
        res = {}
        for tag, nt in (("", 1), ("_t8", 8)):
            torch.set_num_threads(nt)
            with torch.no_grad():
                fa, fb = net(x0, d0, None)[0], net(x1, d1, None)[0]
                res["T12" + tag] = rl.knnsearch_t(fa, fb)[0, :, 0].to(torch.int16)
                res["T21" + tag] = rl.knnsearch_t(fb, fa)[0, :, 0].to(torch.int16)
            if nt == 1:
                res["feat1_q"] = fa[0, ::16].clone()     # every 16th point's feature (spot check of the values)
                res["feat2_q"] = fb[0, ::16].clone()
        torch.set_num_threads(8)
        n64 = f64_twin(rm, net)
        with torch.no_grad():
            fa, fb = n64(x0.double(), d0.double(), None)[0], n64(x1.double(), d1.double(), None)[0]
            res["T12_f64"] = rl.knnsearch_t(fa, fb)[0, :, 0].to(torch.int16)
            res["T21_f64"] = rl.knnsearch_t(fb, fa)[0, :, 0].to(torch.int16)
        torch.set_num_threads(1)
        save("bb_e2e_scape", verts1=v0, faces1=f0.astype(np.int16), verts2=v1, faces2=f1.astype(np.int16),
             dino_seed1=np.int64(620), dino_seed2=np.int64(621), gain=np.float64(STABLE_GAIN), **res)
    # one full training step: backbone x2 -> criterion -> backward; random clouds at N = 192 and the
    # SCAPE shapes of config 1 at N = 256 and N = 1024 (train.py:93-112)
    keep = ["conv6.0.weight", "bn6.weight", "n2p_attention7.q_conv.weight", "n2p_attention7.v_conv.weight",
            "n2p_attention1.k_conv.weight", "n2p_attention1.ff.0.weight", "sa1.q_conv.weight", "sa1.v_conv.weight",
            "sa4.trans_conv.bias", "conv0.0.weight", "bn0.bias", "conv5.0.weight"]
    keep_d = ["conv_layer.weight", "conv_layer.bias", "deformation_decoder_layer.linear.4.weight",
              "deformation_decoder_layer.linear.6.weight", "deformation_decoder_layer.linear.6.bias"]
    for name, N, seed, src in (("bb_trainstep", 192, 630, "rand"), ("bb_trainstep_scape256", 256, 631, "scape"),
                               ("bb_trainstep_scape1024", 1024, 632, "scape")):
        if not want(name):
            continue
        B = 2
        gg = torch.Generator().manual_seed(seed)
        if src == "rand":
            v1, v2 = torch.rand(B, N, 3, generator=gg), torch.rand(B, N, 3, generator=gg)
        else:   # pairs (mesh000 -> mesh001), (mesh001 -> mesh002), like consecutive items of the reference's Dataset
            vs = [scape(i)[0] for i in (0, 1, 2)]
            sel = [torch.randperm(v.shape[0], generator=gg)[:N].numpy() for v in vs]
            v1 = torch.from_numpy(np.stack([vs[0][sel[0]], vs[1][sel[1]]]))
            v2 = torch.from_numpy(np.stack([vs[1][sel[1]], vs[2][sel[2]]]))
        d1, d2 = dino_from_seed(seed + 100, B, N), dino_from_seed(seed + 200, B, N)
        k_dist, n_dist = (40, 64) if N < 512 else (300, 500)
        alpha = np.float64(57.5)
        runs = {}
        for nt in (1, 8):
            torch.set_num_threads(nt)
            net = reinit(rm.Uni3FC(k=40), salt=5, gain=STABLE_GAIN).train()
            dfm = rm.Deformer(10)
            dfm.load_state_dict(torch.load(os.path.join(ref_import.REF, "ckpt/dvmatcher_scape_r/ep_deformer_val_best.pth"),
                                           weights_only=True, map_location="cpu"))
            dfm.train()
            crit = rl.GraphDeformLoss_Neural(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=k_dist, N_dist=n_dist, partial=False,
                                             w_deform=0.5, w_img=0, w_rank=0, w_self_rec=0.5, w_cd=0.1, w_arap=0.01, save_name="fx")
            cwd = os.getcwd()
            log = []
            with tempfile.TemporaryDirectory() as td:
                os.chdir(td)
                try:
                    random.seed(9001)
                    torch.manual_seed(9002)
                    with record_knn(rm, log):
                        f1, _ = net(v1.permute(0, 2, 1), d1, None)
                        f2, _ = net(v2.permute(0, 2, 1), d2, None)
                    out = crit(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, alpha, dfm)
                    out[0].backward()
                finally:
                    os.chdir(cwd)
            runs[nt] = (net, dfm, f1, f2, out, log)
        torch.set_num_threads(1)
        net, dfm, f1, f2, out, log = runs[1]
        net8, dfm8, f1_8, _, out8, _ = runs[8]
        arrs = dict(verts1=v1, verts2=v2, dino_seed1=np.int64(seed + 100), dino_seed2=np.int64(seed + 200), alpha=alpha,
                    k_dist=np.int64(k_dist), N_dist=np.int64(n_dist), gain=np.float64(STABLE_GAIN), feat1=f1, feat1_t8=f1_8,
                    knn_idx=torch.stack([l[0] for l in log]),     # 14 sets: 7 of shape 1's forward, then 7 of shape 2's
                    knn_margin=torch.stack([l[1] for l in log]),
                    losses=torch.stack([torch.as_tensor(o).detach().float() for o in out]),
                    losses_t8=torch.stack([torch.as_tensor(o).detach().float() for o in out8]))
        named, named8 = dict(net.named_parameters()), dict(net8.named_parameters())
        for k in keep:
            arrs["g_" + k.replace(".", "__")] = named[k].grad
            arrs["g8_" + k.replace(".", "__")] = named8[k].grad
        gn = lambda n: torch.sqrt(sum((p.grad ** 2).sum() for p in n.parameters() if p.grad is not None))  # noqa: E731
        arrs["gnorm_backbone"], arrs["gnorm_backbone_t8"] = gn(net), gn(net8)
        arrs["gnorm_deformer"], arrs["gnorm_deformer_t8"] = gn(dfm), gn(dfm8)
        arrs["n_params_without_grad"] = np.int64(sum(1 for p in net.parameters() if p.grad is None))
        nd, nd8 = dict(dfm.named_parameters()), dict(dfm8.named_parameters())
        for k in (nd if name == "bb_trainstep" else keep_d):
            arrs["gd_" + k.replace(".", "__")] = nd[k].grad
            if k in keep_d:
                arrs["gd8_" + k.replace(".", "__")] = nd8[k].grad
        save(name, **arrs)
    # state_dict contract (names + shapes)
    if want("bb_state_dict_keys"):
        net = rm.Uni3FC(k=40)
        sd = net.state_dict()
        save("bb_state_dict_keys", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
             dkeys=np.array(list(rm.Deformer(10).state_dict().keys())))

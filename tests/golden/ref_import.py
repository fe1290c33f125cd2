"""Stub-import harness for the *reference* (this container only).

TEST INFRASTRUCTURE.  Imports /root/reference's hot-path modules on CPU by
stubbing the third-party packages the image lacks.
Used only by tests/golden/make_fixtures.py to emit golden vectors; nothing in
the product, the -m gpu tests, smoke() or bench.py imports this file, and
/root/reference does not exist on the GPU box.
"""
import os
import sys
import types
from unittest.mock import MagicMock

REF = os.environ.get("DVM_REFERENCE", "/root/reference")

_STUBS = [
    "torch_scatter", "featup", "featup.util", "open3d", "torchvision",
    "torchvision.transforms", "timm", "timm.models", "timm.models.layers",
    "knn_cuda", "pytorch3d", "pytorch3d.ops", "pytorch3d.ops.knn",
    "pytorch3d.structures", "pytorch3d.structures.pointclouds",
    "pytorch3d.renderer", "pytorch3d.renderer.cameras",
    "pytorch3d.renderer.mesh", "pytorch3d.renderer.mesh.rasterizer",
    "pytorch3d.renderer.mesh.shader", "pytorch3d.renderer.lighting",
    "trimesh", "potpourri3d", "ChamferDistancePytorch",
    "ChamferDistancePytorch.chamfer3D",
    "ChamferDistancePytorch.chamfer3D.dist_chamfer_3D", "torchmetrics",
    "pytorch_lightning", "torch_geometric", "psbody", "psbody.mesh", "cv2",
    "tensorboardX", "PIL", "PIL.Image",
]


def chamfer_stub_module():
    """Pure-torch squared-NN restatement standing in for the un-vendored
    ChamferDistancePytorch CUDA extension (parity unpinned)."""
    import torch
    import torch.nn as nn

    class chamfer_3DDist(nn.Module):
        def forward(self, a, b):
            d = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)
            d1, i1 = d.min(2)
            d2, i2 = d.min(1)
            return d1, d2, i1.int(), i2.int()

    return chamfer_3DDist


def import_reference():
    """Returns (models.model, models.loss, lib.deformation_graph_point) of the reference."""
    import torch
    import torch.nn as nn
    if not os.path.isdir(REF):
        raise RuntimeError("reference checkout not present at %s" % REF)
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except Exception:
            pass
        m = MagicMock(name=name)
        m.__path__ = []
        m.__name__ = name
        sys.modules[name] = m
    sys.modules["pytorch_lightning"].LightningModule = nn.Module
    sys.modules["timm.models.layers"].DropPath = nn.Identity
    torch.Tensor.cuda = lambda self, *a, **k: self
    # the reference tree must win over this repo's same-named mirror packages
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")
              or k == "lib" or k.startswith("lib.") or k == "misc" or k.startswith("misc.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import models.model as rmodel
    import models.loss as rloss
    import lib.deformation_graph_point as rdg
    rloss.dist_chamfer_3D.chamfer_3DDist = chamfer_stub_module()
    sys.path.remove(REF)
    return rmodel, rloss, rdg

"""Golden vectors for the dataset mirror (runs in the BUILD container only).

TEST INFRASTRUCTURE.  Writes a tiny synthetic cache in the reference's cache-tuple format, loads it
through the *reference's* `models.dataset.Dataset` / `testDataset` / `load_off_point_cloud`
(stub-imported, tests/golden/ref_import.py) and records what they deliver.  Only the data
(tests/golden/dataset_items.npz) is committed.

    cd /tmp && python /root/repo/tests/golden/make_fixtures_dataset.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OFF_TEXT = "OFF\n5 2 0\n0 0 0\n1.5 0 -2e-1\n0 1 0\n0.25 0.5 1\n-3 4 5.5\n3 0 1 2\n3 1 2 3\n"


def main():
    ref_import.import_reference()
    sys.path.insert(0, ref_import.REF)
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "misc" or k.startswith("misc.")]:
        del sys.modules[k]
    import models.dataset as rds
    sys.path.remove(ref_import.REF)

    g = torch.Generator().manual_seed(11)
    names = ["horse_b", "tr_reg_003", "cat_a", "tr_reg_001", "dog_c"]     # unsorted on purpose: the cache keeps its own order
    sizes = [14, 12, 17, 12, 15]
    keep = 10                                                             # stands in for the 4995-point truncation
    verts = [torch.rand(n, 3, generator=g) for n in sizes]
    fps = [torch.randperm(n, generator=g)[:keep] for n in sizes]
    dist = []
    for v in verts:
        d = torch.cdist(v, v)
        dist.append(d + 0.01 * torch.rand(d.shape, generator=g))           # not symmetric: catches a transposed sub-block
    out = {"names": np.array(names), "keep": keep, "off_text": np.array(OFF_TEXT)}
    for i in range(len(names)):
        out["verts%d" % i], out["fps%d" % i], out["dist%d" % i] = verts[i].numpy(), fps[i].numpy(), dist[i].numpy()

    with tempfile.TemporaryDirectory() as root:
        with open(os.path.join(root, "a.off"), "w") as f:
            f.write(OFF_TEXT)
        out["off_points"] = np.asarray(rds.load_off_point_cloud(os.path.join(root, "a.off")), dtype=np.float64)
        for dsname in ("scape_r", "fourleg"):
            torch.save((verts, names, fps, dist), os.path.join(root, "cache_%s_train.pt" % dsname))
            ds = rds.Dataset(root, name=dsname, train=True, use_cache=True)
            combos = np.asarray(ds.combinations, dtype=np.int64)
            out["%s_combinations" % dsname] = combos
            for p in (0, len(ds) // 2, len(ds) - 1):
                item = ds[p]
                for s in ("shape1", "shape2"):
                    out["%s_item%d_%s_xyz" % (dsname, p, s)] = item[s]["xyz"].numpy()
                    out["%s_item%d_%s_dist" % (dsname, p, s)] = item[s]["dist"].numpy()
                    out["%s_item%d_%s_name" % (dsname, p, s)] = np.array(item[s]["name"])
        torch.save((verts, names, fps), os.path.join(root, "cache_scape_r_test_test.pt"))
        ts = rds.testDataset(root, name="scape_r", train=False, use_cache=True)
        out["test_combinations"] = np.asarray(ts.combinations, dtype=np.int64)
        item = ts[7]
        for s in ("shape1", "shape2"):
            out["test_item7_%s_xyz" % s] = item[s]["xyz"].numpy()
            out["test_item7_%s_name" % s] = np.array(item[s]["name"])
            out["test_item7_%s_dist_numel" % s] = item[s]["dist"].numel()
    path = os.path.join(HERE, "dataset_items.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()

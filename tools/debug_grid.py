import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gridfiller as ogrid
from surfd_amd.meshudf import GridFiller
g = np.load("tests/golden/g10_grid_analytic.npz")
def field(c):
    return ogrid.analytic_field(c.cpu()).cuda()
for N in (64, 128):
    gf = GridFiller(N)
    udf, grads = gf.fill_grid(field, 2 ** 30)
    print(N, gf.last_stats, list(g[f"N{N}_fwd_per_level"]), int(g[f"N{N}_grad_points"]))
    ref, rg, st = ogrid.fill_grid(ogrid.analytic_field, N, 2 ** 30)
    d = (udf.cpu() != ref)
    print("  mismatching voxels:", int(d.sum()), "of", N ** 3)
    if d.any():
        idx = d.nonzero()[:10]
        for i, j, k in idx.tolist():
            print("   ", (i, j, k), float(udf[i, j, k]), float(ref[i, j, k]))
    gd = (grads.cpu() - rg).abs().amax(-1)
    print("  grad max abs diff", float(gd.max()), "nonzero mine/ref", int((grads.abs().sum(-1) > 0).sum()), int((rg.abs().sum(-1) > 0).sum()))

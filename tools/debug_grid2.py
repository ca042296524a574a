import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gridfiller as ogrid
from surfd_amd import synth
from surfd_amd.cbndec import CbnDecoder, make_udf_func
from surfd_amd.meshudf import GridFiller
dec = CbnDecoder(63, 32, 512, 5); dec.load_state_dict(synth.synth_decoder_state_dict(), strict=True); dec = dec.cuda().eval()
lat = (torch.randn(1, 32, generator=torch.Generator().manual_seed(9)) * 0.8).cuda()
N = 256
gf = GridFiller(N)
udf, grads = gf.fill_grid(make_udf_func(dec, lat), 2 ** 16)
print(gf.last_stats)
idx = torch.randint(0, N ** 3, (20000,), generator=torch.Generator().manual_seed(1))
i, j, k = idx // (N * N), (idx // N) % N, idx % N
ax = ogrid.axis_coords(N)
pts = torch.stack([ax[i], ax[j], ax[k]], 1).cuda()
direct = dec.udf(pts, 0)
stored = udf.reshape(-1)[idx.cuda()]
same = direct == stored
print("same frac", float(same.float().mean()))
bad = (~same) & (stored < 0.0398)
print("bad count", int(bad.sum()))
b = bad.nonzero().flatten()[:10].cpu()
for q in b.tolist():
    ii, jj, kk = int(i[q]), int(j[q]), int(k[q])
    print((ii, jj, kk), "stored", float(stored[q]), "direct", float(direct[q]), "diff", float(stored[q] - direct[q]),
          "corner2", float(udf[ii // 2 * 2, jj // 2 * 2, kk // 2 * 2]), "corner4", float(udf[ii // 4 * 4, jj // 4 * 4, kk // 4 * 4]))
# callback vs native on same decoder
udf_cb, _ = GridFiller(N).fill_grid(lambda c: dec.udf(c, 0), 2 ** 16, with_grads=False)
print("callback == native:", bool(torch.equal(udf_cb, udf)), int((udf_cb != udf).sum()))
# coordinates emitted by the library vs torch arithmetic
import ctypes as C
from surfd_amd import _native as NN
gf2 = GridFiller(N); L, h = gf2._native()
u = torch.empty(N, N, N, device="cuda"); 
NN.check(L.surfd_grid_begin(h, NN.ptr(u), None, NN.stream()))
n = C.c_int64()
xyz = torch.empty(32768, 3, device="cuda")
NN.check(L.surfd_grid_level_points(h, 0, NN.ptr(xyz), 32768, C.byref(n), NN.stream()))
s = N // 32
I = torch.arange(32) * s
ref = torch.stack(torch.meshgrid(ax[I], ax[I], ax[I], indexing="ij"), -1).reshape(-1, 3)
print("coords bit-equal:", bool(torch.equal(xyz.cpu(), ref)), float((xyz.cpu() - ref).abs().max()))

import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgvector_amd import api
ctx = api.Context(0, stream=0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
n, k, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 4096, 3072
means = torch.rand((1024, dim), generator=g, device=dev)
rows = torch.empty((n, dim), device=dev, dtype=torch.float16)
for lo in range(0, n, 1 << 17):
    hi = min(n, lo + (1 << 17))
    comp = torch.randint(0, 1024, (hi - lo,), generator=g, device=dev)
    rows[lo:hi] = (means[comp] + 0.1 * torch.randn((hi - lo, dim), generator=g, device=dev)).half()
centers = rows[torch.randperm(n, generator=g, device=dev)[:k]].contiguous()
for want in (False, True):
    got, gd = api.assign(ctx, api.PGV_L2SQ, api.PGV_F16, dim, centers, rows, want_dist=want)
    ctx.sync()
    r64, c64 = rows[:8192].double(), centers.double()
    got = got[:8192]
    ref = torch.cdist(r64, c64).pow(2)
    rv, ri = ref.min(dim=1)
    gv = ref.gather(1, got.long()[:, None])[:, 0]
    bad = ((gv - rv).abs() > 1e-5 * (rv.abs() + 1)).nonzero()[:, 0]
    print("want_dist", want, "mismatches", len(bad), "of 8192; n =", n)
    cn = (c64 * c64).sum(1)
    for r in bad[:8].tolist():
        gi, bi = int(got[r]), int(ri[r])
        st_g = float(cn[gi] - 2 * (r64[r] * c64[gi]).sum())
        st_b = float(cn[bi] - 2 * (r64[r] * c64[bi]).sum())
        print(" row", r, "chosen", gi, "d", float(gv[r]), "best", bi, "d", float(rv[r]), "exact s~ chosen", st_g, "best", st_b,
              "x2", float((r64[r] ** 2).sum()))
st = None

import torch, itertools
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld
dev = "cuda"
torch.manual_seed(0)
for m, k, n in [(4100, 64, 256), (4100, 256, 64), (2048, 64, 128), (2048, 96, 128), (2048, 128, 128), (2048, 160, 128), (2048, 256, 32),
                (1024, 32, 64), (5000, 134, 64), (5000, 320, 300), (3000, 67, 20), (1500, 1310, 32)]:
    x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); b = torch.randn(n, device=dev)
    want = x.double() @ w.double().t() + b.double()
    res = []
    for dma in (1, 0):
        lib().ccn_gemm_use_dma(dma)
        y = _rows(m, n, dev); y.fill_(float("nan"))
        stats = torch.zeros((lib().ccn_stats_rows(m) + 1) * 2 * n, dtype=torch.float64, device=dev)
        call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), ptr(b), ptr(y), _ld(y), m, n, k, ptr(stats))
        torch.cuda.synchronize()
        err = float((y.double() - want).abs().max())
        parts = stats[: lib().ccn_stats_rows(m) * 2 * n].view(-1, 2, n).sum(0)
        serr = float((parts[0] - want.sum(0)).abs().max() / want.sum(0).abs().max())
        bad_rows = (y.double() - want).abs().max(1)[0] > 1e-3
        res.append("dma=%d err %.2e stat %.1e badrows %d first %s" % (dma, err, serr, int(bad_rows.sum()), bad_rows.nonzero()[:4].flatten().tolist()))
    lib().ccn_gemm_use_dma(1)
    print((m, k, n), " | ".join(res))

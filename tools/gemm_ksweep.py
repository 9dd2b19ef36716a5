"""Persistent GEMM at fixed M, N over a range of K: time per tile = slices * t_slice + t_tile (per-tile overhead)."""
import torch, sys
sys.path.insert(0, '/root/repo')
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld
dev='cuda'
def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best=1e9
    for r in range(3):
        b.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best=min(best, b.elapsed_time(e)/n)
    return best
m, n = 1048576, 256
for k in (64, 128, 256, 512, 1024, 2048):
    x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); y = _rows(m, n, dev)
    t = timeit(lambda: call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None))
    tiles = (m//256)*(n//128)
    per_wg_tiles = tiles/256
    print("K=%5d  %.3f ms  %.1f TF   per tile %.2f us (T=%d slices)" % (k, t, 2*m*n*k/t/1e9, t*1e3/per_wg_tiles, k//32))
    del x, w, y

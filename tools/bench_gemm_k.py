import torch
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld
dev = "cuda"
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return b.elapsed_time(e) / n
for m, k, n in [(1048576, 128, 128), (1048576, 256, 128), (524288, 512, 128), (262144, 1024, 128), (131072, 2048, 128), (65536, 4096, 128),
                (262144, 1024, 1024), (65536, 4096, 1024), (1048576, 256, 256)]:
    x = _rows(m, k, dev); x.normal_(); w = _rows(n, k, dev, zero=True); w.normal_(); y = _rows(m, n, dev)
    res = []
    for dma in (1, 2, 0):
        lib().ccn_gemm_use_dma(dma)
        t = timeit(lambda: call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), m, n, k, None))
        res.append(2.0 * m * n * k / (t * 1e-3) / 1e12)
    lib().ccn_gemm_use_dma(1)
    print("M=%8d K=%5d N=%5d   persistent-dma %6.1f TF   dma %6.1f TF   staged %6.1f TF" % (m, k, n, res[0], res[1], res[2]))

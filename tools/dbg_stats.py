import torch, sys
sys.path.insert(0, '.')
from tests.test_gpu_gemm_h import _api, _to16, DEV
call, lib, ptr, _ = _api()
for (M,N,K) in [(128,128,64),(5000,256,256),(129,64,200)]:
    gen = torch.Generator().manual_seed(1)
    a16, lda = _to16(torch.randn(M, K, generator=gen)); w16, ldw = _to16(torch.randn(N, K, generator=gen) / K ** 0.5)
    nparts = lib().ccn_stats_rows(M)
    y = torch.empty(M, N, device=DEV)
    s1 = torch.zeros((nparts + 1) * 2 * N, dtype=torch.float64, device=DEV); s2 = torch.zeros_like(s1)
    call("gemm_nt_h", ptr(a16), lda, ptr(w16), ldw, None, ptr(y), N, M, N, K, ptr(s1), 0, 0)
    call("gemm_nt_h_stats", ptr(a16), lda, ptr(w16), ldw, None, M, N, K, ptr(s2), 0)
    d = (s1 - s2).abs()
    print(M, N, K, "max diff", float(d.max()), "nonzero", int((d > 0).sum()), "of", d.numel(), "where", d.nonzero()[:6].flatten().tolist())
    ref = torch.stack([y.double().sum(0), (y.double() ** 2).sum(0)])
    print("  vs fp64 of y: plain", float((s1[:nparts*2*N].view(nparts,2,N).sum(0) - ref).abs().max()), "stats-only", float((s2[:nparts*2*N].view(nparts,2,N).sum(0) - ref).abs().max()))

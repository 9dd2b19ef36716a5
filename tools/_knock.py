import torch, sys
from curvecloudnet_amd._lib import call, ptr, lib
from curvecloudnet_amd.ops import _rows, _ld
dev="cuda"
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    best=1e9
    for _ in range(3):
        b,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        b.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); best=min(best,b.elapsed_time(e)/n)
    return best
for m,n,k in [(1342781,256,256),(58660,1024,1024)]:
    x=_rows(m,k,dev); x.normal_(); w=_rows(n,k,dev,zero=True); w.normal_(); y=_rows(m,n,dev)
    nb=lib().ccn_gemm_x3_workspace_bytes(n,k); sc=torch.empty(nb,dtype=torch.uint8,device=dev)
    stats=torch.empty((lib().ccn_stats_rows(m)+1)*2*n,dtype=torch.float64,device=dev)
    out=[]
    for knock,label in [(0,"full"),(1,"no-dma"),(32,"no-B-dma"),(64,"no-A-dma"),(128,"no-xcd-map"),(8,"no-epilogue"),(8+64,"no-epilogue,no-A-dma"),(8+32,"no-epi,no-B-dma")]:
        lib().ccn_gemm_x3_knock(knock)
        t=timeit(lambda: call("gemm_nt_x3",ptr(x),_ld(x),ptr(w),_ld(w),None,ptr(y),_ld(y),m,n,k,ptr(stats),ptr(sc),nb))
        out.append("%s %.3f" % (label, t))
    lib().ccn_gemm_x3_knock(0)
    print(m,n,k," | ".join(out),flush=True)

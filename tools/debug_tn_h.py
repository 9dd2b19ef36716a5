import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from curvecloudnet_amd._lib import call, lib, ptr, workspace
DEV = "cuda"
M, N, K = 64, 128, 128
def run(dy, x):
    dy16 = dy.to(torch.bfloat16).to(DEV).contiguous(); x16 = x.to(torch.bfloat16).to(DEV).contiguous()
    dw = torch.zeros(N, K, device=DEV)
    nb = lib().ccn_gemm_tn_h_workspace_bytes(M, N, K); ws = workspace(nb, DEV)
    call("gemm_tn_h", ptr(dy16), N, ptr(x16), K, ptr(dw), K, M, N, K, ptr(ws), nb)
    return dw.cpu()
# which m is paired with which m: dY[m][n] = 1 only at (m*, n*), X[m][k] = m
xm = torch.arange(M).float()[:, None].expand(M, K).contiguous()
xk = torch.arange(K).float()[None, :].expand(M, K).contiguous()
for ms, ns in [(0, 0), (1, 0), (5, 3), (9, 17), (20, 40), (37, 100), (63, 127)]:
    dy = torch.zeros(M, N); dy[ms, ns] = 1
    d1, d2 = run(dy, xm), run(dy, xk)
    rows = torch.nonzero(d1.abs().sum(1)).flatten().tolist()
    print("dY one-hot (m=%d, n=%d): nonzero dW rows %s; row n* holds X-row index %s ; column check %s"
          % (ms, ns, rows[:6], sorted(set(d1[ns].tolist()))[:6], bool(torch.equal(d2[ns], torch.arange(K).float()))))
    if rows and rows != [ns]:
        r = rows[0]; print("   row", r, "values", sorted(set(d1[r].tolist()))[:8], " d2 first:", d2[r][:8].tolist())
print("---- X[m][k] = m + 1; dY one-hot at (m*, n*=5): dW[5][0] for m* = 0..63")
xm1 = xm + 1
vals = []
for ms in range(64):
    dy = torch.zeros(M, N); dy[ms, 5] = 1
    d = run(dy, xm1)
    vals.append(int(d[5][0].item()))
print(vals)
print("---- X[m][k] = k + 1, dY one-hot at (3, n*) for n* in 0..127 step 9: first 6 columns of row n*")
for ns in range(0, 128, 9):
    dy = torch.zeros(M, N); dy[3, ns] = 1
    d = run(dy, xk + 1)
    print(ns, d[ns][:6].tolist(), d[ns][60:66].tolist(), "other rows:", torch.nonzero(d.abs().sum(1)).flatten().tolist()[:5])

import sys, torch
sys.path.insert(0, '.')
from curvecloudnet_amd import ops
DEV = "cuda:0"
ops.set_mlp_dtype("bf16")
torch.manual_seed(0)
for (M, K, N) in [(556939, 64, 64), (556939, 64, 128), (289921, 259, 256), (95267, 128, 128), (556939, 160, 128)]:
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    w2 = torch.randn(17, N, device=DEV) / N ** 0.5
    bn = torch.nn.BatchNorm1d(N).to(DEV)
    for act in ("relu", "leaky_relu"):
        ref = None
        bad = 0
        for rep in range(12):
            junk = [torch.full((int(torch.randint(1, 40, (1,))) * 1000003,), float("nan"), device=DEV) for _ in range(3)]
            del junk
            with torch.no_grad():
                y = ops.linear_bn_act(x, w, None, bn, True, act, defer=True)
                out = ops.linear_bn_act(y, w2, None, None, True, None)
            cur = (y.float().clone(), out.clone())
            if ref is None:
                ref = cur
            else:
                d0, d1 = float((cur[0] - ref[0]).abs().max()), float((cur[1] - ref[1]).abs().max())
                if d0 > 0 or d1 > 0 or not bool(torch.isfinite(cur[1]).all()):
                    bad += 1
                    print("   rep", rep, "hidden diff", d0, "out diff", d1, "rows", int(((cur[0] - ref[0]).abs().amax(1) > 0).sum()))
        print((M, K, N), act, "y dtype", y.dtype, "bad repeats:", bad)

"""Stand-alone rates of the BatchNorm passes on 16-bit rows (ccn_bn_act_bwd_reduce_hz / _apply_hz and the y-operand forms)."""
import sys, torch
sys.path.insert(0, '.')
from curvecloudnet_amd._lib import call, lib, ptr
DEV = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rows, C in ((557000, 256), (1870000, 128), (290000, 512), (2300000, 64)):
    ld16 = (C + 7) // 8 * 8
    g16 = torch.randn(rows, ld16, device=DEV).bfloat16()
    z16 = torch.randn(rows, ld16, device=DEV).bfloat16()
    y32 = torch.randn(rows, C, device=DEV)
    dy16 = torch.empty(rows, ld16, dtype=torch.bfloat16, device=DEV)
    par = torch.stack([torch.rand(C) + 0.5, torch.randn(C) * 0.3, torch.randn(C) * 0.1, torch.rand(C) + 0.5]).to(DEV)
    nparts = lib().ccn_stats_rows(rows)
    sums = torch.zeros((nparts + 1) * 2 * C, dtype=torch.float64, device=DEV)
    dgb = torch.zeros(2, C, device=DEV)
    t = {}
    t["reduce_h (y fp32)"] = (timeit(lambda: call("bn_act_bwd_reduce_h", ptr(g16), ld16, ptr(y32), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), 2, 0.01, ptr(sums))), 6)
    t["reduce_hz (z bf16)"] = (timeit(lambda: call("bn_act_bwd_reduce_hz", ptr(g16), 1, ld16, ptr(z16), 1, 0, ld16, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), 2, 0.01, ptr(sums))), 4)
    t["apply_h (y fp32)"] = (timeit(lambda: call("bn_act_bwd_apply_h", ptr(g16), 1, ld16, ptr(y32), C, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), 2, 0.01, ptr(sums), float(rows), 1, 0, ptr(dy16), ld16, ptr(dgb[0]), ptr(dgb[1]), 0)), 8)
    t["apply_hz (z bf16)"] = (timeit(lambda: call("bn_act_bwd_apply_hz", ptr(g16), 1, ld16, ptr(z16), 1, 0, ld16, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), 2, 0.01, ptr(sums), float(rows), 1, 0, ptr(dy16), ld16, ptr(dgb[0]), ptr(dgb[1]))), 6)
    t["apply_hz (t bf16, relu)"] = (timeit(lambda: call("bn_act_bwd_apply_hz", ptr(g16), 1, ld16, ptr(z16), 1, 1, ld16, rows, C, ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), 1, 0.01, ptr(sums), float(rows), 1, 0, ptr(dy16), ld16, ptr(dgb[0]), ptr(dgb[1]))), 6)
    for k, (ms, bpe) in t.items():
        print("%8d x %4d  %-26s %7.3f ms  %6.2f TB/s" % (rows, C, k, ms, bpe * rows * C / ms / 1e9))

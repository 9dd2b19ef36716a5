"""diagnostic: where two runs of the split product differ (tile map), and each run's error against fp64"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from curvecloudnet_amd._lib import call, lib, ptr
from curvecloudnet_amd.ops import _ld, _rows
dev = "cuda"
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (10550, 1024, 1024)))
nb = int(lib().ccn_gemm_nt_split_workspace_bytes())
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
print("parts", lib().ccn_gemm_nt_split_parts(M, N, K, nb))
gen = torch.Generator(device=dev).manual_seed(1)
x = _rows(M, K, dev); x.normal_(generator=gen)
w = _rows(N, K, dev, zero=True); w[:, :K].normal_(generator=gen); w.mul_(K ** -0.5)
ys = []
for i in range(4):
    y = _rows(M, N, dev); y.fill_(float("nan"))
    if i == 0:
        call("gemm_nt", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), M, N, K, None)
    else:
        call("gemm_nt_ws", ptr(x), _ld(x), ptr(w), _ld(w), None, ptr(y), _ld(y), M, N, K, None, ptr(ws), nb)
    torch.cuda.synchronize()
    ys.append(y[:, :N].clone())
    print("run", i, "counters", int(ws[:4096].view(torch.int32).abs().sum()), "nan", int(torch.isnan(y[:, :N]).sum()))
ref = (x[:, :K].double() @ w[:, :K].double().t())
for i, y in enumerate(ys):
    print("run %d: max |y - fp64| = %.3e" % (i, float((y.double() - ref).abs().max())))
gm, gn = (M + 127) // 128, (N + 127) // 128
def tilemap(a, b):
    d = (a != b)
    pad = torch.zeros((gm * 128, gn * 128), dtype=torch.bool, device=dev)
    pad[:M, :N] = d
    t = pad.view(gm, 128, gn, 128).any(1).any(2)
    ids = torch.nonzero(t.flatten()).flatten().tolist()
    return ids, float((a - b).abs().max())
for i in range(1, 4):
    for j in range(i + 1, 4):
        ids, mx = tilemap(ys[i], ys[j])
        print("split run %d vs %d: %d tiles differ (max %.3e): %s" % (i, j, len(ids), mx, ids[:20]))
ids, mx = tilemap(ys[0], ys[1])
print("unsplit vs split run 1: %d tiles differ (max %.3e), first %s last %s; tiles=%d full=%d" % (len(ids), mx, ids[:5], ids[-5:], gm * gn, gm * gn // 512 * 512))

# ---- which parts does a wrong tile contain?  partial sums per K range in fp64, and the partials left in the scratch
s = lib().ccn_gemm_nt_split_parts(M, N, K, nb)
TT = K // 32 + (1 if K % 32 else 0)
full = gm * gn // 512 * 512
rem = gm * gn - full
wsf = ws[4096:].view(torch.float32)
for tt in (0, 1, rem - 1):
    tile = full + tt
    m0, n0 = (tile // gn) * 128, (tile % gn) * 128
    rows = min(128, M - m0)
    xs, wsl = x[m0:m0 + rows, :K].double(), w[n0:n0 + 128, :K].double()
    parts64 = []
    for p in range(s):
        k0, k1 = (p * TT // s) * 32, min(K, ((p + 1) * TT // s) * 32)
        parts64.append(xs[:, k0:k1] @ wsl[:, k0:k1].t())
    got = ys[1][m0:m0 + rows, n0:n0 + 128].double()
    print("tile %d (tail %d): |got - sum(all)| = %.3e" % (tile, tt, float((got - sum(parts64)).abs().max())),
          " ".join("|got - all + P%d| = %.3e" % (p, float((got - sum(parts64) + parts64[p]).abs().max())) for p in range(s)))
    # the scratch: partial p of this tile in register order [64][256]: thread t = wave*64+lane, wave: wm = wave&1, wn = wave>>1
    for p in range(s):
        raw = wsf[(tt * s + p) * 16384:(tt * s + p + 1) * 16384].view(2, 2, 16, 4, 64)      # ab, t, r, wave, lane
        tile_img = torch.zeros(128, 128, dtype=torch.float32, device=dev)
        for wave in range(4):
            wm_, wn_ = wave & 1, wave >> 1
            for ab in range(2):
                for t_ in range(2):
                    blk = raw[ab, t_, :, wave, :]                      # (16 regs, 64 lanes)
                    r_idx = torch.arange(16, device=dev)
                    lane = torch.arange(64, device=dev)
                    row = (r_idx[:, None] & 3) + 8 * (r_idx[:, None] >> 2) + 4 * (lane[None, :] >> 5)
                    col = (lane[None, :] & 31).expand(16, 64)
                    tile_img[wm_ * 64 + ab * 32 + row, wn_ * 64 + t_ * 32 + col] = blk
        print("   scratch part %d vs fp64 partial: max diff %.3e (|partial| max %.2f)" % (
            p, float((tile_img[:rows].double() - parts64[p]).abs().max()), float(parts64[p].abs().max())))

import itertools
tile = full
m0, n0 = (tile // gn) * 128, (tile % gn) * 128
xs, wsl = x[m0:m0 + 128, :K].double(), w[n0:n0 + 128, :K].double()
P = []
for p in range(s):
    k0, k1 = (p * TT // s) * 32, min(K, ((p + 1) * TT // s) * 32)
    P.append(xs[:, k0:k1] @ wsl[:, k0:k1].t())
for run in (1, 2):
    got = ys[run][m0:m0 + 128, n0:n0 + 128].double()
    print("run", run, "tile", tile, ": best integer combination of the fp64 partials per 32x32 block (rows = block row)")
    for br in range(4):
        line = []
        for bc in range(4):
            sl = (slice(br * 32, br * 32 + 32), slice(bc * 32, bc * 32 + 32))
            best = min(itertools.product(range(3), repeat=s), key=lambda c: float((got[sl] - sum(ci * Pi[sl] for ci, Pi in zip(c, P))).abs().max()))
            res = float((got[sl] - sum(ci * Pi[sl] for ci, Pi in zip(best, P))).abs().max())
            line.append("%s(%.0e)" % ("".join(str(c) for c in best), res))
        print("   ", " ".join(line))

got = ys[1][m0:m0 + 128, n0:n0 + 128].double()
G = (got - P[1] - P[2]).float()
print("G = got - P1 - P2: max %.3e mean|.| %.3e ; P0 max %.3e" % (float(G.abs().max()), float(G.abs().mean()), float(P[0].abs().max())))
# does G match some other slot of the scratch?  rebuild every slot's image and compare
def slot_img(slot):
    raw = wsf[slot * 16384:(slot + 1) * 16384].view(2, 2, 16, 4, 64)
    img = torch.zeros(128, 128, dtype=torch.float32, device=dev)
    r_idx = torch.arange(16, device=dev); lane = torch.arange(64, device=dev)
    row = (r_idx[:, None] & 3) + 8 * (r_idx[:, None] >> 2) + 4 * (lane[None, :] >> 5)
    col = (lane[None, :] & 31).expand(16, 64)
    for wave in range(4):
        for ab in range(2):
            for t_ in range(2):
                img[(wave & 1) * 64 + ab * 32 + row, (wave >> 1) * 64 + t_ * 32 + col] = raw[ab, t_, :, wave, :]
    return img
best = min(range(rem * s), key=lambda sl: float((slot_img(sl) - G).abs().max()))
print("closest scratch slot to G: %d (tile %d part %d), max diff %.3e" % (best, best // s, best % s, float((slot_img(best) - G).abs().max())))
y0t = ys[0][m0:m0 + 128, n0:n0 + 128]
print("G vs unsplit tile values: %.3e ; G vs zeros %.3e" % (float((G - y0t).abs().max()), float(G.abs().max())))
a, b = torch.sort(G.flatten().double())[0], torch.sort(P[0].flatten())[0]
print("sorted(G) vs sorted(P0): max diff %.3e" % float((a - b).abs().max()))
# per 32x32 block and per register row: is G a block / row permutation of P0?
P0f = P[0].float()
for br in range(4):
    for bc in range(4):
        g = G[br * 32:br * 32 + 32, bc * 32:bc * 32 + 32]
        hits = [(r2, c2) for r2 in range(4) for c2 in range(4)
                if float((g - P0f[r2 * 32:r2 * 32 + 32, c2 * 32:c2 * 32 + 32]).abs().max()) < 1e-4]
        print("G block (%d,%d) equals P0 block(s) %s" % (br, bc, hits), end=" | ")
    print()
# row-level: which P0 row does G row i equal (same column block)?
rowmap = []
for i in range(16):
    cand = [j for j in range(128) if float((G[i, :32] - P0f[j, :32]).abs().max()) < 1e-4]
    rowmap.append(cand)
print("G rows 0..15 (cols 0..31) equal P0 rows:", rowmap)

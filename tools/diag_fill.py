"""Fill rate of the FRNN neighbour tables in the KITTI bench: fraction of the B*Nmax*(K+1) dense SGCNN rows that are real."""
import torch
from curvecloudnet_amd import configs, ops
from curvecloudnet_amd.model import build_model
from curvecloudnet_amd.synth import make_batch, to_device

orig = ops.fast_knn
log = []


def spy(p1, p2, l1, l2, K, r, return_dists=False):
    out = orig(p1, p2, l1, l2, K, r, return_dists)
    idx = out[0] if return_dists else out
    b, n, k = idx.shape
    real = int((idx >= 0).sum())
    log.append((b, n, k, int(l1.sum()), real))
    return out


ops.fast_knn = spy
dev = torch.device("cuda")
model = build_model(configs.kitti_config(), 4, 20).to(dev).train()
data = to_device(make_batch(list(range(8))), dev)
torch.manual_seed(7)
model(data)
for b, n, k, pts, real in log:
    dense = b * n * (k + 1)
    print("B=%d Nmax=%6d K=%2d points=%7d: neighbours found %.1f%% of K, dense rows %8d, real rows (self + found) %8d = %.1f%%"
          % (b, n, k, pts, 100.0 * real / (pts * k), dense, pts + real, 100.0 * (pts + real) / dense))

"""One rank of a data-parallel rehearsal, started as a FRESH process by tests/test_gpu_parallel.py:
the reference's KITTI model section at 1/8 width + GradientAllReduce + FlatAdam, `steps` optimiser steps on this rank's
clouds; writes {params after each step, gradients of each step, bucket / collective counts} to a .pt file.

    python tests/dp_worker.py <out.pt> <mode> [steps]
modes:  dp      rank RANK of WORLD_SIZE over CCN_DIST_BACKEND (gloo / nccl) on device CCN_FORCE_DEVICE
        single  no process group: this process runs the clouds of rank CCN_AS_RANK alone
        twice   one-rank group (CCN_SINGLE_RANK_GROUP): the model runs TWICE per backward pass (two clouds, two forwards,
                one summed loss): every fused weight has two outstanding uses per bucket reduction
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    import torch
    from curvecloudnet_amd import configs
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.parallel import FlatAdam, GradientAllReduce, init_process_group_from_env, shard_clouds
    from curvecloudnet_amd.synth import make_batch, to_device
    rank, world, local_rank = init_process_group_from_env(backend=os.environ.get("CCN_DIST_BACKEND"))
    dev = torch.device("cuda", int(os.environ.get("CCN_FORCE_DEVICE", local_rank)))
    torch.cuda.set_device(dev)
    as_rank, as_world = (int(os.environ["CCN_AS_RANK"]), 2) if mode == "single" else (rank, max(world, 1))
    torch.manual_seed(1234)                                  # identical replicas on every rank
    model = build_model(configs.kitti_config(width=0.125), in_dim=4, n_out=20).to(dev).train()
    sync = GradientAllReduce(model, bucket_bytes=64 * 1024)   # several buckets at this width
    opt = FlatAdam(sync, lr=1e-3)
    clouds = shard_clouds(range(4), as_rank, as_world) if mode != "twice" else [0, 1]
    batches = [to_device(make_batch([c], n_curves=120), dev) for c in clouds]
    labels = [torch.randint(0, 20, (b.pos.size(0),), generator=torch.Generator().manual_seed(c)).to(dev)
              for b, c in zip(batches, clouds)]
    if mode != "twice":                                      # one batch of this rank's clouds per step
        data = to_device(make_batch(clouds, n_curves=120), dev)
        y = torch.cat(labels)
    rec = {"grads": [], "params": [], "buckets": len(sync.buckets), "reduce_calls": [], "world": world, "rank": rank,
           "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None}
    for step in range(steps):
        opt.zero_grad()
        torch.manual_seed(100 + step)                        # CurveFPS phases: the same draw on every rank
        if mode == "twice":
            loss = sum(segmentation_loss(model(b), l) for b, l in zip(batches, labels))
        else:
            loss = segmentation_loss(model(data), y)
        loss.backward()
        sync.finish()
        torch.cuda.synchronize()
        rec["grads"].append(torch.cat([p.grad.detach().flatten().cpu() for p in model.parameters()]))
        rec["reduce_calls"].append(sync.reduce_calls)
        opt.step()
        torch.cuda.synchronize()
        rec["params"].append(torch.cat([p.detach().flatten().cpu() for p in model.parameters()]))
    rec["lr"] = opt.param_groups[0]["lr"]
    rec["opt_state_keys"] = sorted(opt.state_dict()["state"].keys())[:3]
    torch.save(rec, out_path)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

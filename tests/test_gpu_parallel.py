"""The data-parallel code path with the REAL model on the GPU: fresh rank processes started from
here (never a fork of this process's GPU state: plain `python tests/dp_worker.py` children).

  * two ranks over gloo sharing device 0: replicas bit-identical after 2 optimiser steps, gradients = the mean of the
    two single-rank runs;
  * one rank over RCCL (backend "nccl"): RCCL initialises and all-reduces under the test runner, hooks + fused
    weight-gradient accumulation + FlatAdam, with the model run twice per backward pass.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CCN_SINGLE_RANK_GROUP", "CCN_DIST_BACKEND", "CCN_AS_RANK"):
        env.pop(k, None)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", CCN_FORCE_DEVICE="0", OMP_NUM_THREADS="4")
    env.update({k: str(v) for k, v in kw.items()})
    return env


def _run(procs):
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out.decode(errors="replace")[-3000:]


def test_two_ranks_gloo_real_model(tmp_path):
    port = _free_port()
    outs = [str(tmp_path / ("rank%d.pt" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, WORKER, outs[r], "dp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=_env(RANK=r, WORLD_SIZE=2, LOCAL_RANK=r, MASTER_PORT=port, CCN_DIST_BACKEND="gloo"))
             for r in range(2)]
    _run(procs)
    r0, r1 = (torch.load(o) for o in outs)
    assert r0["world"] == 2 and r0["backend"] == "gloo" and r0["buckets"] > 1
    # every bucket reduced exactly once per step
    assert r0["reduce_calls"] == [r0["buckets"], 2 * r0["buckets"]]
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b), "replicas diverged"
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)
    # the averaged gradient of step 0 = mean of the two ranks' own gradients (each rank alone, no group)
    singles = []
    for r in range(2):
        o = str(tmp_path / ("single%d.pt" % r))
        _run([subprocess.Popen([sys.executable, WORKER, o, "single", "1"], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, env=_env(CCN_AS_RANK=r, MASTER_PORT=port))])
        singles.append(torch.load(o)["grads"][0])
    want = (singles[0] + singles[1]) / 2
    rel = float((r0["grads"][0] - want).norm() / want.norm())
    worst = float((r0["grads"][0] - want).abs().max() / want.abs().max())
    print("2-rank gradient vs mean of single-rank runs: rel l2 %.2e, max %.2e" % (rel, worst))
    assert rel < 1e-4 and worst < 1e-4
    assert float((r0["params"][1] - r0["params"][0]).abs().max()) > 0        # the optimiser moved the weights


def test_one_rank_rccl_group_two_uses_per_backward(tmp_path):
    port = _free_port()
    o_dp, o_ref = str(tmp_path / "rccl.pt"), str(tmp_path / "plain.pt")
    _run([subprocess.Popen([sys.executable, WORKER, o_dp, "twice", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           env=_env(CCN_SINGLE_RANK_GROUP="nccl", MASTER_PORT=port))])
    _run([subprocess.Popen([sys.executable, WORKER, o_ref, "twice", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           env=_env(MASTER_PORT=port))])
    dp, ref = torch.load(o_dp), torch.load(o_ref)
    assert dp["backend"] == "nccl" and ref["backend"] is None
    assert dp["reduce_calls"] == [dp["buckets"], 2 * dp["buckets"]] and ref["reduce_calls"] == [0, 0]
    # a one-rank all-reduce is the identity: the gradient of step 0 differs by the atomic-add order of the scatter kernels
    # only.  (Later steps are not comparable entry by entry: Adam normalises every entry, so a 1e-7 gradient difference
    # on a near-zero entry becomes an lr-sized parameter difference, which moves ReLU kinks in the next forward.)
    a, b = dp["grads"][0], ref["grads"][0]
    rel = float((a - b).norm() / b.norm())
    assert rel < 1e-4, rel
    assert float((dp["params"][0] - ref["params"][0]).abs().max()) <= 2.1e-3     # one Adam step moves an entry by <= lr
    assert float((dp["params"][1] - ref["params"][1]).abs().max()) <= 4.2e-3


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_flat_adam_keeps_weights_aligned_for_the_16bit_kernels(dtype):
    """use_bias model with an odd class count + FlatAdam + the 16-bit MLP mode.  Bucket slots are 16-byte
    aligned, so the re-homed weights stay on the vector / LDS-DMA paths and ccn_gemm_nt_bf16's alignment requirement
    holds; one optimiser step runs and moves every weight."""
    from curvecloudnet_amd import configs, ops
    from curvecloudnet_amd.model import build_model, segmentation_loss
    from curvecloudnet_amd.parallel import FlatAdam, GradientAllReduce
    from curvecloudnet_amd.synth import make_batch, to_device
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = build_model(configs.shapenet_seg_config(0.125), in_dim=3, n_out=50).to(dev).train()
    assert model.use_bias and any(p.numel() % 4 for p in model.parameters())
    sync = GradientAllReduce(model)
    opt = FlatAdam(sync, lr=1e-3)
    for p in model.parameters():
        assert p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0
    data = make_batch([0, 1], n_curves=90)
    data.x = None
    data.pos = data.pos / 3.0
    data = to_device(data, dev)
    y = torch.randint(0, 50, (data.pos.size(0),), generator=torch.Generator().manual_seed(0)).to(dev)
    before = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    ops.set_mlp_dtype(dtype)
    try:
        opt.zero_grad()
        torch.manual_seed(1)
        loss = segmentation_loss(model(data, **{"shapenet-categories": torch.tensor([3, 11], device=dev)}), y)
        loss.backward()
        sync.finish()
        opt.step()
    finally:
        ops.set_mlp_dtype("fp32")
    after = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert bool(torch.isfinite(after).all()) and float((after - before).abs().max()) > 0
    # torch.optim.Adam-format state: per-parameter entries in module.parameters() order, reloadable
    sd = opt.state_dict()
    assert len(sd["state"]) == len(list(model.parameters())) and sd["param_groups"][0]["lr"] == 1e-3
    assert sd["state"][0]["exp_avg"].shape == next(model.parameters()).shape
    opt.load_state_dict(sd)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)       # the reference's scheduler drives it unchanged
    opt.step()
    sched.step()
    assert opt.param_groups[0]["lr"] == 5e-4


def test_flat_adam_state_matches_torch_adam():
    """FlatAdam against torch.optim.Adam on the same gradients: parameters after 3 steps to 1 ulp-level agreement and
    identical state_dict structure."""
    from curvecloudnet_amd.parallel import FlatAdam, GradientAllReduce
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    b = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    b.load_state_dict(a.state_dict())
    sync = GradientAllReduce(a)
    fa, ta = FlatAdam(sync, lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    x = torch.randn(11, 7, device=dev)
    for _ in range(3):
        fa.zero_grad(); ta.zero_grad()
        a(x).square().mean().backward(); b(x).square().mean().backward()
        sync.finish()
        fa.step(); ta.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert float((p - q).abs().max()) < 1e-6
    sa, sb = fa.state_dict(), ta.state_dict()
    assert sorted(sa["state"].keys()) == sorted(sb["state"].keys())
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 3.0
        assert float((sa["state"][k]["exp_avg"] - sb["state"][k]["exp_avg"]).abs().max()) < 1e-6


def test_bench_two_ranks_describes_itself(tmp_path):
    """`python bench.py --gpus 2` starts its own ranks (fresh torch.distributed.run children) and rank 0's JSON line
    carries what SURVEY section 8(e) asks of a data-parallel run: the world size the process group reports, the points per
    rank and the load imbalance (max / mean), the bytes all-reduced per step (= the gradient buckets: gradients only),
    the time finish() waits.  Rehearsed here with two gloo ranks sharing device 0 on a reduced network; a failing rank
    makes the launcher exit non-zero."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config",
           "hotpath", "--width", "0.25", "--clouds-per-gpu", "2", "--curves", "96", "--no-cpu-baseline", "--no-kernel-timing"]
    env = _env(CCN_DIST_BACKEND="gloo")
    env.pop("MASTER_ADDR", None)
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-3000:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    mg = res["multi_gpu"]
    assert mg["world_size"] == 2 and mg["world_size_observed"] == 2 and mg["backend"] == "gloo"
    assert len(mg["points_per_rank"]) == 2 and all(p > 0 for p in mg["points_per_rank"])
    assert 1.0 <= mg["load_imbalance_max_over_mean"] < 1.5
    assert mg["gradient_bytes"] > 0 and abs(mg["allreduce_bytes_per_step"] - mg["gradient_bytes"]) < 1e-6 * mg["gradient_bytes"]
    assert mg["finish_wait_ms_per_step"] >= 0.0
    # overlap really happens: a bucket's all-reduce is issued from backward when its last gradient reports; at most the one
    # bucket that closes with the very last gradient of the pass may be left for finish()
    assert mg["buckets_reduced_in_finish_per_step"] <= 1.0, mg
    # a rank that fails takes the launcher's exit code with it
    bad = subprocess.run(cmd, env=dict(env, CCN_BENCH_FAIL_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=600)
    assert bad.returncode != 0

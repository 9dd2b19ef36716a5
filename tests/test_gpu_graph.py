"""BASELINE configs[4]: the feature pass of a prepared plan captured in a hipGraph -- bit-identical logits vs the eager
pass, in eval mode (inference) and in training mode (batch statistics), fp32 and fp16 products."""
import pytest
import torch

from tests.util import batch_to

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("which,dtype,train", [("a2d2", "fp32", False), ("a2d2", "fp16", False), ("kitti", "fp32", True)])
def test_captured_forward_is_bit_identical_to_eager(which, dtype, train):
    from curvecloudnet_amd import configs, ops
    from curvecloudnet_amd.graph import CapturedForward
    from curvecloudnet_amd.model import build_model
    from curvecloudnet_amd.synth import make_batch
    cfg, n_out = (configs.a2d2_config(0.25), 55) if which == "a2d2" else (configs.kitti_config(0.25), 20)
    torch.manual_seed(4)
    model = build_model(cfg, in_dim=4, n_out=n_out).to(DEV)
    model.train(train)
    data = batch_to(make_batch([0, 1], n_curves=300, mixed_lengths=(which == "a2d2")), DEV)
    ops.set_mlp_dtype(dtype)
    try:
        torch.manual_seed(9)
        cap = CapturedForward(model, data)
        eager = cap.eager().clone()
        first = cap.replay().clone()
        second = cap.replay().clone()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(eager).all())
        if train:
            # every pass moves the BatchNorm running statistics but the logits use batch statistics
            assert torch.equal(first, eager) and torch.equal(second, eager)
        else:
            assert torch.equal(first, eager) and torch.equal(second, first)
        # and against an ordinary forward (geometry recomputed inside the call, same sampling draws)
        torch.manual_seed(9)
        with torch.no_grad():
            plain = model(data)
        assert torch.equal(plain, eager)
    finally:
        ops.set_mlp_dtype("fp32")

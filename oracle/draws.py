"""Record / replay of the random draws on the hot path (TEST INFRASTRUCTURE ONLY).

The reference draws from torch's global CPU generator with no seeding anywhere: ``torch.rand(1)`` in CurveFPS
(fps_ops.py:30), ``torch.rand(N)`` in VoxelFPS (:56), ``torch.randperm`` in SAModule (pointnet2.py:51), the random
start of farthest point sampling (point_ops.py:64).  ``Draws()`` records every such call made inside its ``with`` block;
``Draws(replay=log)`` hands the recorded values back in order to whatever runs inside it -- the reference, the oracle or
the HIP product (which takes the same draws through the same three torch functions on the CPU generator).  A golden
fixture stores the log, so it does not depend on torch's generator producing the same stream on another build.
"""
import torch

_NAMES = ("rand", "randint", "randperm")


class Draws:
    def __init__(self, replay=None):
        self.replaying = replay is not None
        self.log = list(replay) if self.replaying else []          # [(name, tensor)]
        self._saved = {}

    def _wrap(self, name):
        original = self._saved[name]

        def call(*args, **kwargs):
            if kwargs.get("generator") is not None:           # explicit generators are test inputs, not path draws
                return original(*args, **kwargs)
            if self.replaying:
                assert self.log, "more random draws than the fixture recorded (%s)" % name
                kind, value = self.log.pop(0)
                assert kind == name, "draw order differs from the fixture: %s where %s was recorded" % (name, kind)
                want = original(*args, **kwargs)              # (keeps the generator advancing as it would have)
                assert want.shape == value.shape, (name, tuple(want.shape), tuple(value.shape))
                return value.clone().to(want.dtype)
            value = original(*args, **kwargs)
            self.log.append((name, value.detach().clone()))
            return value
        return call

    def __enter__(self):
        for name in _NAMES:
            self._saved[name] = getattr(torch, name)
            setattr(torch, name, self._wrap(name))
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(torch, name, fn)
        if self.replaying and exc[0] is None:
            assert not self.log, "%d recorded draws were never taken" % len(self.log)
        return False

    # ---- fixture (de)serialisation: {prefix.draw.<i>.<name>: array}
    def to_blob(self, prefix):
        return {"%s.draw.%03d.%s" % (prefix, i, name): value.numpy() for i, (name, value) in enumerate(self.log)}

    @staticmethod
    def from_blob(blob, prefix):
        keys = sorted(k for k in blob.files if k.startswith(prefix + ".draw."))
        return [(k.rsplit(".", 1)[1], torch.from_numpy(blob[k])) for k in keys]

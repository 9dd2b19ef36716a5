"""MLP block with the state-dict layout of ``torch_geometric.nn.MLP`` (PyG 2.3.0), which the
reference instantiates everywhere (src/models/base.py:7,32,64,90-125; src/models/modules/mlp.py:13).

Keys: ``lins.{j}.weight (out,in)``, ``lins.{j}.bias``, ``norms.{j}.module.{weight,bias,running_mean,
running_var,num_batches_tracked}`` -- reference checkpoints load with ``strict=True``.
``torch.nn.Linear`` / ``torch.nn.BatchNorm1d`` are used as parameter containers only (same
initialisation as PyG); their ``forward`` is never called: every layer runs as
``ops.linear_bn_act`` (MFMA GEMM with BatchNorm statistics in the epilogue + fused scale/shift/act).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class BatchNorm(nn.Module):
    """PyG's BatchNorm wrapper: the real ``BatchNorm1d`` sits in ``.module``."""

    def __init__(self, channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.module = nn.BatchNorm1d(channels, eps=eps, momentum=momentum)


class MLP(nn.Module):
    def __init__(self, channel_list, dropout=0.0, act="relu", norm="batch_norm", plain_last=True, bias=True, **kwargs):
        super().__init__()
        if norm != "batch_norm":
            raise NotImplementedError("only batch_norm is used by the reference configs")
        if act not in ("relu", "leaky_relu"):
            raise NotImplementedError("activation %r" % (act,))
        self.channel_list = list(channel_list)
        self.act, self.plain_last = act, plain_last
        # PyG 2.3.0: a scalar dropout becomes a per-layer list and the plain last layer is never dropped
        # ([upstream, from memory of torch_geometric/nn/models/mlp.py]; every shipped config sets 0.0)
        n_layers = len(channel_list) - 1
        if isinstance(dropout, (int, float)):
            drops = [float(dropout)] * n_layers
            if plain_last and n_layers:
                drops[-1] = 0.0
        else:
            drops = [float(d) for d in dropout]
            if len(drops) != n_layers:
                raise ValueError("Number of dropout values provided (%d) does not match the number of layers "
                                 "specified (%d)" % (len(drops), n_layers))
        self.dropouts = drops
        self.dropout = max(drops) if drops else 0.0          # > 0: some layer drops (the fused first-layer forms check this)
        self.lins = nn.ModuleList(nn.Linear(a, b, bias=bias) for a, b in zip(channel_list[:-1], channel_list[1:]))
        normed = channel_list[1:-1] if plain_last else channel_list[1:]
        self.norms = nn.ModuleList(BatchNorm(c) for c in normed)

    @property
    def in_channels(self):
        return self.channel_list[0]

    @property
    def out_channels(self):
        return self.channel_list[-1]

    def forward(self, x, start=0, tail=None, post=None, post_x=None, dual=False):
        """``start`` > 0 resumes after the first ``start`` layers (a caller computed them in fused form).
        ``tail`` = (first weighted row, weights, total count): the rows from that index on stand for several identical
        rows each (compact SGCNN rows, ops.LinearBNActTail); the final plain layer needs no weights.
        ``post`` (+ ``post_x``): return that reduction of the MLP's output (ops.apply_post; handed to the plain last layer, which
        can produce the reduction's gradient in the form its own backward products read: ops.linear_bn_act).
        ``dual``: return (output, its 16-bit copy or None) -- see ops.linear_bn_act."""
        n_hidden = len(self.norms)
        for idx, (lin, norm) in enumerate(zip(self.lins, self.norms)):
            if idx < start:
                continue
            if tail is not None:
                if lin.bias is not None or self.dropout > 0.0:
                    raise NotImplementedError("weighted rows: bias / dropout layers")
                x = ops.linear_bn_act_tail(x, lin.weight, norm.module, self.training, self.act, *tail)
                continue
            # (the next layer of this MLP is the only consumer: the activation may stay unwritten, ops.LAZY_ACT)
            defer = self.dropouts[idx] == 0.0 and (idx + 1 < n_hidden or self.plain_last)
            x = ops.linear_bn_act(x, lin.weight, lin.bias, norm.module, self.training, self.act, defer=defer)
            if self.dropouts[idx] > 0.0:
                x = F.dropout(x, p=self.dropouts[idx], training=self.training)
        if self.plain_last and start <= n_hidden:
            last = self.lins[-1]
            if (post is not None or dual) and self.dropouts[-1] == 0.0:
                return ops.linear_bn_act(x, last.weight, last.bias, None, self.training, None, post=post, post_x=post_x, dual=dual)
            x = ops.linear_bn_act(x, last.weight, last.bias, None, self.training, None)
            if self.dropouts[-1] > 0.0:
                x = F.dropout(x, p=self.dropouts[-1], training=self.training)
        if dual:
            return x, None
        return x if post is None else ops.apply_post(x, post, post_x)

    def __repr__(self):
        return "MLP(%s)" % ", ".join(str(c) for c in self.channel_list)

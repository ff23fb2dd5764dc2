"""Self-attention head of the CAPL model - same constructor, parameter names
(q_map/k_map/v_map.weight) and forward contract as the reference
model/attention.py:10-48.  On the GPU with the reference's head size (64) the
softmax(q^T k / sqrt(d)) v product runs as ONE fused HIP kernel (K5, gfs3d_attention_forward_f32: no (B,N,N) tensor,
dropout decisions from a counter-based hash so that the backward regenerates them); other head sizes use the torch
formulation below.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class SelfAttention(nn.Module):
    def __init__(self, in_channel, out_channel=None, attn_dropout=0.1):
        super().__init__()
        self.in_channel = in_channel
        self.out_channel = in_channel if out_channel is None else out_channel
        self.temperature = self.out_channel ** 0.5
        for name in ("q_map", "k_map", "v_map"):          # registration order = reference order
            setattr(self, name, nn.Conv1d(in_channel, self.out_channel, 1, bias=False))
        self.dropout = nn.Dropout(attn_dropout)
        self._forced_keep_mask = None      # tests: (B,N,N) bool, replaces the RNG draw of nn.Dropout

    def forward(self, x):
        """x (B, in_channel, N) -> (B, out_channel, N)."""
        q, k, v = self.q_map(x), self.k_map(x), self.v_map(x)
        if x.is_cuda and self.out_channel == 64:
            from gfs3d import ops
            p = float(self.dropout.p) if self.training else 0.0
            mask = self._forced_keep_mask if (self.training and self._forced_keep_mask is not None) else None
            # one draw from torch's generator per call: reproducible under torch.manual_seed like nn.Dropout
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p > 0.0 and mask is None) else 0
            return ops.attention(q, k, v, 1.0 / self.temperature, p, seed, mask)
        return self._attend_torch(q, k, v)

    def forward_pm(self, x_pm):
        """x_pm (B, N, in_channel) point-major -> (B, 64, N): the three projections as ONE HIP GEMM on the concatenated
        weights (gfs3d_linear_forward_f32, channel-first q / k / v straight from its epilogue), then the fused K5 kernel."""
        from gfs3d import ops
        if self.out_channel != 64 or not x_pm.is_cuda:
            raise NotImplementedError("SelfAttention.forward_pm: head size 64 on the GPU")
        w = torch.cat((self.q_map.weight, self.k_map.weight, self.v_map.weight), dim=0)
        q, k, v = ops.linear_cf(x_pm, w)
        p = float(self.dropout.p) if self.training else 0.0
        mask = self._forced_keep_mask if (self.training and self._forced_keep_mask is not None) else None
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p > 0.0 and mask is None) else 0
        return ops.attention(q, k, v, 1.0 / self.temperature, p, seed, mask)

    def _attend_torch(self, q, k, v):
        attn = torch.bmm(q.transpose(1, 2) / self.temperature, k)
        attn = F.softmax(attn, dim=-1)
        if self._forced_keep_mask is not None and self.training:
            attn = attn * (self._forced_keep_mask.to(attn.dtype) / (1.0 - self.dropout.p))
        else:
            attn = self.dropout(attn)
        return torch.bmm(attn, v.transpose(1, 2)).transpose(1, 2)

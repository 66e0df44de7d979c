"""DGCNNAttn: kNN-graph EdgeConv self-attention.

Mirror of projects/mmdet3d_plugin/models/utils/dgcnn_attn.py:10-96 (registered in ATTENTION by the reference, used by
no shipped config): same constructor (`embed_dims, num_heads, dropout, K=...`), `forward` signature and state-dict keys
(`conv1.0.weight`, `conv1.1.{weight,bias,running_mean,running_var,num_batches_tracked}`, same for `conv2`).
Reproduced quirks: the neighbours are the K FARTHEST points (topk of the distances, :84-86); the second stage always
uses K = 16 (default argument of edge_feats, :75).

Eval mode runs on HIP: gd4d_knn_farthest_fwd (no (B, N, N) distance matrix in memory), the 1x1 convolution as one
(N, C) x (C, 2C) Linear (conv(cat(x_j, x_i)) = W[:, :C] x_j + W[:, C:] x_i), gd4d_edge_conv_max_fwd (BatchNorm affine,
ReLU, max over K; no (B, 2C, N, K) edge tensor).  Training mode (BatchNorm batch statistics, autograd) is the
reference's op sequence in torch.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .registry import ATTENTION


@ATTENTION.register_module()
class DGCNNAttn(nn.Module):
    def __init__(self, embed_dims, num_heads, dropout=0., init_cfg=None, **kwargs):
        super().__init__()
        self.embed_dims, self.num_heads, self.init_cfg = embed_dims, num_heads, init_cfg
        self.conv1 = nn.Sequential(nn.Conv2d(embed_dims * 2, embed_dims, kernel_size=1, bias=False),
                                   nn.BatchNorm2d(embed_dims), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(nn.Conv2d(embed_dims * 2, embed_dims, kernel_size=1, bias=False),
                                   nn.BatchNorm2d(embed_dims), nn.ReLU(inplace=True))
        self.K = kwargs['K']
        self.dropout = nn.Dropout(dropout)
        # Training this module (batch-statistics BatchNorm, autograd) has no kernels of its own: `torch_ops=True` in the config dict
        # (or module.torch_ops = True / Fn.torch_ops_for(module) / GD4D_TORCH_OPS=1) chooses its torch-op route - for THIS module
        # only; everything else keeps its kernels.
        self.torch_ops = bool(kwargs.get('torch_ops', False))

    # ---- HIP path (eval) ------------------------------------------------------------------------------------------
    @staticmethod
    def _stage(x, conv, bn, k):
        """x (B, N, C) -> (B, N, C): edge_feats + conv + BatchNorm(eval) + ReLU + max over the K neighbours."""
        c = x.shape[-1]
        w = conv.weight.view(c, 2 * c)
        w_lin = torch.cat([w[:, :c], w[:, c:]], 0).contiguous()                      # rows: W_a (neighbour), W_b (self)
        ab = Fn.linear(x, w_lin)                                                      # (B, N, 2C)
        idx = ops.knn_farthest_fwd(x.contiguous(), k)
        alpha = bn.weight * torch.rsqrt(bn.running_var + bn.eps)                      # eval-mode BatchNorm as an affine
        beta = bn.bias - bn.running_mean * alpha
        return ops.edge_conv_max_fwd(ab, idx, alpha.contiguous(), beta.contiguous())

    # ---- reference op sequence (training / autograd) ---------------------------------------------------------------
    @staticmethod
    def _edge_feats(x, k):
        idx = torch.topk(torch.cdist(x, x), k=k, dim=2)[1]                            # :84-86
        b, n, c = x.shape
        nbr = x.reshape(b * n, c)[(idx + torch.arange(b, device=x.device).view(-1, 1, 1) * n).view(-1)].view(b, n, k, c)
        return torch.cat((nbr, x.view(b, n, 1, c).expand(-1, -1, k, -1)), -1).permute(0, 3, 1, 2).contiguous()

    def forward(self, query, key=None, value=None, residual=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        """query (N, B, C) -> (N, B, C) (:41-80)."""
        if residual is None:
            residual = query
        x = query if query_pos is None else query + query_pos
        x = x.permute(1, 0, 2)                                                        # (B, N, C)
        Fn.require_gpu(query, 'query')
        if (self.training or Fn.wants_grad(self, query, query_pos)) and Fn.torch_ops_route(
                'DGCNNAttn with autograd / in train mode (batch-statistics BatchNorm; the kernels are inference-only)', False, module=self):
            f1 = self.conv1(self._edge_feats(x, self.K)).max(dim=-1)[0]               # (B, C, N)
            f2 = self.conv2(self._edge_feats(f1.permute(0, 2, 1), 16)).max(dim=-1)[0]
            return residual + self.dropout((f1 + f2).permute(2, 0, 1))
        x = x.contiguous().float()
        f1 = self._stage(x, self.conv1[0], self.conv1[1], self.K)
        f2 = self._stage(f1, self.conv2[0], self.conv2[1], 16)
        return residual + (f1 + f2).permute(1, 0, 2)

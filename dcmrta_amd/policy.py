"""Stock-PyTorch attention policy with the reference's forward contract (attention.py:288-297).

The policy is a CONSUMER of the hot path and out of scope as a product: a maintainer keeps
using the reference's own attention.py unchanged.  The reference file cannot travel to the GPU box, so BASELINE
config 3 ("attention policy on PyTorch-ROCm + HIP env step") and the batched runner need a stand-in with the same
architecture and tensor interface:

    net(tasks f32[B,T+1,5], agents f32[B,A,6], mask bool[B,T+1]) -> logp f32[B,T+1]

Architecture (attention.py:248-297): Linear embeddings, one encoder layer each for tasks
and agents, a 2-layer cross decoder (tasks attend to agents), two 2-layer global decoders and a single-head
pointer with 10*tanh clipping and log-softmax.  Rows that are all -1 are padding (attention.py:10-18).
Written from the architecture description with head-fused projections (one [D, H*dk] matmul per Q/K/V instead of
per-head bmm); `load_reference_state_dict` maps a reference checkpoint onto it.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _pad_rows(x):
    """attention.py:14-15: a row is padding when every feature equals -1."""
    return x.eq(-1).all(dim=2)


class MultiHead(nn.Module):
    """attention.py:84-153: bias-free multi-head attention, masked logits -inf, masked probabilities forced to 0."""

    def __init__(self, dim, heads=8):
        super().__init__()
        self.h, self.dk = heads, dim // heads
        self.wq = nn.Parameter(torch.empty(dim, dim))
        self.wk = nn.Parameter(torch.empty(dim, dim))
        self.wv = nn.Parameter(torch.empty(dim, dim))
        self.wo = nn.Parameter(torch.empty(dim, dim))
        for p in (self.wq, self.wk, self.wv):
            nn.init.uniform_(p, -1 / math.sqrt(self.dk), 1 / math.sqrt(self.dk))  # attention.py:101-104
        nn.init.uniform_(self.wo, -1 / math.sqrt(dim), 1 / math.sqrt(dim))

    def forward(self, q, kv=None, mask=None):
        kv = q if kv is None else kv
        B, Nq, D = q.shape
        Nk = kv.shape[1]
        Q = (q @ self.wq).view(B, Nq, self.h, self.dk).transpose(1, 2)   # B,h,Nq,dk
        K = (kv @ self.wk).view(B, Nk, self.h, self.dk).transpose(1, 2)
        V = (kv @ self.wv).view(B, Nk, self.h, self.dk).transpose(1, 2)
        U = (Q @ K.transpose(2, 3)) / math.sqrt(self.dk)                 # attention.py:130
        if mask is not None:
            m = mask.view(B, 1, -1, Nk).expand_as(U)
            U = U.masked_fill(m, float("-inf"))                          # :135
        P = torch.softmax(U, dim=-1)
        if mask is not None:
            P = P.masked_fill(m, 0.0)                                    # :138-141 (also clears all-masked NaN rows)
        heads = (P @ V).transpose(1, 2).reshape(B, Nq, D)                # :144-148 heads concatenated
        return heads @ self.wo


class GatedFFN(nn.Module):
    """attention.py:156-182: LN(x + W2(sigmoid(W x) * V x)), hidden 512, no biases."""

    def __init__(self, dim, hidden=512):
        super().__init__()
        self.W = nn.Linear(dim, hidden, bias=False)
        self.V = nn.Linear(dim, hidden, bias=False)
        self.W2 = nn.Linear(hidden, dim, bias=False)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(x + self.W2(torch.sigmoid(self.W(x)) * self.V(x)))


class Block(nn.Module):
    """Encoder layer (attention.py:193-206) when kv is None, decoder layer (:209-223, cross-attention only) otherwise."""

    def __init__(self, dim, heads):
        super().__init__()
        self.attn = MultiHead(dim, heads)
        self.norm = nn.LayerNorm(dim)
        self.ffn = GatedFFN(dim)

    def forward(self, x, kv=None, mask=None):
        return self.ffn(self.norm(self.attn(x, kv, mask) + x))


class Pointer(nn.Module):
    """attention.py:28-81: log_softmax(10*tanh(Q K^T / sqrt(D))) with masked logits = -1e4."""

    def __init__(self, dim):
        super().__init__()
        self.wq = nn.Parameter(torch.empty(dim, dim))
        self.wk = nn.Parameter(torch.empty(dim, dim))
        for p in (self.wq, self.wk):
            nn.init.uniform_(p, -1 / math.sqrt(dim), 1 / math.sqrt(dim))
        self.scale = 1 / math.sqrt(dim)

    def forward(self, q, h, mask):
        U = 10.0 * torch.tanh(self.scale * ((q @ self.wq) @ (h @ self.wk).transpose(1, 2)))
        U = U.masked_fill(mask.view(U.shape[0], -1, U.shape[2]).expand_as(U), -1e4)
        return torch.log_softmax(U, dim=-1)


class AttentionNet(nn.Module):
    def __init__(self, agent_input_dim=6, task_input_dim=5, embedding_dim=128, heads=8):
        super().__init__()
        D = embedding_dim
        self.agent_embedding = nn.Linear(agent_input_dim, D)
        self.task_embedding = nn.Linear(task_input_dim, D)
        self.task_encoder = nn.ModuleList([Block(D, heads)])                      # Encoder(n_layer=1) :254
        self.agent_encoder = nn.ModuleList([Block(D, heads)])                     # :256
        self.cross_decoder = nn.ModuleList([Block(D, heads) for _ in range(2)])   # :255
        self.global_decoder1 = nn.ModuleList([Block(D, heads) for _ in range(2)])  # :257
        self.global_decoder2 = nn.ModuleList([Block(D, heads) for _ in range(2)])  # :258
        self.pointer = Pointer(D)

    def forward(self, tasks, agents, mask):
        tpad, apad = _pad_rows(tasks), _pad_rows(agents)                           # attention.py:289-291
        task_mask = tpad.unsqueeze(2) | tpad.unsqueeze(1)                          # B,T+1,T+1
        agent_mask = apad.unsqueeze(2) | apad.unsqueeze(1)                         # B,A,A
        task_agent_mask = tpad.unsqueeze(2) | apad.unsqueeze(1)                    # B,T+1,A
        task_emb = self.task_embedding(tasks)                                      # :263
        x = task_emb
        for blk in self.task_encoder:
            x = blk(x, None, task_mask)
        task_enc = x
        keep = (~task_mask[:, 0, :]).unsqueeze(2).to(task_emb.dtype)               # :267-269 nanmean over non-pad rows
        compressed = (task_emb * keep).sum(1, keepdim=True) / keep.sum(1, keepdim=True)
        y = self.agent_embedding(agents)                                           # :273-274
        for blk in self.agent_encoder:
            y = blk(y, None, agent_mask)
        feat = task_enc
        for blk in self.cross_decoder:                                             # :278
            feat = blk(feat, y, task_agent_mask)
        state = compressed
        for blk in self.global_decoder1:                                           # :295
            state = blk(state, y, None)
        m = mask.view(mask.shape[0], 1, -1)
        for blk in self.global_decoder2:                                           # :282
            state = blk(state, feat, m)
        return self.pointer(state, feat, m).squeeze(1)                             # :283-284


def _fuse_heads(w):
    """reference per-head weight [H, D, dk] -> fused [D, H*dk]."""
    return w.permute(1, 0, 2).reshape(w.shape[1], -1)


def load_reference_state_dict(net, ref_sd):
    """Map a state_dict of the reference AttentionNet (attention.py) onto this module."""
    sd = {}

    def block(dst, src, attn="multiHeadAttention", ln="normalization1"):
        sd[f"{dst}.attn.wq"] = _fuse_heads(ref_sd[f"{src}.{attn}.w_query"])
        sd[f"{dst}.attn.wk"] = _fuse_heads(ref_sd[f"{src}.{attn}.w_key"])
        sd[f"{dst}.attn.wv"] = _fuse_heads(ref_sd[f"{src}.{attn}.w_value"])
        sd[f"{dst}.attn.wo"] = ref_sd[f"{src}.{attn}.w_out"].reshape(-1, ref_sd[f"{src}.{attn}.w_out"].shape[-1])
        sd[f"{dst}.norm.weight"] = ref_sd[f"{src}.{ln}.normalizer.weight"]
        sd[f"{dst}.norm.bias"] = ref_sd[f"{src}.{ln}.normalizer.bias"]
        for k in ("W", "V", "W2"):
            sd[f"{dst}.ffn.{k}.weight"] = ref_sd[f"{src}.feedForward.DenseReluDense.{k}.weight"]
        sd[f"{dst}.ffn.norm.weight"] = ref_sd[f"{src}.feedForward.layer_norm.normalizer.weight"]
        sd[f"{dst}.ffn.norm.bias"] = ref_sd[f"{src}.feedForward.layer_norm.normalizer.bias"]

    for k in ("agent_embedding", "task_embedding"):
        sd[f"{k}.weight"], sd[f"{k}.bias"] = ref_sd[f"{k}.weight"], ref_sd[f"{k}.bias"]
    block("task_encoder.0", "taskEncoder.layers.0")
    block("agent_encoder.0", "agentEncoder.layers.0")
    for dst, src in (("cross_decoder", "crossDecoder"), ("global_decoder1", "globalDecoder1"), ("global_decoder2", "globalDecoder2")):
        for i in range(2):
            block(f"{dst}.{i}", f"{src}.layers.{i}", ln="normalization")
    sd["pointer.wq"], sd["pointer.wk"] = ref_sd["pointer.w_query"], ref_sd["pointer.w_key"]
    net.load_state_dict(sd)
    return net

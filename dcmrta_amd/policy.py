"""Stock-PyTorch attention policy with the reference's forward contract (attention.py:288-297).

The policy is a CONSUMER of the hot path and out of scope as a product: a maintainer keeps
using the reference's own attention.py unchanged.  The reference file cannot travel to the GPU box, so BASELINE
config 3 ("attention policy on PyTorch-ROCm + HIP env step") and the batched runner need a stand-in with the same
architecture and tensor interface:

    net(tasks f32[B,T+1,5], agents f32[B,A,6], mask bool[B,T+1]) -> logp f32[B,T+1]

Architecture (attention.py:248-297): Linear embeddings, one encoder layer each for tasks
and agents, a 2-layer cross decoder (tasks attend to agents), two 2-layer global decoders and a single-head
pointer with 10*tanh clipping and log-softmax.  Rows that are all -1 are padding (attention.py:10-18).

Written from the architecture description for throughput at rollout batch sizes (thousands of envs per forward, where
the forward is bound by HBM traffic of the activations), with stock torch ops only: one fused [D, 3D] projection for
Q|K|V (the reference runs 3 x 8 per-head bmm), one fused [D, 2H] projection for the value and gate halves of the
feed-forward followed by `F.glu`, `F.scaled_dot_product_attention` instead of materialised [B, 8, N, N] score / mask /
probability tensors, both residual additions folded into the GEMM epilogue (`torch.addmm`), and LayerNorm evaluated
through the (2x faster at D = 128) group-norm kernels.  `rollout_copy(dtype)`
gives a bf16 / fp16 shadow for rollouts (weights converted once instead of autocast's per-forward casts, LayerNorm reads
and writes the low-precision activations directly; the pointer logits and the log-softmax stay fp32).
`load_reference_state_dict` maps a reference checkpoint onto it.
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _pad_rows(x):
    """attention.py:14-15: a row is padding when every feature equals -1."""
    return x.eq(-1).all(dim=2)


def _layer_norm(x, ln):
    """nn.LayerNorm over the last dim, evaluated as a one-group GroupNorm of the [rows, D] matrix: the same arithmetic
    (per-row moments over D, per-channel affine), but at D = 128 and ~2e5 rows torch's layer-norm kernel takes 213 us per
    call on MI355X and its group-norm kernels 110 us (examples/policy_tools/policy_bench.py).  Only without autograd (rollouts): the BACKWARD of
    that group-norm call reduces the affine parameters' gradients over the ~1e5 rows one "sample" at a time
    (GammaBeta1dBackwardCUDAKernel2: 94 of the 162 ms of a forward + backward over 8192 decisions, examples/policy_tools/policy_bwd_breakdown.py),
    so a forward that is going to be differentiated uses the layer-norm kernels."""
    if torch.is_grad_enabled() and (x.requires_grad or ln.weight.requires_grad):
        return F.layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, ln.eps)
    return F.group_norm(x.reshape(-1, x.shape[-1]), 1, ln.weight, ln.bias, ln.eps).view(x.shape)


class MultiHead(nn.Module):
    """attention.py:84-153: bias-free multi-head attention, masked logits -inf, masked probabilities forced to 0."""

    def __init__(self, dim, heads=8):
        super().__init__()
        self.h, self.dk, self.dim = heads, dim // heads, dim
        self.wqkv = nn.Parameter(torch.empty(dim, 3 * dim))        # columns: Q heads | K heads | V heads
        self.wo = nn.Parameter(torch.empty(dim, dim))
        nn.init.uniform_(self.wqkv, -1 / math.sqrt(self.dk), 1 / math.sqrt(self.dk))  # attention.py:101-104
        nn.init.uniform_(self.wo, -1 / math.sqrt(dim), 1 / math.sqrt(dim))

    def forward(self, q, kv=None, mask=None):
        """q + MHA(q, kv): mask bool [B, Nq or 1, Nk], True = attention not possible (attention.py:132-141), or None.
        The residual of the calling layer (attention.py:201,217) rides on the output projection's GEMM."""
        B, Nq, D = q.shape
        h, dk = self.h, self.dk
        if kv is None:
            Q, K, V = (q @ self.wqkv).view(B, Nq, 3, h, dk).permute(2, 0, 3, 1, 4)       # each B,h,Nq,dk
        else:
            Nk = kv.shape[1]
            Q = (q @ self.wqkv[:, :D]).view(B, Nq, h, dk).transpose(1, 2)
            K, V = (kv @ self.wqkv[:, D:]).view(B, Nk, 2, h, dk).permute(2, 0, 3, 1, 4)
        if mask is None:
            out = F.scaled_dot_product_attention(Q, K, V)                                # attention.py:130,143-144
        else:
            keep = ~mask.view(B, 1, -1, K.shape[2])
            out = F.scaled_dot_product_attention(Q, K, V, attn_mask=keep)
            # a query whose keys are ALL masked: the reference zeroes its (NaN) probabilities, i.e. outputs 0 (:138-141)
            dead = mask.view(B, -1, K.shape[2]).all(dim=2)                               # B, Nq or 1
            out = out.masked_fill(dead.view(B, 1, -1, 1), 0.0)
            out = torch.nan_to_num(out, nan=0.0)
        heads = out.transpose(1, 2).reshape(B * Nq, D)                                   # :144-148 heads concatenated
        return torch.addmm(q.reshape(B * Nq, D), heads, self.wo).view(B, Nq, D)


class GatedFFN(nn.Module):
    """attention.py:156-182: LN(x + W2(sigmoid(W x) * V x)), hidden 512, no biases."""

    def __init__(self, dim, hidden=512):
        super().__init__()
        self.hidden = hidden
        self.VW = nn.Linear(dim, 2 * hidden, bias=False)           # rows: value V | gate W  (F.glu: first half * sigmoid(second))
        self.W2 = nn.Linear(hidden, dim, bias=False)
        self.norm = nn.LayerNorm(dim)

    # rows per pass of the rollout-size evaluation below: the [rows, 2 hidden] intermediate of one pass is 128 MiB
    CHUNK_BYTES = 128 << 20

    def forward(self, x):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        n = self.CHUNK_BYTES // (2 * self.hidden * x2.element_size())
        if x2.element_size() >= 4 and x2.shape[0] > 2 * n and not (torch.is_grad_enabled() and (x2.requires_grad or self.W2.weight.requires_grad)):
            # Rollout batch sizes (2e5 token rows): in row chunks, so that the [rows, 1024] product of the first GEMM is still in
            # the 256 MB MALL when F.glu and the second GEMM read it back instead of making two round trips through HBM
            # (1.15 -> 1.01 ms per call in fp32 at 4096 x 51 rows, examples/policy_tools/ffn_chunk_probe.py; +1.5 ... 3 % on every fp32 row of
            # bench_configs.py --config 3).  fp32 only: with TunableOp-selected GEMMs the 2-byte shadows were 3 % FASTER in one
            # pass.  Same arithmetic per row; hipBLASLt may pick another tile for the smaller M (differences of the last bit).
            y = torch.empty_like(x2)
            for lo in range(0, x2.shape[0], n):
                xs = x2[lo:lo + n]
                torch.addmm(xs, F.glu(xs @ self.VW.weight.t(), dim=-1), self.W2.weight.t(), out=y[lo:lo + n])
        else:
            y = torch.addmm(x2, F.glu(x2 @ self.VW.weight.t(), dim=-1), self.W2.weight.t())   # x + W2(V x * sigmoid(W x))
        return _layer_norm(y, self.norm).view(shape)


class Block(nn.Module):
    """Encoder layer (attention.py:193-206) when kv is None, decoder layer (:209-223, cross-attention only) otherwise."""

    def __init__(self, dim, heads):
        super().__init__()
        self.attn = MultiHead(dim, heads)
        self.norm = nn.LayerNorm(dim)
        self.ffn = GatedFFN(dim)

    def forward(self, x, kv=None, mask=None):
        return self.ffn(_layer_norm(self.attn(x, kv, mask), self.norm))


class Pointer(nn.Module):
    """attention.py:28-81: log_softmax(10*tanh(Q K^T / sqrt(D))) with masked logits = -1e4.  Always fp32."""

    def __init__(self, dim):
        super().__init__()
        self.wq = nn.Parameter(torch.empty(dim, dim))
        self.wk = nn.Parameter(torch.empty(dim, dim))
        for p in (self.wq, self.wk):
            nn.init.uniform_(p, -1 / math.sqrt(dim), 1 / math.sqrt(dim))
        self.scale = 1 / math.sqrt(dim)

    def forward(self, q, h, mask):
        """q [B,1,D] (one query per env), h [B,N,D], mask [B,1,N].  (q Wq)(h Wk)^T is evaluated as ((q Wq) Wk^T) h^T: the
        same product re-associated, so the N-token side is never projected (at rollout batch sizes the fp32 [B*N,D]x[D,D]
        projection of h was 5 % of the whole forward); fp32 accumulation throughout."""
        with torch.autocast(device_type=q.device.type, enabled=False):
            v = (q.float() @ self.wq) @ self.wk.t()                                   # B,1,D   (tiny)
            U = (h * v).sum(-1, dtype=torch.float32).unsqueeze(1)                     # B,1,N   fp32 products and sums
            U = 10.0 * torch.tanh(self.scale * U)
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
        # True: the caller guarantees that no row of `tasks` / `agents` is padding (a uniform batch: every env has the
        # batch's own A and T), so the three padding masks are all-False and the encoders / cross decoder run unmasked
        self.assume_no_padding = False

    def rollout_copy(self, dtype=torch.bfloat16):
        """A low-precision shadow of this net for rollouts (no gradients): every weight converted once to `dtype`, the
        pointer kept in fp32.  `sync_rollout_copy(shadow)` refreshes it in place after a weight update (graph-safe)."""
        shadow = copy.deepcopy(self).to(dtype)
        shadow.pointer.float()
        shadow.assume_no_padding = self.assume_no_padding
        for p in shadow.parameters():
            p.requires_grad_(False)
        return shadow.eval()

    @torch.no_grad()
    def sync_rollout_copy(self, shadow):
        for (_, src), (_, dst) in zip(self.state_dict().items(), shadow.state_dict().items()):
            dst.copy_(src)
        return shadow

    def forward(self, tasks, agents, mask):
        dt = self.task_embedding.weight.dtype
        tasks, agents = tasks.to(dt), agents.to(dt)
        if self.assume_no_padding:
            task_mask = agent_mask = task_agent_mask = None
            task_emb = self.task_embedding(tasks)                                  # attention.py:263
            compressed = task_emb.mean(1, keepdim=True)                            # :267-269 (no pad rows: plain mean)
        else:
            tpad, apad = _pad_rows(tasks), _pad_rows(agents)                       # attention.py:289-291
            task_mask = tpad.unsqueeze(2) | tpad.unsqueeze(1)                      # B,T+1,T+1
            agent_mask = apad.unsqueeze(2) | apad.unsqueeze(1)                     # B,A,A
            task_agent_mask = tpad.unsqueeze(2) | apad.unsqueeze(1)                # B,T+1,A
            task_emb = self.task_embedding(tasks)                                  # :263
            keep = (~tpad).unsqueeze(2).to(task_emb.dtype)                         # :267-269 nanmean over non-pad rows
            compressed = (task_emb * keep).sum(1, keepdim=True) / keep.sum(1, keepdim=True)
        x = task_emb
        for blk in self.task_encoder:
            x = blk(x, None, task_mask)
        y = self.agent_embedding(agents)                                           # :273-274
        for blk in self.agent_encoder:
            y = blk(y, None, agent_mask)
        feat = x
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
        sd[f"{dst}.attn.wqkv"] = torch.cat([_fuse_heads(ref_sd[f"{src}.{attn}.w_{k}"]) for k in ("query", "key", "value")], dim=1)
        sd[f"{dst}.attn.wo"] = ref_sd[f"{src}.{attn}.w_out"].reshape(-1, ref_sd[f"{src}.{attn}.w_out"].shape[-1])
        sd[f"{dst}.norm.weight"] = ref_sd[f"{src}.{ln}.normalizer.weight"]
        sd[f"{dst}.norm.bias"] = ref_sd[f"{src}.{ln}.normalizer.bias"]
        ff = f"{src}.feedForward.DenseReluDense"
        sd[f"{dst}.ffn.VW.weight"] = torch.cat([ref_sd[f"{ff}.V.weight"], ref_sd[f"{ff}.W.weight"]], dim=0)
        sd[f"{dst}.ffn.W2.weight"] = ref_sd[f"{ff}.W2.weight"]
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

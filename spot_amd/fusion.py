"""TransT feature-fusion network on the HIP kernels.

Host-side mirror of ltr/models/neck/transt_fusion.py: same class names,
constructor arguments, forward signature and state-dict keys
(`encoder.layers.N.{self_attn1,self_attn2,multihead_attn1,multihead_attn2}.
{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}`, `linear11..22`,
`norm11..23`, `decoder.layers.0.{multihead_attn,linear1,linear2,norm1,norm2}`,
`decoder.norm`), so reference checkpoints load unchanged.  The modules below only
hold parameters; every arithmetic step is a kernel from libspot_hip.so.

Internally tokens are batch-first [B, L, C] (the reference is [L, B, C]).
Dropout (train mode): attention-probability dropout inside the fused attention kernels, output and
FFN-hidden dropouts in the GEMM epilogues, all with counter-based masks that backward regenerates
from a seed (ops.manual_seed); the random stream differs from torch's, the distribution does not.
"""
import math


import torch
from torch import nn

from . import ops


class _LinearParams(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.empty(n_out))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(n_in)
        nn.init.uniform_(self.bias, -bound, bound)


class _NormParams(nn.Module):
    def __init__(self, d, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))
        self.eps = eps

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class _AttentionParams(nn.Module):
    """Parameter layout of nn.MultiheadAttention (packed in-projection)."""

    def __init__(self, d_model, nhead, dropout=0.0):
        super().__init__()
        assert d_model % nhead == 0
        self.embed_dim, self.num_heads, self.dropout = d_model, nhead, dropout
        self.attn_bf16 = False      # bf16 matrix products in the attention core (FeatureFusionNetwork.set_attention_dtype)
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = _LinearParams(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, xq, xkv, pos_q, pos_k, kpm, add_residual, self_attn, p_out=0.0):
        """p_out: the nn.Dropout the layer applies to this block's output before the residual add."""
        p_attn = self.dropout if self.training else 0.0
        return ops.mha_block(xq, xkv, pos_q, pos_k, self.in_proj_weight, self.in_proj_bias,
                             self.out_proj.weight, self.out_proj.bias, kpm, add_residual, self.num_heads, self_attn,
                             p_attn, p_out, self.attn_bf16)


# One-clip inference under ops.frozen_weights (the online tracker): issue the two streams' launches in pairs
# (see FeatureFusionLayer._forward_paired).
PAIRED_LAUNCHES = True
# Run the template stream of every fusion layer on a second HIP stream (see _forward_two_streams).
TWO_STREAMS = True
_SIDE_STREAMS = {}
if TWO_STREAMS and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    # parameters of the template stream receive their gradients on the second stream by design
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


def side_stream(device):
    """The second stream that belongs to the current stream (one per (device, current stream) pair, so passes
    issued from different streams do not share a side stream)."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def _ffn(x, l1, l2, p=0.0):
    return ops.FFNFn.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, p)


class FeatureFusionLayer(nn.Module):
    """ECA on both streams, CFA both ways, FFN on both (transt_fusion.py:167-244)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        if activation != "relu":
            raise RuntimeError("activation should be relu, not %s." % activation)
        self.self_attn1 = _AttentionParams(d_model, nhead, dropout)
        self.self_attn2 = _AttentionParams(d_model, nhead, dropout)
        self.multihead_attn1 = _AttentionParams(d_model, nhead, dropout)
        self.multihead_attn2 = _AttentionParams(d_model, nhead, dropout)
        self.linear11 = _LinearParams(d_model, dim_feedforward)
        self.linear12 = _LinearParams(dim_feedforward, d_model)
        self.linear21 = _LinearParams(d_model, dim_feedforward)
        self.linear22 = _LinearParams(dim_feedforward, d_model)
        for n in ("norm11", "norm12", "norm13", "norm21", "norm22", "norm23"):
            setattr(self, n, _NormParams(d_model))
        self.dropout = dropout

    def forward(self, src1, src2, kpm1=None, kpm2=None, pos_src1=None, pos_src2=None):
        p = self.dropout if self.training else 0.0      # dropout11..23, dropout1/2 of the reference
        if PAIRED_LAUNCHES and src1.is_cuda and src1.shape[0] == 1 and ops.frozen_weights._active and not torch.is_grad_enabled():
            return self._forward_paired(src1, src2, kpm1, kpm2, pos_src1, pos_src2)
        if TWO_STREAMS and src1.is_cuda:
            return self._forward_two_streams(src1, src2, kpm1, kpm2, pos_src1, pos_src2, p)
        src1 = self.norm11(self.self_attn1(src1, src1, pos_src1, pos_src1, kpm1, True, True, p))
        src2 = self.norm21(self.self_attn2(src2, src2, pos_src2, pos_src2, kpm2, True, True, p))
        # both cross attentions read the post-ECA streams (:223-230)
        z1 = self.multihead_attn1(src1, src2, pos_src1, pos_src2, kpm2, True, False, p)
        z2 = self.multihead_attn2(src2, src1, pos_src2, pos_src1, kpm1, True, False, p)
        src1 = self.norm12(z1)
        src1 = self.norm13(_ffn(src1, self.linear11, self.linear12, p))
        src2 = self.norm22(z2)
        src2 = self.norm23(_ffn(src2, self.linear21, self.linear22, p))
        return src1, src2

    def _forward_paired(self, src1, src2, kpm1, kpm2, pos_src1, pos_src2):
        """One clip, inference (the online tracker): the launches of such a pass last as long as their latencies, so
        the two streams' kernels go out in pairs - the library calls of either stream are collected, then issued side
        by side through the two-problem entry points (one launch for both projections, both attention cores, both
        LayerNorms ...).  The arithmetic of each call is unchanged."""
        from . import lib
        with lib.record() as ra:
            a1 = self.norm11(self.self_attn1(src1, src1, pos_src1, pos_src1, kpm1, True, True, 0.0))
        with lib.record() as rb:
            a2 = self.norm21(self.self_attn2(src2, src2, pos_src2, pos_src2, kpm2, True, True, 0.0))
        lib.run_paired(ra.calls, rb.calls)
        with lib.record() as ra:
            z1 = self.multihead_attn1(a1, a2, pos_src1, pos_src2, kpm2, True, False, 0.0)
            out1 = self.norm13(_ffn(self.norm12(z1), self.linear11, self.linear12, 0.0))
        with lib.record() as rb:
            z2 = self.multihead_attn2(a2, a1, pos_src2, pos_src1, kpm1, True, False, 0.0)
            out2 = self.norm23(_ffn(self.norm22(z2), self.linear21, self.linear22, 0.0))
        lib.run_paired(ra.calls, rb.calls)
        return out1, out2

    def _forward_two_streams(self, src1, src2, kpm1, kpm2, pos_src1, pos_src2, p):
        """Same computation with the template stream (256 tokens per clip) on a second HIP stream: its kernels
        are too small to fill the chip and run beside the search stream's (1024 tokens) instead of between
        them.  The streams meet where cross attention needs the other stream's post-ECA tokens."""
        main = torch.cuda.current_stream()
        side = side_stream(src1.device)
        ops.register_side_stream(side, main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            src1 = self.norm11(self.self_attn1(src1, src1, pos_src1, pos_src1, kpm1, True, True, p))
        src2 = self.norm21(self.self_attn2(src2, src2, pos_src2, pos_src2, kpm2, True, True, p))
        main.wait_stream(side)
        side.wait_stream(main)
        src1.record_stream(main)
        src2.record_stream(side)
        with torch.cuda.stream(side):
            z1 = self.multihead_attn1(src1, src2, pos_src1, pos_src2, kpm2, True, False, p)
            out1 = self.norm13(_ffn(self.norm12(z1), self.linear11, self.linear12, p))
        z2 = self.multihead_attn2(src2, src1, pos_src2, pos_src1, kpm1, True, False, p)
        out2 = self.norm23(_ffn(self.norm22(z2), self.linear21, self.linear22, p))
        main.wait_stream(side)
        out1.record_stream(main)
        return out1, out2


class DecoderCFALayer(nn.Module):
    """search <- template CFA + FFN (transt_fusion.py:116-154)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        if activation != "relu":
            raise RuntimeError("activation should be relu, not %s." % activation)
        self.multihead_attn = _AttentionParams(d_model, nhead, dropout)
        self.linear1 = _LinearParams(d_model, dim_feedforward)
        self.linear2 = _LinearParams(dim_feedforward, d_model)
        self.norm1 = _NormParams(d_model)
        self.norm2 = _NormParams(d_model)
        self.dropout = dropout

    def forward(self, tgt, memory, memory_kpm=None, pos_enc=None, pos_dec=None):
        p = self.dropout if self.training else 0.0
        tgt = self.norm1(self.multihead_attn(tgt, memory, pos_dec, pos_enc, memory_kpm, True, False, p))
        return self.norm2(_ffn(tgt, self.linear1, self.linear2, p))


class Encoder(nn.Module):
    def __init__(self, make_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])
        self.num_layers = num_layers

    def forward(self, src1, src2, kpm1, kpm2, pos1, pos2):
        for layer in self.layers:
            src1, src2 = layer(src1, src2, kpm1, kpm2, pos1, pos2)
        return src1, src2


class Decoder(nn.Module):
    def __init__(self, layer, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([layer])
        self.norm = norm

    def forward(self, tgt, memory, memory_kpm, pos_enc, pos_dec):
        out = tgt
        for layer in self.layers:
            out = layer(out, memory, memory_kpm, pos_enc, pos_dec)
        return self.norm(out) if self.norm is not None else out


def _tokens(t):
    """[B,C,H,W] -> [B, H*W, C] contiguous."""
    return t.flatten(2).transpose(1, 2).contiguous()


def _pos_rows(pos):
    """Positional encoding [B,C,H,W] -> [R,C] rows, R = L when shared by the batch, else B*L."""
    if pos.shape[0] == 1 or pos.stride(0) == 0:
        return _tokens(pos[:1])[0]
    return _tokens(pos).reshape(-1, pos.shape[1])


def _kpm(mask):
    if mask is None:
        return None
    return mask.flatten(1).to(torch.uint8).contiguous()


class FeatureFusionNetwork(nn.Module):

    def __init__(self, d_model=512, nhead=8, num_featurefusion_layers=4,
                 dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        if d_model // nhead != 32:
            raise RuntimeError("spot_amd: the attention kernels are built for head_dim 32 (d_model/nhead = %d)"
                               % (d_model // nhead))
        self.encoder = Encoder(lambda: FeatureFusionLayer(d_model, nhead, dim_feedforward, dropout, activation),
                               num_featurefusion_layers)
        self.decoder = Decoder(DecoderCFALayer(d_model, nhead, dim_feedforward, dropout, activation),
                               _NormParams(d_model))
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead
        self.dropout = dropout

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def set_attention_dtype(self, dtype):
        """"f32": exact fp32 attention products; "bf16": bf16 products with fp32 softmax and accumulation."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("attention dtype must be 'f32' or 'bf16', not %r" % (dtype,))
        for m in self.modules():
            if isinstance(m, _AttentionParams):
                m.attn_bf16 = dtype == "bf16"
        return self

    def forward_tokens(self, src_temp, src_search, pos_temp, pos_search, kpm_temp=None, kpm_search=None):
        """Token-level entry: src [B,L,C], pos [R,C]; returns hs [B, Ls, C]."""
        mem_t, mem_s = self.encoder(src_temp, src_search, kpm_temp, kpm_search, pos_temp, pos_search)
        return self.decoder(mem_s, mem_t, kpm_temp, pos_temp, pos_search)

    def forward(self, src_temp, mask_temp, src_search, mask_search, pos_temp, pos_search):
        hs = self.forward_tokens(_tokens(src_temp), _tokens(src_search), _pos_rows(pos_temp), _pos_rows(pos_search),
                                 _kpm(mask_temp), _kpm(mask_search))
        return hs.unsqueeze(0)


def build_featurefusion_network(settings):
    net = FeatureFusionNetwork(
        d_model=settings.hidden_dim,
        dropout=settings.dropout,
        nhead=settings.nheads,
        dim_feedforward=settings.dim_feedforward,
        num_featurefusion_layers=settings.featurefusion_layers,
    )
    return net.set_attention_dtype(getattr(settings, "attention_dtype", "f32"))

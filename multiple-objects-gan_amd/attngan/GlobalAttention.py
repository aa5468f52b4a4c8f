"""Word-level attention over image regions (mirror of code/coco/attngan/GlobalAttention.py).

GlobalAttentionGeneral keeps the reference's stateful applyMask()/forward() surface and its
mask-indexing behaviour (GlobalAttention.py:104-108; cfg.ATT_MASK_MODE=1 selects the
per-sample mask instead).  func_attention is the DAMSM attention used by words_loss.
"""
import torch
import torch.nn as nn

from ..hip import ops
from .miscc.config import cfg


def conv1x1(in_planes, out_planes):
    from .model_base import HipConv2d
    return HipConv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=False)


def func_attention(query, context, gamma1):
    """GlobalAttention.py:31-69.  query (B,ndf,T), context (B,ndf,ih,iw) ->
    weightedContext (B,ndf,T), attn (B,T,ih,iw)."""
    B, T = query.size(0), query.size(2)
    ih, iw = context.size(2), context.size(3)
    S = ih * iw
    ctx = context.reshape(B, -1, S)
    attn = ops.bmm(ctx.transpose(1, 2), query)                   # B,S,T   Eq. (7)
    attn = ops.softmax(attn, 2)                                  # over the words, Eq. (8)
    attn = ops.softmax(attn.transpose(1, 2), 2, gamma1)          # over the regions, Eq. (9): B,T,S
    wc = ops.bmm(ctx, attn.transpose(1, 2))                      # B,ndf,T
    return wc, attn.reshape(B, T, ih, iw)


class GlobalAttentionGeneral(nn.Module):
    def __init__(self, idf, cdf):
        super(GlobalAttentionGeneral, self).__init__()
        self.conv_context = conv1x1(cdf, idf)
        self.sm = nn.Softmax(dim=1)
        self.mask = None

    def applyMask(self, mask):
        self.mask = mask  # batch x sourceL

    def forward(self, input, context):
        """input: B x idf x ih x iw (queryL = ih*iw); context: B x cdf x sourceL."""
        ih, iw = input.size(2), input.size(3)
        B, T = context.size(0), context.size(2)
        sourceT = self.conv_context(context.unsqueeze(3)).squeeze(3)        # B x idf x T
        wc, attn = ops.attention(input.reshape(B, -1, ih * iw), sourceT, self.mask, int(cfg.ATT_MASK_MODE))
        return wc.view(B, -1, ih, iw), attn.view(B, T, ih, iw)

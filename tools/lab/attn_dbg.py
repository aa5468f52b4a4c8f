import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
torch.manual_seed(0)
B, idf, Q, T = 3, 6, 16, 5
h = torch.randn(B, idf, Q); src = torch.randn(B, idf, T) * 0.3
for name, mask in (("nomask", None), ("mask", torch.zeros(B, T, dtype=torch.bool))):
    if mask is not None:
        for b in range(B): mask[b, max(1, T - 1 - b):] = True
    sc = torch.bmm(h.double().transpose(1, 2), src.double())            # B,Q,T
    if mask is not None:
        rows = (torch.arange(B * Q) % B)
        sc = sc.reshape(B * Q, T).masked_fill(mask[rows], -float("inf")).reshape(B, Q, T)
    att = torch.softmax(sc, 2).transpose(1, 2)                           # B,T,Q
    wc = torch.bmm(src.double(), att)
    wcd, attd = ops.attention(h.cuda(), src.cuda(), mask.cuda() if mask is not None else None, 0)
    print(name, "attn err", float((attd.cpu().double() - att).abs().max()), "wc err", float((wcd.cpu().double() - wc).abs().max()))
    print(" attn got b0 q0", attd[0, :, 0].cpu().tolist()); print(" attn ref b0 q0", att[0, :, 0].tolist())

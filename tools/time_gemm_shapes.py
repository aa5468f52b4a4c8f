"""Would im2col + plain GEMM beat the implicit GEMM on the small-spatial layers?  Times ops.bmm_raw on the equivalent
GEMM shapes (NT layout = both operands K-contiguous -> 16-byte loads) next to the conv calls they would replace."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (Cin, Cout, Hin, k, s): conv on a B=16 batch
for (Cin, Cout, H, k, s) in [(3072, 1536, 4, 3, 1), (1536, 768, 4, 3, 1), (1024, 768, 4, 3, 1), (1536, 3072, 8, 4, 2),
                             (768, 1536, 16, 4, 2), (384, 768, 32, 4, 2)]:
    B = 16
    OH = (H + 2 - k) // s + 1
    M, N, K = Cout, B * OH * OH, Cin * k * k
    gf = 2.0 * M * N * K / 1e9
    W = torch.randn(M, K, device=dev) * 0.02
    Bt = torch.randn(N, K, device=dev)          # im2col, patch vectors contiguous
    C = torch.empty(M, N, device=dev)
    dY = torch.randn(M, N, device=dev)
    dBt = torch.empty(N, K, device=dev)
    dW = torch.zeros(M, K, device=dev)
    tf = t(lambda: ops.bmm_raw(W.unsqueeze(0), Bt.t().unsqueeze(0), C.unsqueeze(0)))
    # dgrad: dBt[N][K] = dY^T[N][M] @ W[M][K]   (A = dY^T: m-contiguous..., B = W: n(K)-contiguous)
    dYt = dY.t().contiguous()
    td = t(lambda: ops.bmm_raw(dYt.unsqueeze(0), W.unsqueeze(0), dBt.unsqueeze(0)))
    # wgrad: dW[M][K] += dY[M][N] @ Bt[N][K]
    tw = t(lambda: ops.bmm_raw(dY.unsqueeze(0), Bt.unsqueeze(0), dW.unsqueeze(0), accumulate=True))
    x = torch.randn(B, Cin, H, H, device=dev); w4 = W.view(Cout, Cin, k, k)
    y = ops.conv2d_forward(x, w4, s, 1, 1, 0); dy4 = torch.randn_like(y); g4 = torch.zeros_like(w4)
    cf = t(lambda: ops.conv2d_forward(x, w4, s, 1, 1, 0))
    cd = t(lambda: ops.conv2d_dgrad(dy4, w4, x.shape, s, 1, 1, 0))
    cw = t(lambda: ops.conv2d_wgrad(dy4, x, w4.shape, s, 1, 1, 0, out=g4, accumulate=True))
    print("%4d->%4d %2dx%-2d k%d s%d  M %5d N %5d K %6d %6.1f GF | gemm fwd %5.1f dgrad %5.1f wgrad %5.1f TF | conv fwd %5.1f dgrad %5.1f wgrad %5.1f TF | im2col %5.1f MB"
          % (Cin, Cout, H, H, k, s, M, N, K, gf, gf / tf, gf / td, gf / tw, gf / cf, gf / cd, gf / cw, N * K * 4 / 1e6), flush=True)

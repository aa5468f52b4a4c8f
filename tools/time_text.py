"""The text encoder (RNN_ENCODER: Embedding + bi-LSTM over packed captions, eval, no grad) alone: device time and host time per call.
python tools/time_text.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.attngan import model
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
set_coco_train_defaults()
dev = torch.device("cuda")
torch.manual_seed(0)
enc = model.RNN_ENCODER(27297, nhidden=cfg.TEXT.EMBEDDING_DIM).to(dev).eval()
B, T = 16, cfg.TEXT.WORDS_NUM
lens = sorted([T] + [int(v) for v in torch.randint(5, T + 1, (B - 1,))], reverse=True)
cap = torch.zeros(B, T, dtype=torch.int64)
for i, n in enumerate(lens):
    cap[i, :n] = torch.randint(1, 27297, (n,))
cap = cap.to(dev)
lens_t = torch.tensor(lens)
def run():
    with torch.no_grad():
        return enc(cap, lens_t, enc.init_hidden(B))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
t0 = time.perf_counter(); e0.record()
for _ in range(50): run()
e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
print("text encoder B=%d T=%d: device %.1f us per call, host %.1f us per call" % (B, T, e0.elapsed_time(e1) * 20, (t1 - t0) * 2e4))

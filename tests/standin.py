"""Stand-in image encoder used where the fixtures need *an* image encoder with fixed weights
(torchvision's Inception-v3 cannot be imported in the build container: SURVEY.md §8(c)).
Plain torch ops, test scaffolding only -- it stands for CNN_ENCODER's output contract
(model.py:252-313): (B,3,256,256) -> regions (B,nef,17,17), code (B,nef)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class StandInEncoder(nn.Module):
    def __init__(self, nef):
        super().__init__()
        self.emb_features = nn.Conv2d(3, nef, 1, bias=False)
        self.emb_cnn_code = nn.Linear(12, nef)

    def forward(self, x):
        f = F.adaptive_avg_pool2d(x, 17)
        feat = self.emb_features(torch.tanh(f * 3.0))
        code = self.emb_cnn_code(F.adaptive_avg_pool2d(x, 2).flatten(1))
        return feat, code

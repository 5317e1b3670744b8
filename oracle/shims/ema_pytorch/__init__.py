"""Import-only stub for `ema_pytorch` (off the hot path)."""
import copy
from torch import nn
class EMA(nn.Module):
    def __init__(self, model, beta=0.99, forward_method_names=(), **kw):
        super().__init__()
        self.online_model = [model]
        self.ema_model = copy.deepcopy(model)
        self.beta = beta
    def update(self):
        import torch
        with torch.no_grad():
            for pe, po in zip(self.ema_model.parameters(), self.online_model[0].parameters()):
                pe.lerp_(po, 1. - self.beta)
    def forward(self, *a, **k):
        return self.ema_model(*a, **k)

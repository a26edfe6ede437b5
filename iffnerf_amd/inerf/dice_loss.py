"""inerf/dice_loss.py:8-80 of the reference: soft Dice loss on sigmoid(logits).  The reference writes the gradient by hand
for stability; it is the analytic derivative of the same expression, so plain autograd on the forward formula gives the
same numbers."""
import torch


class SoftDiceLossV2(torch.nn.Module):
    def __init__(self, p=1, smooth=1):
        super().__init__()
        self.p, self.smooth = p, smooth

    def forward(self, logits, labels):
        logits, labels = logits.reshape(1, -1).float(), labels.reshape(1, -1).float()
        probs = torch.sigmoid(logits)
        numer = 2 * (probs * labels).sum(dim=1) + self.smooth
        denor = (probs.pow(self.p) + labels.pow(self.p)).sum(dim=1) + self.smooth
        return 1.0 - numer / denor

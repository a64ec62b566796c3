"""Loss counterparts the train/eval loops dispatch on (reference utils/custom_loss_functions.py).

`Masked_L2_loss` (:10-46) is the reference's default loss (utils/argument_parser.py:36-39): two masked MSE means over
(N, 4).  On HIP tensors it is `pfn_masked_l2_loss` (loss + gradient in two launches); on host tensors
(metrics on the CPU) plain torch ops.  The physics losses
(`PowerImbalance`, `MixedMSEPoweImbalance`, :99-306) are row N4 of SURVEY.md 8(f) -- outside the hot path of this
round -- and exist here only as named placeholders so that `isinstance` dispatch keeps the reference's order.
"""
import torch
import torch.nn as nn


class Masked_L2_loss(nn.Module):
    """mean((out-y)^2 over predicted entries) [+ regcoeff * mean(... over the given entries)]."""

    def __init__(self, regularize=True, regcoeff=1):
        super().__init__()
        self.criterion = nn.MSELoss(reduction="mean")
        self.regularize = regularize
        self.regcoeff = regcoeff

    def forward(self, output, target, mask):
        if output.is_cuda:
            from ..loss import masked_l2_loss
            return masked_l2_loss(output, target, mask, self.regularize, self.regcoeff)
        sel = mask.type(torch.bool)
        loss = self.criterion(torch.masked_select(output, sel), torch.masked_select(target, sel))
        if self.regularize:
            rest = (1 - mask).type(torch.bool)
            loss = loss + self.regcoeff * self.criterion(torch.masked_select(output, rest), torch.masked_select(target, rest))
        return loss

    @staticmethod
    def unit_grad(loss):
        from ..loss import unit_grad
        return unit_grad(loss)


class PowerImbalance(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("PowerImbalance (utils/custom_loss_functions.py:99-286) is out of this round's scope "
                                  "(SURVEY.md 8f, row N4)")


class MixedMSEPoweImbalance(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("MixedMSEPoweImbalance (utils/custom_loss_functions.py:289-306) is out of this round's "
                                  "scope (SURVEY.md 8f, row N4)")

"""Loss counterparts the train/eval loops dispatch on (reference utils/custom_loss_functions.py).

`Masked_L2_loss` (:10-46) is the reference's default loss (utils/argument_parser.py:36-39): two masked MSE means over
(N, 4), computed by `pfn_masked_l2_loss` (loss + gradient in two launches).  Like everything else in
this package it has NO CPU path: host tensors raise RuntimeError (the CPU restatement lives in oracle/ref_cpu.py, for tests
only).  The physics losses (`PowerImbalance`, `MixedMSEPoweImbalance`, :99-306; SURVEY.md
8f row N4) run as `pfn_power_imbalance` (csrc/physics.hip).
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
        from ..loss import MASKED_L2_WS_FLOATS, _Workspace
        self._ws = _Workspace(MASKED_L2_WS_FLOATS)
        self._tail_ws = _Workspace(2052)     # pfn_mpn_backward_masked_l2: 2048 partials + the arrival counter

    def attach(self, model, target, mask):
        """See loss.MSELoss.attach: the same promise for this loss -- the model's backward pass forms the output rows, the loss
        and its gradient in its first launch (pfn_mpn_backward_masked_l2), three launches fewer per step."""
        from ..loss import masked_l2_attach
        masked_l2_attach(model, target, mask, self.regularize, self.regcoeff, self._tail_ws)

    def forward(self, output, target, mask):
        from ..loss import masked_l2_loss
        return masked_l2_loss(output, target, mask, self.regularize, self.regcoeff, self._ws)   # raises on host tensors

    @staticmethod
    def unit_grad(loss):
        from ..loss import unit_grad
        return unit_grad(loss)


def _per_feature_terms(error, mask):
    """Shared tail of the two evaluation metrics below (reference :70-79, :88-97): masked column means, their overall and
    their balanced mean, and the four named features."""
    cnt = mask.sum(dim=0).clamp(min=1e-6)
    per_feature = (error * mask.float()).sum(dim=0) / cnt
    terms = {"total": (per_feature * cnt).sum() / mask.sum().clamp(min=1e-6), "balanced total": per_feature.mean()}
    terms.update(zip(("vm", "va", "p", "q"), per_feature[:4]))
    return terms


class MaskedL2V2(nn.Module):
    """Evaluation metric of test.py (:113-118): per-feature masked MSE as a dict of terms (reference :48-79).  A metric on
    (N, 4) outputs, outside the training path: plain tensor ops on whatever device the tensors live on."""

    def __init__(self, regularize=False, regcoeff=1):   # both unused, as in the reference
        super().__init__()

    def forward(self, output, target, mask):
        return _per_feature_terms((output - target) ** 2, mask)


class MaskedL1(nn.Module):
    """Per-feature masked mean absolute error as a dict of terms (reference :82-97)."""

    def forward(self, output, target, mask):
        return _per_feature_terms((output - target).abs(), mask)


class PowerImbalance(nn.Module):
    """Power-imbalance loss (reference :99-286): mean over buses of dP^2 + dQ^2, where dP/dQ are the mismatch between the
    predicted injections and the branch flows implied by the predicted voltages.  Same constructor and forward signature
    as the reference class; on HIP tensors the two CSR walks of `pfn_power_imbalance` (csrc/physics.hip; SURVEY.md 8f row
    N4), the adjacency cached per `edge_index` tensor like the model's."""
    base_sn = 100        # kva
    base_voltage = 345   # kv
    base_ohm = 1190.25   # v**2/sn

    def __init__(self, xymean, xystd, edgemean, edgestd, reduction="mean"):
        super().__init__()
        self.xymean = xymean[0:1] if xymean.shape[0] > 1 else xymean
        self.xystd = xystd[0:1] if xystd.shape[0] > 1 else xystd
        self.edgemean, self.edgestd = edgemean, edgestd
        self._stats = None
        self._graphs = None
        # True: every forward rebuilds the adjacency from the batch's edge_index on the device, no host sync (like
        # MaskEmbdMultiMPN.dynamic_topology: a captured training step whose topology changes per batch -- GraphedTrainStep sets
        # both; with the cache a replayed step would keep walking the CAPTURE-time topology)
        self.dynamic_topology = False
        from ..loss import POWER_IMBALANCE_WS_FLOATS, _Workspace
        self._ws = _Workspace(POWER_IMBALANCE_WS_FLOATS)

    def forward(self, x, edge_index, edge_attr):
        if not x.is_cuda:
            raise RuntimeError("PowerImbalance: there is no CPU path (move the tensors to the HIP device)")
        from ..loss import power_imbalance
        from ..networks.MPN import _GraphCache
        if self._stats is None:      # 12 host floats, read once (the statistics are constants of the dataset)
            self._stats = [float(v) for t in (self.xymean, self.xystd, self.edgemean, self.edgestd) for v in t.reshape(-1).tolist()]
            if len(self._stats) != 12:
                raise RuntimeError("PowerImbalance: expected 4 node and 2 branch statistics")
        if self._graphs is None:
            self._graphs = _GraphCache()
        graph = self._graphs.get(edge_index, x.shape[0], -1, rebuild=self.dynamic_topology)
        return power_imbalance(x, graph, edge_attr, self._stats, self._ws)

    @staticmethod
    def unit_grad(loss):
        from ..loss import unit_grad
        return unit_grad(loss)


class MixedMSEPoweImbalance(nn.Module):
    """alpha * MSE(x, y) + (1 - alpha) * 0.020 * PowerImbalance(x) (reference :289-306)."""

    def __init__(self, xymean, xystd, edgemean, edgestd, alpha=0.5, reduction="mean"):
        super().__init__()
        assert alpha <= 1. and alpha >= 0
        from ..loss import MSELoss
        self.power_imbalance = PowerImbalance(xymean, xystd, edgemean, edgestd, reduction)
        self.mse_loss_fn = MSELoss()
        self.alpha = alpha

    def forward(self, x, edge_index, edge_attr, y):
        power_imb_loss = self.power_imbalance(x, edge_index, edge_attr)
        mse_loss = self.mse_loss_fn(x, y)
        return self.alpha * mse_loss + (1 - self.alpha) * 0.020 * power_imb_loss

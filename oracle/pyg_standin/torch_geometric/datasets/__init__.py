"""TEST INFRASTRUCTURE ONLY -- `Planetoid` only has to be an importable name (datasets/PowerFlowData.py:15 imports it and
never uses it)."""


class Planetoid:  # pragma: no cover
    pass

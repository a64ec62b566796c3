from .MPN import EdgeAggregation, GraphCSR, MaskEmbdMultiMPN, TAGConv  # noqa: F401

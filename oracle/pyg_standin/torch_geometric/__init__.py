"""TEST INFRASTRUCTURE ONLY -- stand-in for the five torch_geometric symbols that
/root/reference/networks/MPN.py imports (networks/MPN.py:3-4).

torch_geometric (PyG, version unpinned by the reference: it ships no requirements file)
is not installed in this image and cannot be installed (no network).  This package
restates PyG's *published* algorithm for exactly the primitives the hot path reaches
(MessagePassing.propagate with aggr='add', TAGConv, utils.degree) so that the reference's
own networks/MPN.py can be imported UNMODIFIED by oracle/make_goldens.py and executed on
CPU to produce the golden vectors in tests/golden/.

Parity status: everything in networks/MPN.py executes as written; the PyG primitives
underneath are a restatement -> parity at the PyG boundary is UNPINNED against PyG itself
(the reference has no tests or golden vectors at that boundary).  They are cross-checked
analytically instead (dense sum_k A_hat^k X W_k^T + b; explicit per-edge loops) in
tests/test_oracle.py.

Never imported by the product (poweflownet_amd/), never shipped to the GPU box as
anything but dead files.
"""

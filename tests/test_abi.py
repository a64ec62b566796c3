"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol
include/pfn_hip.h declares; pure-host entry points answer; the product has no CPU fallback."""
import ctypes as C
import os
import re

import pytest
import torch

from poweflownet_amd import _lib as L
from poweflownet_amd.networks.MPN import EdgeAggregation, MaskEmbdMultiMPN, TAGConv
from poweflownet_amd.synth import make_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pfn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    declared = _declared_symbols()
    assert set(declared) == set(L.SYMBOLS), (declared, L.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pfn_abi_version() == L.ABI_VERSION == 8


def test_host_only_entry_points():
    lib = L.load()
    assert lib.pfn_padded_ld(129) == 132 and lib.pfn_padded_ld(4) == 4
    cfg = L.MpnConfig(4, 2, 4, 129, 4, 3, 0.2, 1)
    assert lib.pfn_mpn_num_params(C.byref(cfg)) == 35
    cfgw = L.MpnConfig(4, 2, 4, 129, 6, 6, 0.2, 1)
    assert lib.pfn_mpn_num_params(C.byref(cfgw)) == 68
    assert lib.pfn_graph_workspace_bytes(15104, 23808) > 4 * (2 * 15104 + 8 * 23808)
    ws = lib.pfn_mpn_workspace_bytes(C.byref(cfg), 15104, 23808)
    assert ws > 15104 * 132 * 4 * 20
    cfgb = L.MpnConfig(4, 2, 4, 129, 4, 3, 0.2, 1, 1)      # need_backward = 1: the backward pass's buffers come on top
    assert lib.pfn_mpn_workspace_bytes(C.byref(cfgb), 15104, 23808) > 1.9 * ws
    bad = L.MpnConfig(4, 2, 4, 129, 1, 3, 0.2, 1)          # L == 1 is rejected
    assert lib.pfn_mpn_workspace_bytes(C.byref(bad), 10, 10) == 0


def test_argument_errors_are_reported_not_fatal():
    lib = L.load()
    rc = lib.pfn_graph_build(None, 5, 4, -1, None, 0, None)
    assert rc == -1 and b"null" in lib.pfn_last_error()


def test_state_dict_contract_matches_reference_keys():
    from tests.util import load, params_from
    m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.2)
    want = params_from(load("g4_params_standard"))
    assert sorted(m.state_dict().keys()) == sorted(want.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(want[k].shape), k
    m.load_state_dict(want)                                 # a reference checkpoint loads as-is
    assert sum(p.numel() for p in m.parameters()) == 354_500
    assert len(m._ordered_params()) == 35


@pytest.mark.parametrize("tag", ["MPN", "SkipMPN", "MaskEmbdMPN", "MultiMPN", "MaskEmbdMultiMPN_NoMP"])
def test_sibling_models_state_dict_contract(tag):
    """same constructor and state_dict keys / shapes as the reference classes (fixture G12 holds the
    reference instances' parameters), the stale-width assert kept, no CPU path."""
    import poweflownet_amd.networks.MPN as M
    from tests.util import load
    fx = load("g12_sibling_models")
    f, o, h, L_, K = (int(v) for v in fx[f"{tag}.cfg"])
    m = getattr(M, tag)(nfeature_dim=f, efeature_dim=2, output_dim=o, hidden_dim=h, n_gnn_layers=L_, K=K, dropout_rate=0.0)
    want = {k[len(tag) + 7:]: v for k, v in fx.items() if k.startswith(f"{tag}.param.")}
    assert sorted(m.state_dict().keys()) == sorted(want.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(want[k].shape), k
    m.load_state_dict(want)
    with pytest.raises(AssertionError):                     # networks/MPN.py:194 etc.: a 4-wide x is refused first
        m(make_batch("14", 1))
    from poweflownet_amd.data import Data
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(Data(x=fx[f"{tag}.x"], edge_index=fx["edge_index"], edge_attr=fx[f"{tag}.edge_attr"]))


def test_no_cpu_fallback():
    m = MaskEmbdMultiMPN(4, 2, 4, 8, 2, 3, 0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(make_batch("14", 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EdgeAggregation(4, 2, 8, 8)(torch.zeros(3, 4), torch.zeros(2, 2, dtype=torch.long), torch.zeros(2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TAGConv(4, 4, 2)(torch.zeros(3, 4), torch.zeros(2, 2, dtype=torch.long))


def test_constructor_surface():
    with pytest.raises(ValueError):
        MaskEmbdMultiMPN(4, 2, 4, 8, 1, 3, 0.0)
    m = MaskEmbdMultiMPN(nfeature_dim=4, efeature_dim=2, output_dim=4, hidden_dim=8, n_gnn_layers=3, K=2, dropout_rate=0.1)
    assert [type(l).__name__ for l in m.layers] == ["EdgeAggregation", "TAGConv"] * 2 + ["EdgeAggregation"]
    for attr in ("nfeature_dim", "efeature_dim", "output_dim", "hidden_dim", "n_gnn_layers", "K", "dropout_rate"):
        assert hasattr(m, attr)
    with pytest.raises(AssertionError):                     # networks/MPN.py:528
        bad = make_batch("14", 1)
        bad.x = torch.zeros(14, 6)
        m(bad)


def test_graft_entry_build_runs():
    """The driver's "does it build" hook: compiles every HIP source (incremental make) and checks the library's ABI version
    against the Python mirror's."""
    import __graft_entry__
    __graft_entry__.build()


def test_every_environment_switch_of_the_library_is_documented():
    """The library's only getenv is pfn::diag_env (csrc/graph.hip); every name passed to it appears in include/pfn_hip.h's
    "Diagnostic environment switches" list, and the list names nothing that does not exist."""
    import glob
    used = set()
    for f in glob.glob(os.path.join(ROOT, "poweflownet_amd", "csrc", "*.h*")):
        text = open(f).read()
        used |= set(re.findall(r'diag_env\("([A-Z0-9_]+)"\)', text))
        if not f.endswith("graph.hip"):
            assert "getenv(" not in text, f
    assert open(os.path.join(ROOT, "poweflownet_amd", "csrc", "graph.hip")).read().count("getenv(") == 1
    header = open(os.path.join(ROOT, "include", "pfn_hip.h")).read()
    documented = set(re.findall(r"^ \*   (PFN_[A-Z0-9_]+)", header, flags=re.M))
    assert used == documented, (sorted(used - documented), sorted(documented - used))


def test_isa_of_the_async_operand_fragments_is_hazard_free():
    """seg_tile.hpp seg_load_a_async: 17 inline-asm loads whose registers hold garbage until a hand-placed wait.  The compiler does
    not know; tools/check_async_fragments.py reads the ISA of the objects that are linked (csrc/Makefile keeps it, -save-temps; no GPU needed): every fragment
    has its wait sequence, and nothing names a fragment register between its load and its wait."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_async_fragments.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("2 async fragment(s) checked, 0 problem(s)") == 2, r.stdout     # ea_seg.hip: fwd + bwd; seg_lin_hops.hip: <1> + <2>


def test_experiment_patches_still_apply_to_the_product_sources(tmp_path):
    """tools/ubench/*.patch.txt hold the timestamp / ablation instrumentation that was taken out of the product kernels (the product
    sources carry no experiment switches); every run_*_ts.sh / run_*_exp.sh applies them to a copy of csrc/ first.  A kernel change
    that moves their context breaks those tools silently -- so: they must apply cleanly to a copy of the current sources."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = tmp_path / "csrc"
    shutil.copytree(os.path.join(root, "poweflownet_amd", "csrc"), dst, ignore=shutil.ignore_patterns("*.o", "*.so"))
    r = subprocess.run(["bash", os.path.join(root, "tools", "ubench", "apply_experiments.sh"), str(dst)], capture_output=True, text=True)
    assert r.returncode == 0 and not list(dst.glob("*.rej")), r.stdout + r.stderr

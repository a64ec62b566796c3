"""Host-side mini PyG surface: collate rule (G7, analytic) and loader sharding."""
import torch

from oracle import ref_cpu
from poweflownet_amd.data import Batch, Data, DataLoader
from poweflownet_amd.synth import CASES, make_batch, make_dataset, make_topology
from tests.util import assert_close, load, params_from


def _graphs(fx):
    return [Data(**{k: fx[f"g{b}.{k}"] for k in ("x", "y", "bus_type", "pred_mask", "edge_index", "edge_attr")})
            for b in range(3)]


def test_g7_collate_matches_fixture():
    fx = load("g7_collate")
    big = Batch.from_data_list(_graphs(fx))
    for k in ("x", "y", "bus_type", "pred_mask", "edge_index", "edge_attr", "batch"):
        assert torch.equal(getattr(big, k), fx[f"big.{k}"]), k
    assert big.ptr.tolist() == fx["ptr"].tolist()
    assert big.num_graphs == 3


def test_g7_batch_equals_concat_of_singles():
    fx = load("g7_collate")
    m = ref_cpu.MaskEmbdMultiMPN(4, 2, 4, 8, 2, 3, 0.0).eval()
    m.load_state_dict(params_from(fx))
    big = Batch.from_data_list(_graphs(fx))
    assert_close(m(big), fx["batch_out"], 1e-6, "batch_out")
    assert_close(m(big), fx["singles_out"], 1e-6, "singles")


def test_data_len_to_and_immutability():
    d = _graphs(load("g7_collate"))[0]
    assert len(d) == 6                      # number of stored attributes (PyG Data.__len__)
    d2 = d.to("cpu")
    assert d2 is not d and torch.equal(d2.x, d.x)
    assert "edge_index" in d.keys()


def test_topology_is_connected_and_sized():
    for case in ("14", "118"):
        n, e = CASES[case]
        ei = make_topology(n, e, 0)
        assert ei.shape == (2, e) and int(ei.max()) == n - 1 and (ei[0] != ei[1]).all()
        seen, frontier = {0}, [0]
        adj = {}
        for s, t in ei.t().tolist():
            adj.setdefault(s, []).append(t); adj.setdefault(t, []).append(s)
        while frontier:
            v = frontier.pop()
            for u in adj.get(v, []):
                if u not in seen:
                    seen.add(u); frontier.append(u)
        assert len(seen) == n


def test_hub_topology_has_high_degree():
    ei = make_topology(6470, 9005, 0, hub_frac=0.2)
    deg = torch.bincount(ei.flatten(), minlength=6470)
    assert int(deg.max()) >= 64


def test_loader_shards_partition_global_batch():
    ds = make_dataset("14", 8)
    full = list(DataLoader(ds, batch_size=4))
    assert len(full) == 2 and full[0].x.shape[0] == 4 * 14
    r0 = list(DataLoader(ds, batch_size=4, shard=(0, 2)))
    r1 = list(DataLoader(ds, batch_size=4, shard=(1, 2)))
    x = torch.cat([r0[0].x.view(2, 14, 4), r1[0].x.view(2, 14, 4)], 0)
    want = full[0].x.view(4, 14, 4)[[0, 2, 1, 3]]
    assert torch.equal(x, want)


def test_make_batch_offsets():
    b = make_batch("14", 3)
    assert b.edge_index.shape == (2, 60) and int(b.edge_index[:, 20:40].min()) >= 14
    assert b.ptr.tolist() == [0, 14, 28, 42]

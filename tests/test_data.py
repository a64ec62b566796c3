"""Host-side mini PyG surface: collate rule (G7, analytic) and loader sharding."""
import pytest
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


@pytest.mark.parametrize("n,bs,world", [(5, 4, 2), (9, 4, 2), (11, 8, 4), (13, 6, 3), (7, 8, 2), (8, 4, 2)])
def test_loader_shards_are_equal_on_every_rank(n, bs, world):
    """Every rank yields the same number of batches with the same number of graphs each (a rank that skipped a step would
    leave the others waiting in the gradient all-reduce; unequal shards break mean-of-rank-means == global mean).
    Covers n % (bs * world)-style tails with fewer samples than ranks."""
    ds = make_dataset("14", n)
    per_rank = [list(DataLoader(ds, batch_size=bs, shard=(r, world))) for r in range(world)]
    lens = [len(DataLoader(ds, batch_size=bs, shard=(r, world))) for r in range(world)]
    assert len(set(lens)) == 1 and all(len(b) == lens[0] for b in per_rank)
    for step in range(lens[0]):
        sizes = {b[step].num_graphs for b in per_rank}
        assert len(sizes) == 1 and sizes.pop() >= 1
    tail = n % bs
    expect = n // bs + (1 if tail >= world else 0)
    assert lens[0] == expect
    with pytest.raises(ValueError):
        DataLoader(ds, batch_size=1, shard=(0, 2))


def test_make_batch_offsets():
    b = make_batch("14", 3)
    assert b.edge_index.shape == (2, 60) and int(b.edge_index[:, 20:40].min()) >= 14
    assert b.ptr.tolist() == [0, 14, 28, 42]


# ------------------------------------------------------------------------------ PowerFlowData
def _raw_dir(tmp_path, fx, case="7x"):
    import numpy as np
    (tmp_path / "raw").mkdir()
    np.save(tmp_path / "raw" / f"case{case}_edge_features.npy", fx["raw_edge"].numpy())
    np.save(tmp_path / "raw" / f"case{case}_node_features.npy", fx["raw_node"].numpy())
    return str(tmp_path)


def test_g9_powerflowdata_matches_reference_class(tmp_path):
    """Every per-sample field, the statistics and the keyword paths of the constructor against what the reference's own
    PowerFlowData produced from the same raw files (tests/golden/g9_powerflowdata.npz)."""
    from poweflownet_amd.datasets import PowerFlowData, denormalize
    fx = load("g9_powerflowdata")
    root = _raw_dir(tmp_path, fx)
    sets = {}
    for task in ("train", "val", "test"):
        ds = PowerFlowData(root=root, case="7x", split=[.5, .2, .3], task=task, normalize=True)
        sets[task] = ds
        assert len(ds) == int(fx[f"{task}.len"])
        for k in ("x", "y", "edge_attr"):
            got = torch.stack([getattr(ds[i], k) for i in range(len(ds))])
            assert_close(got, fx[f"{task}.{k}"], 1e-6, f"{task}.{k}")
        for k in ("bus_type", "pred_mask", "edge_index"):
            got = torch.stack([getattr(ds[i], k) for i in range(len(ds))])
            assert torch.equal(got, fx[f"{task}.{k}"]), f"{task}.{k}"
        for k in ("xymean", "xystd", "edgemean", "edgestd"):
            assert_close(getattr(ds, k), fx[f"{task}.{k}"], 1e-6, f"{task}.{k}")
    assert list(sets["train"].get_data_dimensions()) == fx["dims"].tolist()
    assert_close(torch.cat([t.reshape(-1) for t in sets["train"].get_data_means_stds()]), fx["train.means_stds"], 1e-6, "means_stds")
    tr = sets["train"]
    ds = PowerFlowData(root=root, case="7x", split=[.5, .2, .3], task="val", xymean=tr.xymean, xystd=tr.xystd,
                       edgemean=tr.edgemean, edgestd=tr.edgestd)
    assert_close(torch.stack([ds[i].x for i in range(len(ds))]), fx["val_trainstats.x"], 1e-6, "val with train stats")
    assert_close(torch.stack([ds[i].edge_attr for i in range(len(ds))]), fx["val_trainstats.edge_attr"], 1e-6, "val ea")
    ds = PowerFlowData(root=root, case="7x", split=[.5, .2, .3], task="test", normalize=False)
    assert_close(torch.stack([ds[i].x for i in range(len(ds))]), fx["test_raw.x"], 1e-7, "raw x")
    assert_close(torch.stack([ds[i].y for i in range(len(ds))]), fx["test_raw.y"], 1e-7, "raw y")
    assert_close(denormalize(tr[0].y, tr.xymean, tr.xystd), fx["denorm.y"], 1e-6, "denormalize")
    with pytest.raises(RuntimeError):       # fractions that do not cover the samples: torch.split refuses, as in the reference
        PowerFlowData(root=root, case="7x", split=[.5, .2, .2], task="train")


def test_powerflowdata_device_batches_equal_per_sample_collate(tmp_path):
    from poweflownet_amd.data import Batch, DataLoader
    from poweflownet_amd.datasets import PowerFlowData, random_bus_type
    fx = load("g9_powerflowdata")
    ds = PowerFlowData(root=_raw_dir(tmp_path, fx), case="7x", split=[.5, .2, .3], task="train")
    batches = list(DataLoader(ds, batch_size=2, shuffle=False))
    assert len(batches) == 3 and batches[-1].num_graphs == 1
    for bi, b in enumerate(batches):
        want = Batch.from_data_list([ds[i] for i in range(2 * bi, min(2 * bi + 2, len(ds)))])
        for k in ("x", "y", "bus_type", "pred_mask", "edge_index", "edge_attr", "batch", "ptr"):
            assert torch.equal(getattr(b, k), getattr(want, k)), k
    assert batches[0].edge_index is batches[1].edge_index          # static topology: one cached tensor per batch size
    # a transform switches to the per-sample path and is applied
    (tmp_path / "t").mkdir()
    ds_t = PowerFlowData(root=_raw_dir(tmp_path / "t", fx), case="7x", split=[.5, .2, .3], task="train", transform=random_bus_type)
    b = next(iter(DataLoader(ds_t, batch_size=5)))
    assert int(b.bus_type.max()) <= 1 and b.x.shape == (35, 4)
    # rank sharding of a global batch
    shards = [next(iter(DataLoader(ds, batch_size=4, shard=(r, 2)))) for r in range(2)]
    assert torch.equal(shards[0].x, Batch.from_data_list([ds[0], ds[2]]).x)
    assert torch.equal(shards[1].x, Batch.from_data_list([ds[1], ds[3]]).x)

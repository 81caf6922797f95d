"""Ingest: CSV / DataFrame -> TemporalGraph / PathData.  Parsing is host code; building the TemporalGraph
sorts on the GPU, so those cases carry the gpu marker."""
import io

import numpy as np
import pandas as pd
import pytest
import torch

import pathpyg_amd as pp


def test_read_csv_path_data(tmp_path):
    f = tmp_path / "paths.ngram"
    f.write_text("a,c,d,2.0\nb,c,e,1.5\na,c,1\n")
    paths = pp.io.read_csv_path_data(str(f), weight=True)
    assert paths.num_paths == 3
    assert paths.get_walk(1) == ("b", "c", "e")
    assert paths.data.dag_weight.tolist() == [2.0, 1.5, 1.0]
    assert paths.data.edge_index.tolist() == [[0, 1, 3, 4, 6], [1, 2, 4, 5, 7]]
    f.write_text("x;y\ny;z;x\n")
    paths = pp.io.read_csv_path_data(str(f), weight=False, sep=";")
    assert paths.data.dag_weight.tolist() == [1.0, 1.0] and paths.get_walk(1) == ("y", "z", "x")


@pytest.mark.gpu
def test_df_and_csv_roundtrip_temporal_graph(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    rng = np.random.default_rng(0)
    names = np.array([f"n{i:02d}" for i in range(30)])
    m = 5000
    df = pd.DataFrame({"v": names[rng.integers(0, 30, m)], "w": names[rng.integers(0, 30, m)], "t": rng.integers(0, 2000, m),
                       "weight": rng.random(m).round(3), "label": rng.choice(["x", "y"], m)})
    g = pp.io.df_to_temporal_graph(df, multiedges=True, device="cuda")
    assert g.n == 30 and g.data.num_edges == m and g.data.time.dtype == torch.int64
    order = np.argsort(df["t"].values, kind="stable")                 # stable event order
    assert g.data.time.cpu().tolist() == df["t"].values[order].tolist()
    assert g.mapping.to_ids(g.data.edge_index[0].cpu()).tolist() == df["v"].values[order].tolist()
    assert torch.allclose(g.data.edge_weight.cpu().double(), torch.tensor(df["weight"].values[order]))
    assert g.data.edge_label.tolist() == df["label"].values[order].tolist()
    dedup = pp.io.df_to_temporal_graph(df, device="cuda")
    assert dedup.data.num_edges == len(df.drop_duplicates(subset=["v", "w", "t"]))
    # csv round trip, header-less variant and string timestamps
    path = tmp_path / "g.csv"
    pp.io.write_csv(g, path_or_buf=str(path))
    back = pp.io.read_csv_temporal_graph(str(path), multiedges=True, device="cuda")
    assert torch.equal(back.data.edge_index, g.data.edge_index) and torch.equal(back.data.time, g.data.time)
    raw = pd.read_csv(io.StringIO("a,b,2020-01-01 00:00:05\nb,c,2020-01-01 00:00:01\nc,a,2020-01-01 00:00:09\n"), header=None)
    tg = pp.io.df_to_temporal_graph(raw, time_rescale=10 ** 9, device="cuda")
    assert tg.data.time.tolist() == [0, 4, 8] and tg.temporal_edges[0] == ("b", "c", 0)
    from oracle import lift as ol
    want = ol.temporal_lift_sorted(tg.data.edge_index.cpu(), tg.data.time.cpu(), 8, 3)
    assert torch.equal(pp.algorithms.lift_order_temporal(tg, delta=8).cpu(), want) and want.tolist() == [[0], [2]]
    m2 = pp.MultiOrderModel.from_temporal_graph(tg, delta=5, max_order=2)
    assert m2.layers[2].n == 3

"""Ingest of time-stamped edge lists and n-gram path files, reference ``pathpyG.io.pandas``:
``df_to_temporal_graph`` (:318-397), ``temporal_graph_to_df`` (:431-470), ``read_csv_temporal_graph`` (:511-546),
``write_csv`` (:548-570), ``read_csv_path_data`` (:572-599), and their static-graph counterparts ``df_to_graph`` (:109-181),
``add_node_attributes`` (:183-235), ``add_edge_attributes`` (:237-316), ``graph_to_df`` (:399-429), ``read_csv_graph`` (:472-509).

Same column conventions as the reference (``v``, ``w``, ``t`` + edge attributes; header-less frames take the first three
columns).  Node IDs are mapped to indices with ONE vectorised ``np.unique(..., return_inverse=True)`` instead of a Python
dictionary lookup per endpoint, and the event sort happens on the GPU inside :class:`TemporalGraph`.
"""
from __future__ import annotations

import ast
import csv
from typing import Any, Optional

import numpy as np
import pandas as pd
import torch

from ..core.graph import Graph
from ..core.index_map import IndexMap
from ..core.path_data import PathData
from ..core.temporal_graph import TemporalGraph
from ..data import Data


def _parse_timestamps(df: pd.DataFrame, timestamp_format: str, time_rescale: int) -> pd.Series:
    t = df["t"]
    if pd.api.types.is_string_dtype(t) or pd.api.types.is_datetime64_any_dtype(t):
        if not pd.api.types.is_datetime64_any_dtype(t):
            t = pd.to_datetime(t, format=timestamp_format)
        t = t.astype("int64") // time_rescale
        return t - t.min()                                   # seconds (or ns / rescale) since the first event
    if t.dtype == "int64" or t.dtype == "float64":
        return t // time_rescale
    raise ValueError(f"Column `t` must be of type `object`, `int64`, `float64`, or a datetime type. Found {t.dtype} instead.")


def _column_to_attribute(values: np.ndarray, device):
    """Numeric columns -> tensors, strings holding numbers / literals -> tensors, other strings -> NumPy string arrays."""
    if values.dtype.kind in "iufb":
        return torch.tensor(values, device=device)
    first = values[0]
    if isinstance(first, str):
        try:
            parsed = [ast.literal_eval(x) for x in values]
            return torch.tensor(parsed, device=device)
        except (ValueError, SyntaxError, TypeError):
            return np.asarray(values).astype(str)
    if isinstance(first, (list, tuple, np.ndarray)):
        return torch.tensor(np.asarray([np.asarray(x) for x in values]), device=device)
    raise ValueError(f"Unsupported data type for attribute column: {type(first)}")


def df_to_temporal_graph(df: pd.DataFrame, multiedges: bool = False, timestamp_format="%Y-%m-%d %H:%M:%S", time_rescale=1,
                         num_nodes: int | None = None, device: Optional[torch.device] = None) -> TemporalGraph:
    """Temporal graph from a DataFrame with columns ``v``, ``w``, ``t`` (+ edge attributes); one row = one event."""
    df = df.copy()
    if all(isinstance(c, (int, np.integer)) for c in df.columns.values.tolist()):
        df.columns = ["v", "w", "t"] + [f"edge_attr_{i - 2}" for i in range(3, len(df.columns))]
    df["t"] = _parse_timestamps(df, timestamp_format, time_rescale)
    if not multiedges:
        df = df.drop_duplicates(subset=["v", "w", "t"])
    endpoints = df[["v", "w"]].values
    ids, inverse = np.unique(endpoints, return_inverse=True)
    mapping = IndexMap(ids)
    edge_index = torch.from_numpy(inverse.reshape(endpoints.shape).T.astype(np.int64)).contiguous()
    data = Data(edge_index=edge_index if device is None else edge_index.to(device),
                time=torch.tensor(df["t"].values, device=device),
                num_nodes=num_nodes if num_nodes is not None else len(ids))
    for col in df.columns:
        if col not in ("v", "w", "t"):
            data[col if col.startswith("edge_") else "edge_" + col] = _column_to_attribute(df[col].values, device)
    return TemporalGraph(data=data, mapping=mapping)


def df_to_graph(df: pd.DataFrame, is_undirected: bool = False, multiedges: bool = False, num_nodes: int | None = None,
                device: Optional[torch.device] = None) -> Graph:
    """Graph from a DataFrame with columns ``v``, ``w`` (+ edge attributes; header-less frames: the first two columns); duplicate
    (v, w) rows are dropped unless ``multiedges``; ``is_undirected`` adds every edge in the opposite direction."""
    df = df.copy()
    if all(isinstance(c, (int, np.integer)) for c in df.columns.values.tolist()):
        df.columns = ["v", "w"] + [f"edge_attr_{i - 2}" for i in range(2, len(df.columns))]
    if not multiedges and df[["v", "w"]].duplicated().any():
        df = df.drop_duplicates(subset=["v", "w"])
    endpoints = df[["v", "w"]].values
    ids, inverse = np.unique(endpoints, return_inverse=True)
    mapping = IndexMap(ids)
    edge_index = torch.from_numpy(inverse.reshape(endpoints.shape).T.astype(np.int64)).contiguous()
    data = Data(edge_index=edge_index if device is None else edge_index.to(device),
                num_nodes=num_nodes if num_nodes is not None else len(ids))
    for col in df.columns:
        if col not in ("v", "w"):
            data[col if col.startswith("edge_") else "edge_" + col] = _column_to_attribute(df[col].values, device)
    g = Graph(data=data, mapping=mapping)
    return g.to_undirected() if is_undirected else g


def _store_column(data: Data, name: str, values: np.ndarray) -> None:
    value = _column_to_attribute(values, None)
    data[name] = value.to(data.edge_index.device) if isinstance(value, torch.Tensor) else value


def add_node_attributes(df: pd.DataFrame, g: Graph) -> None:
    """Node attributes from a DataFrame: nodes in column ``v`` (IDs) or ``index`` (indices), every other column ``x`` becomes
    ``node_x`` ordered by node index; the frame must cover every node exactly once."""
    if "v" in df:
        attributed = list(df["v"])
    elif "index" in df:
        attributed = list(df["index"])
    else:
        raise ValueError("DataFrame must either have `index` or `v` column")
    if len(set(attributed)) < len(attributed):
        raise ValueError("DataFrame cannot contain multiple attribute values for single node")
    if "v" in df:
        if set(attributed) != set(g.nodes):
            raise ValueError("Mismatch between nodes in DataFrame and nodes in graph")
        node_idx = np.asarray(g.mapping.to_idxs(attributed).tolist())
    else:
        if set(attributed) != set(range(g.n)):
            raise ValueError("Mismatch between nodes in DataFrame and nodes in graph")
        node_idx = np.asarray(attributed)
    order = np.argsort(node_idx, kind="stable")               # row of the frame that describes node 0, 1, ...
    for attr in df.columns:
        if attr not in ("v", "index"):
            _store_column(g.data, attr if attr.startswith("node_") else "node_" + attr, df[attr].values[order])


def add_edge_attributes(df: pd.DataFrame, g: Graph, time_attr: str | None = None) -> None:
    """Edge attributes from a DataFrame with columns ``v``, ``w`` (and ``time_attr`` for temporal graphs): every other column ``x``
    becomes ``edge_x`` in the graph's edge order; the frame must describe every edge of the graph."""
    if "v" not in df or "w" not in df:
        raise ValueError("Data frame must have columns `v` and `w` for source and target nodes")
    node_ids = set(df["v"]).union(set(df["w"]))
    if not node_ids.issubset(set(g.nodes)):
        raise ValueError(f"DataFrame contains nodes {node_ids - set(g.nodes)} that do not exist in the graph. "
                         "Please ensure all nodes in the DataFrame are present in the graph.")
    if g.m != len(df):
        raise ValueError(f"DataFrame contains {len(df)} edges, but the graph has {g.m} edges. "
                         "Please ensure the DataFrame matches the number of edges in the graph.")
    src = g.mapping.to_idxs(df["v"].tolist()).tolist()
    tgt = g.mapping.to_idxs(df["w"].tolist()).tolist()
    attrs = [a for a in df.columns if a not in ("v", "w")]
    position = np.empty(len(df), dtype=np.int64)             # position[k] = index of the graph edge that row k describes
    if time_attr is not None:
        if time_attr not in df:
            raise ValueError(f"Data frame must have column {time_attr} for time stamps")
        attrs.remove(time_attr)
        lookup = g.tedge_to_index
        for k, (s_, t_, ts) in enumerate(zip(src, tgt, df[time_attr].values.tolist())):
            if (s_, t_, ts) not in lookup:
                raise ValueError(f"Edge ({s_}, {t_}) does not exist at time {ts} in the graph.")
            position[k] = lookup[s_, t_, ts]
    else:
        lookup = g.edge_to_index
        for k, (s_, t_) in enumerate(zip(src, tgt)):
            if (s_, t_) not in lookup:
                raise ValueError(f"Edge ({s_}, {t_}) does not exist in the graph.")
            position[k] = lookup[s_, t_]
    # like the reference (pandas.py:308-315: ``df.iloc[edge_idx]``) the rows are taken in the order of the positions found
    for attr in attrs:
        _store_column(g.data, attr if attr.startswith("edge_") else "edge_" + attr, df[attr].values[position])


def graph_to_df(graph: Graph, node_indices: Optional[bool] = False) -> pd.DataFrame:
    """One row per edge: ``v``, ``w`` and every ``edge_*`` attribute."""
    ei = graph.data.edge_index.cpu()
    if node_indices or not graph.mapping.has_ids:
        v, w = ei[0].numpy(), ei[1].numpy()
    else:
        v, w = graph.mapping.to_ids(ei[0]), graph.mapping.to_ids(ei[1])
    frame = pd.DataFrame({"v": v, "w": w})
    for attr in graph.edge_attrs():
        value = graph.data[attr]
        value = value.cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
        frame[attr] = list(value) if value.ndim > 1 else value
    return frame


def read_csv_graph(filename: str, sep: str = ",", header: bool = True, is_undirected: bool = False, multiedges: bool = False,
                   **kwargs: Any) -> Graph:
    df = pd.read_csv(filename, header=0 if header else None, sep=sep)
    return df_to_graph(df, is_undirected=is_undirected, multiedges=multiedges, **kwargs)


def temporal_graph_to_df(graph: TemporalGraph, node_indices: Optional[bool] = False) -> pd.DataFrame:
    """One row per event: ``v``, ``w``, ``t`` and every ``edge_*`` attribute, in time order."""
    ei = graph.data.edge_index.cpu()
    if node_indices or not graph.mapping.has_ids:
        v, w = ei[0].numpy(), ei[1].numpy()
    else:
        v, w = graph.mapping.to_ids(ei[0]), graph.mapping.to_ids(ei[1])
    frame = pd.DataFrame({"v": v, "w": w, "t": graph.data.time.cpu().numpy()})
    for attr in graph.edge_attrs():
        value = graph.data[attr]
        value = value.cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
        frame[attr] = list(value) if value.ndim > 1 else value
    return frame


def read_csv_temporal_graph(filename: str, sep: str = ",", header: bool = True, timestamp_format: str = "%Y-%m-%d %H:%M:%S",
                            time_rescale: int = 1, **kwargs: Any) -> TemporalGraph:
    df = pd.read_csv(filename, header=0 if header else None, sep=sep)
    return df_to_temporal_graph(df, timestamp_format=timestamp_format, time_rescale=time_rescale, **kwargs)


def write_csv(graph, node_indices: bool = False, path_or_buf: Any = None, **pdargs: Any) -> None:
    frame = temporal_graph_to_df(graph, node_indices) if isinstance(graph, TemporalGraph) else graph_to_df(graph, node_indices)
    frame.to_csv(index=False, path_or_buf=path_or_buf, **pdargs)


def read_csv_path_data(path_or_buf: Any = None, weight: bool = True, sep=",", device: Optional[torch.device] = None) -> PathData:
    """Walks from an n-gram file: one walk per line, node IDs separated by ``sep``, optionally a trailing weight."""
    with open(path_or_buf, "r") as fh:
        rows = [r for r in csv.reader(fh, delimiter=sep) if r]
    if weight:
        paths = [r[:-1] for r in rows]
        weights = [ast.literal_eval(r[-1]) for r in rows]
    else:
        paths, weights = rows, [1.0] * len(rows)
    mapping = IndexMap()
    mapping.add_ids(np.unique(np.hstack(paths)))
    out = PathData(mapping, device)
    out.append_walks(node_seqs=paths, weights=weights)
    return out

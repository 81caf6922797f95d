"""Ingest of time-stamped edge lists and n-gram path files, reference ``pathpyG.io.pandas``:
``df_to_temporal_graph`` (:318-397), ``temporal_graph_to_df`` (:431-470), ``read_csv_temporal_graph`` (:511-546),
``write_csv`` (:548-570), ``read_csv_path_data`` (:572-599).

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


def write_csv(graph: TemporalGraph, node_indices: bool = False, path_or_buf: Any = None, **pdargs: Any) -> None:
    temporal_graph_to_df(graph, node_indices=node_indices).to_csv(index=False, path_or_buf=path_or_buf, **pdargs)


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

"""Rolling time window over a temporal graph (reference src/pathpyG/algorithms/rolling_time_window.py): an iterator of
time-slice graphs ``to_static_graph(weighted, (t, t + window_size))`` for ``t = start_time, start_time + step_size, ...``."""
from __future__ import annotations


class RollingTimeWindow:
    def __init__(self, temporal_graph, window_size, step_size=1, return_window: bool = False, weighted: bool = True):
        self.g = temporal_graph
        self.window_size = window_size
        self.step_size = step_size
        self.current_time = self.g.start_time
        self.return_window = return_window
        self.weighted = weighted

    def __iter__(self):
        return self

    def __next__(self):
        if self.current_time is None or self.current_time > self.g.end_time:
            raise StopIteration()
        time_window = (self.current_time, self.current_time + self.window_size)
        s = self.g.to_static_graph(weighted=self.weighted, time_window=time_window)
        self.current_time += self.step_size
        return (s, time_window) if self.return_window else s

"""LossMeter (evo_motion_networks/include/evo_motion_networks/metrics.h:13-55, src/metrics.cpp:12-75): the windowed mean the
reference's agents report through Agent::get_metrics() and its train loop prints and writes to CSV.  fp32 like the reference
(`float` values, std::accumulate from 0.f in order).  The reference's own known answers for it
(evo_motion_networks/tests/src/test_metrics.cpp:20-25 — the only golden values its test suite holds) are tests/test_metrics.py."""
import os

import numpy as np


class LossMeter:
    def __init__(self, name, window_size=None):
        self.name, self.window_size, self.curr_step, self.results = name, window_size, 0, []

    def add(self, value):
        # metrics.cpp:17-23: room is made BEFORE the value goes in, so the window never holds more than window_size values
        while self.window_size is not None and len(self.results) >= self.window_size:
            self.results.pop(0)
        self.results.append(np.float32(value))
        self.curr_step += 1

    def set_window_size(self, new_window_size):
        self.window_size = new_window_size

    def loss(self):
        """mean of the window; the default value 0 when nothing was added (metrics.cpp:25-29,65-66)"""
        if not self.results:
            return 0.0
        s = np.float32(0.0)
        for v in self.results:
            s = np.float32(s + v)
        return float(np.float32(s / np.float32(len(self.results))))

    def loss_to_string(self, loss_value):
        return "%.6f" % loss_value   # std::fixed << std::setprecision(6), metrics.cpp:70-74

    def to_csv(self, output_directory):
        """metrics.cpp:37-49, literally: a new file gets the header `step,loss`; then the file is REOPENED for writing (which
        truncates it) and receives the one line `curr_step,loss` — the file always holds the latest value only"""
        path = os.path.join(output_directory, self.name + ".csv")
        if not os.path.exists(path):
            with open(path, "w") as f:
                f.write("step,loss\n")
        with open(path, "w") as f:
            f.write("%d,%s\n" % (self.curr_step, self.loss_to_string(self.loss())))

    def to_string(self):
        return "%s = %s" % (self.name, self.loss_to_string(self.loss()))

    __str__ = to_string

    @property
    def values(self):
        return [float(v) for v in self.results]

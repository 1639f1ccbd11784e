"""Minimal in-memory EpochLogger with the calls the path's host classes make (``utilities/logx.py:435-481``):
``store(**kw)`` appends to per-key lists, ``log_tabular`` turns a key's list into Average / Std / Max / Min entries of
the current row (or records a given value), ``dump_tabular`` returns the row and starts a new one.  File / stdout
output, MPI statistics and TF graph saving of the reference's logger are control plane and not reproduced."""
import numpy as np


class EpochLogger:
    def __init__(self, *a, **k):
        self.epoch_dict = {}
        self.row = {}
        self.stored = {}          # last value stored per key (convenience for tests)

    def log(self, msg, color=None):
        pass

    def store(self, **kwargs):
        for k, v in kwargs.items():
            self.epoch_dict.setdefault(k, []).append(v)
            self.stored[k] = v

    def log_tabular(self, key, val=None, with_min_and_max=False, average_only=False):
        if val is not None:
            self.row[key] = val
        else:
            v = self.epoch_dict.get(key, [])
            if v:
                vals = np.concatenate([np.asarray(x, np.float64).reshape(-1) for x in v])
                self.row[key if average_only else key + 'Average'] = float(np.mean(vals))
                if not average_only:
                    self.row[key + 'Std'] = float(np.std(vals))
                if with_min_and_max:
                    self.row[key + 'Max'] = float(np.max(vals))
                    self.row[key + 'Min'] = float(np.min(vals))
        self.epoch_dict[key] = []

    def dump_tabular(self, output_dir=None, print_out=False):
        row, self.row = self.row, {}
        if print_out:
            for k in sorted(row):
                print("| %25s | %15s |" % (k, row[k]))
        return row

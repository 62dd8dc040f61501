"""Scalar log of a training run: what the reference hands to Lightning's loggers.

The reference logs ``train_loss`` / ``train_r2`` / ``val_loss`` / ``val_r2`` per step (``self.log(..., on_step=True)``,
src/sps/models/models.py:74-75,80-81) and the learning rate per step (``LearningRateMonitor(logging_interval="step")``,
scripts/train.py:38) into ``./tb_logs/<EXPERIMENT.ID>/version_<n>/`` (TensorBoardLogger, scripts/train.py:46-50).  TensorBoard
is not part of this environment; the same scalars go into ``metrics.csv`` in the same directory layout, in the column layout of
Lightning's CSVLogger (``epoch, step, <one column per metric>``; a row holds the metrics logged at that step).

Values may be device tensors: they are kept as they are and only read when ``flush()`` is called (once per epoch by
scripts/train.py), so logging never synchronises the training loop.
"""
from __future__ import annotations

import csv
import os


class ScalarLog:
    def __init__(self, root: str, name: str):
        base = os.path.join(root, name)
        os.makedirs(base, exist_ok=True)
        taken = [int(d.split("_")[1]) for d in os.listdir(base) if d.startswith("version_") and d.split("_")[1].isdigit()]
        self.dir = os.path.join(base, f"version_{max(taken) + 1 if taken else 0}")
        os.makedirs(self.dir)
        self.path = os.path.join(self.dir, "metrics.csv")
        self._rows: list[tuple[int, int, dict]] = []     # (epoch, step, {name: value or tensor})
        self._written: list[dict] = []
        self._names: list[str] = []

    def log(self, epoch: int, step: int, **scalars) -> None:
        self._rows.append((int(epoch), int(step), scalars))

    def flush(self) -> None:
        """Reads the pending values (this is where device tensors are synchronised) and rewrites metrics.csv."""
        for epoch, step, scalars in self._rows:
            row = {"epoch": epoch, "step": step}
            for k, v in scalars.items():
                row[k] = float(v)
                if k not in self._names:
                    self._names.append(k)
            self._written.append(row)
        self._rows.clear()
        tmp = self.path + ".tmp"
        with open(tmp, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["epoch", "step"] + self._names, restval="")
            w.writeheader()
            w.writerows(self._written)
        os.replace(tmp, self.path)

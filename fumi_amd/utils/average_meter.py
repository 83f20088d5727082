"""Running mean of a scalar metric with the interface of the reference's meter (fumi/utils/average_meter.py:1-17:
``update(val, n=1)``, ``reset()``, attributes ``val`` / ``sum`` / ``count`` / ``avg``).

The values handed in by the validation / test loops are the lazily materialised scalars of ``fumi_amd/lazy.py`` (they live
in pinned host memory the GPU writes to): the meter only records them, and the weighted sum is formed when ``sum`` or
``avg`` is read -- so a test loop queues all its meta-batches without waiting for the device after each one."""


class AverageMeter:
    __slots__ = ("_items", "_weight")

    def __init__(self):
        self.reset()

    def reset(self):
        self._items = []        # (value, weight) pairs; values may be lazy scalars
        self._weight = 0

    def update(self, val, n=1):
        self._items.append((val, n))
        self._weight += n

    @property
    def val(self):
        """The most recent value (0 before the first update, like the reference)."""
        return self._items[-1][0] if self._items else 0

    @property
    def count(self):
        return self._weight

    @property
    def sum(self):
        total = 0
        for v, n in self._items:
            total += v * n
        return total

    @property
    def avg(self):
        return self.sum / self._weight if self._weight else 0

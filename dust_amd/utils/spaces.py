"""`Box(dim, low, high, dtype)` - same attributes as dust/utils/spaces.py:4-65 (`.dim`, `.shape`, `.low`, `.high`, `.dtype`)."""
import torch


class Box:
    def __init__(self, dim, low=None, high=None, dtype=torch.float):
        assert dtype is not None and isinstance(dtype, torch.dtype), "Data type must be of class `torch.dtype`."
        assert dim > 0, "Dimension must be a strictly positive integer."
        self.dtype = dtype
        self._dim = dim
        self._shape = torch.Size([dim])
        self.low = self._bound(low, -float("inf"), "Lower")
        self.high = self._bound(high, float("inf"), "Higher")

    def _bound(self, v, default, name):
        if v is None:
            return torch.full(self._shape, default, dtype=torch.float)
        t = torch.as_tensor(v)
        if t.ndim == 0:
            return torch.full(self._shape, float(t), dtype=torch.float)
        assert t.shape == self._shape, "%s boundary must have same dimensions as space Box." % name
        return t

    @property
    def dim(self):
        return self._dim

    @property
    def shape(self):
        return self._shape

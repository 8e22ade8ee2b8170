"""Parameter constraints with gpytorch's parameterisation (gpytorch.constraints, absent from this image): a module
maps an unconstrained raw parameter to the constrained value.  ``Positive`` / ``GreaterThan`` use softplus
(value = lower + softplus(raw)), ``Interval`` a sigmoid (value = lower + (upper - lower) * sigmoid(raw)); a raw
parameter initialised at 0 therefore starts at lower + 0.6931 resp. at the middle of the interval, as upstream.
The reference's BO driver passes ``Interval(1e-4, 12.0)`` for length- and output-scales
(experiments/bayesopt/bayesopt.py:72-76)."""
import math

import torch
from torch.nn.functional import softplus


class _Constraint(torch.nn.Module):
    lower_bound = -math.inf
    upper_bound = math.inf

    def transform(self, raw):
        raise NotImplementedError

    def inverse_transform(self, value):
        raise NotImplementedError

    def check(self, value):
        return bool(((value >= self.lower_bound) & (value <= self.upper_bound)).all())


class GreaterThan(_Constraint):
    def __init__(self, lower_bound):
        super().__init__()
        self.lower_bound = float(lower_bound)

    def transform(self, raw):
        return softplus(raw) + self.lower_bound

    def inverse_transform(self, value):
        x = (torch.as_tensor(value, dtype=torch.float64) - self.lower_bound).clamp_min(1e-300)
        return x + torch.log(-torch.expm1(-x))

    def __repr__(self):
        return f"{type(self).__name__}({self.lower_bound:.3E})"


class Positive(GreaterThan):
    def __init__(self):
        super().__init__(0.0)


class Interval(_Constraint):
    def __init__(self, lower_bound, upper_bound):
        super().__init__()
        if not float(lower_bound) < float(upper_bound):
            raise RuntimeError("Interval needs lower_bound < upper_bound")
        self.lower_bound = float(lower_bound)
        self.upper_bound = float(upper_bound)

    def transform(self, raw):
        return self.lower_bound + (self.upper_bound - self.lower_bound) * torch.sigmoid(raw)

    def inverse_transform(self, value):
        v = torch.as_tensor(value, dtype=torch.float64)
        if not self.check(v):
            raise RuntimeError(f"value outside the constraint interval [{self.lower_bound}, {self.upper_bound}]")
        u = ((v - self.lower_bound) / (self.upper_bound - self.lower_bound)).clamp(1e-15, 1 - 1e-15)
        return torch.log(u) - torch.log1p(-u)

    def __repr__(self):
        return f"Interval({self.lower_bound:.3E}, {self.upper_bound:.3E})"

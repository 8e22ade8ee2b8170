"""Feature extractors in front of the GP (reference online_gp/models/stems.py): any torch module with ``input_dim`` /
``output_dim`` attributes whose outputs lie inside the inducing grid's bounds.

``Identity``    the pass-through of the hot-path configurations (raw inputs are the features).  It owns no parameters and
                no sub-modules; the streaming wrapper gives parameter-free stems a no-op optimiser, so nothing here has to
                pretend otherwise.
``LinearStem``  one affine layer, batch normalisation without learned scale / shift, then tanh(z / 2): features in (-1, 1).
``MLP``         ReLU hidden layers in front of the same normalised, squashed output layer.

The two learned stems are plain torch.nn; they exist so that the stem-loss path (sm_partial_mll: input gradients of the
interpolation weights, mlls/streaming_partial_mll.py) can be exercised end to end."""
import torch


class Identity(torch.nn.Identity):
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = self.output_dim = int(input_dim)


class _SquashedHead(torch.nn.Sequential):
    """layers -> BatchNorm1d(affine=False) -> tanh(z / 2)"""

    def __init__(self, layers, input_dim, feature_dim, momentum=0.1):
        super().__init__(*layers, torch.nn.BatchNorm1d(feature_dim, affine=False, momentum=momentum))
        self.input_dim, self.output_dim = int(input_dim), int(feature_dim)

    def forward(self, inputs):
        return torch.tanh(0.5 * super().forward(inputs))


class LinearStem(_SquashedHead):
    def __init__(self, input_dim, feature_dim):
        super().__init__([torch.nn.Linear(input_dim, feature_dim)], input_dim, feature_dim)


class MLP(_SquashedHead):
    """`depth` hidden ReLU layers of widths hidden_dims[0..depth-1] (a list or a comma-separated string, as the reference's
    configuration files write it), then the squashed linear head."""

    def __init__(self, input_dim, feature_dim, depth, hidden_dims):
        widths = [int(w) for w in hidden_dims.split(",")] if isinstance(hidden_dims, str) else [int(w) for w in hidden_dims]
        if depth < 1 or len(widths) < depth:
            raise ValueError("MLP needs depth >= 1 and one hidden width per layer")
        layers, fan_in = [], input_dim
        for width in widths[:depth]:
            layers += [torch.nn.Linear(fan_in, width), torch.nn.ReLU()]
            fan_in = width
        layers.append(torch.nn.Linear(fan_in, feature_dim))
        super().__init__(layers, input_dim, feature_dim)

"""Parameter containers with the reference's module tree (radiance_fields/mlp.py:14-111,168-208), so that
state_dict keys / shapes / dtypes are identical and checkpoints are interchangeable.

These modules hold parameters only: on the product path all arithmetic happens in libeonerf_hip.so; they are not
meant to be called.
"""
import torch
import torch.nn as nn


class MLP(nn.Module):
    """hidden_layers.{i}.{weight,bias} (+ output_layer.{weight,bias}); skip-concat widens layer skip+1's input."""

    def __init__(self, input_dim, output_dim=None, net_depth=8, net_width=256, skip_layer=4, output_enabled=True):
        super().__init__()
        self.input_dim, self.net_depth, self.net_width, self.skip_layer = input_dim, net_depth, net_width, skip_layer
        self.hidden_layers = nn.ModuleList()
        fan_in = input_dim
        for i in range(net_depth):
            self.hidden_layers.append(nn.Linear(fan_in, net_width))
            widen = skip_layer is not None and i % skip_layer == 0 and i > 0
            fan_in = net_width + input_dim if widen else net_width
        if output_enabled:
            self.output_layer = nn.Linear(fan_in, output_dim)
            self.output_dim = output_dim
        else:
            self.output_dim = fan_in
        self.reset_parameters()

    def reset_parameters(self):
        """Xavier-uniform weights, zero biases (radiance_fields/mlp.py:22-28,67-85)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)


class DenseLayer(MLP):
    def __init__(self, input_dim, output_dim):
        super().__init__(input_dim=input_dim, output_dim=output_dim, net_depth=0)


class SinusoidalEncoder(nn.Module):
    """Only the int64 `scales` buffer of the reference encoder (mlp.py:177-179); the encoding is fused in the kernels."""

    def __init__(self, x_dim, min_deg, max_deg, use_identity=True):
        super().__init__()
        self.x_dim, self.min_deg, self.max_deg, self.use_identity = x_dim, min_deg, max_deg, use_identity
        self.register_buffer("scales", torch.tensor([2 ** i for i in range(min_deg, max_deg)]))

    @property
    def latent_dim(self):
        return (int(self.use_identity) + (self.max_deg - self.min_deg) * 2) * self.x_dim

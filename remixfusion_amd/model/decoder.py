"""Decoder MLPs: same classes / parameter layout as the reference (model/decoder.py:6-146) so that
``state_dict`` keys and optimizer parameter groups line up.  The hot path never calls these
modules' ``forward``: JointEncoding hands their weights to the fused MFMA kernels.  ``forward`` is
kept for API completeness and runs on the device the tensors live on (plain ``F.linear``)."""
from __future__ import annotations

import torch
import torch.nn as nn


def _mlp(in_dim: int, hidden: int, out_dim: int, n_layers: int) -> nn.Sequential:
    layers = []
    for l in range(n_layers):
        i = in_dim if l == 0 else hidden
        o = out_dim if l == n_layers - 1 else hidden
        layers.append(nn.Linear(i, o, bias=False))
        if l != n_layers - 1:
            layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class ColorNet(nn.Module):
    def __init__(self, config, input_ch=4, geo_feat_dim=15, hidden_dim_color=64, num_layers_color=3):
        super().__init__()
        if config["decoder"]["tcnn_network"]:
            raise NotImplementedError("tcnn FullyFusedMLP branch: every reference config sets tcnn_network False")
        self.config, self.input_ch, self.geo_feat_dim = config, input_ch, geo_feat_dim
        self.hidden_dim_color, self.num_layers_color = hidden_dim_color, num_layers_color
        self.model = _mlp(input_ch + geo_feat_dim, hidden_dim_color, 3, num_layers_color)

    def forward(self, input_feat):
        return self.model(input_feat)


class SDFNet(nn.Module):
    def __init__(self, config, input_ch=3, geo_feat_dim=15, hidden_dim=64, num_layers=2):
        super().__init__()
        if config["decoder"]["tcnn_network"]:
            raise NotImplementedError("tcnn FullyFusedMLP branch: every reference config sets tcnn_network False")
        self.config, self.input_ch, self.geo_feat_dim = config, input_ch, geo_feat_dim
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        self.model = _mlp(input_ch, hidden_dim, 1 + geo_feat_dim, num_layers)

    def forward(self, x, return_geo=True):
        out = self.model(x)
        return out if return_geo else out[..., :1]


class ColorSDFNet(nn.Module):
    """sdf_net: [emb, pos, tsdf] -> (sdf, geo15); color_net: [pos, geo15, ex_rgb] -> rgb."""

    def __init__(self, config, input_ch=3, input_ch_pos=12):
        super().__init__()
        dec = config["decoder"]
        self.config = config
        self.color_net = ColorNet(config, input_ch=input_ch_pos + 3, geo_feat_dim=dec["geo_feat_dim"],
                                  hidden_dim_color=dec["hidden_dim_color"], num_layers_color=dec["num_layers_color"])
        self.sdf_net = SDFNet(config, input_ch=input_ch + input_ch_pos + 1, geo_feat_dim=dec["geo_feat_dim"],
                              hidden_dim=dec["hidden_dim"], num_layers=dec["num_layers"])

    def fused_weights(self):
        """(W1 [32,81], W2 [16,32], W3 [32,66], W4 [3,32]) for the fused kernels; raises if the
        architecture differs from what librfx implements."""
        s, c = self.sdf_net.model, self.color_net.model
        ws = (s[0].weight, s[2].weight, c[0].weight, c[2].weight)
        if len(s) != 3 or len(c) != 3 or tuple(map(lambda w: tuple(w.shape), ws)) != ((32, 81), (16, 32), (32, 66), (3, 32)):
            raise NotImplementedError("librfx fuses the reference architecture 81->32->16 / 66->32->3 only")
        return ws

    def forward(self, embed, embed_pos, ex_tsdf, ex_rgb):
        if embed_pos is not None:
            h = self.sdf_net(torch.cat([embed, embed_pos, ex_tsdf], dim=-1), return_geo=True)
        else:
            h = self.sdf_net(embed, return_geo=True)
        sdf, geo_feat = h[..., :1], h[..., 1:]
        if embed_pos is not None:
            rgb = self.color_net(torch.cat([embed_pos, geo_feat, ex_rgb], dim=-1))
        else:
            rgb = self.color_net(torch.cat([geo_feat], dim=-1))
        return torch.cat([rgb, sdf], -1)

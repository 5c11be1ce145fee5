"""Neighbourhood-limit calibration.

The reference's samplers iterate ~1000 batches WITHOUT limits, histogram the number of real
neighbours of every point per layer and keep, per layer, the smallest width that leaves
`untouched_ratio` (0.9) of the neighbourhoods uncropped
(datasets/DALES_PseudoLabel.py:1186-1189 histogram size, :1238-1240 counts, :1321-1324 percentile;
the crop itself is datasets/common.py:336-346).  The batch-size PID controller of the same routine
(:1195-1279) belongs to the dataset samplers and is not part of this path.

Here the histograms are accumulated on the device from the index matrices of un-limited pyramids
(`weasal_amd.pyramid.build_batch(..., neighborhood_limits=())`).
"""
import numpy as np
import torch


def histogram_size(config):
    """DALES_PseudoLabel.py:1186"""
    return int(np.ceil(4 / 3 * np.pi * (config.deform_radius + 1) ** 3))


def limits_from_histograms(neighb_hists, untouched_ratio=0.9):
    """neighb_hists [num_layers, hist_n] -> int limits per layer (DALES_PseudoLabel.py:1321-1324)"""
    neighb_hists = np.asarray(neighb_hists)
    hist_n = neighb_hists.shape[1]
    cumsum = np.cumsum(neighb_hists.T, axis=0)
    return np.sum(cumsum < (untouched_ratio * cumsum[hist_n - 1, :]), axis=0)


class NeighborhoodCalibrator:
    def __init__(self, config, untouched_ratio=0.9):
        self.hist_n = histogram_size(config)
        self.num_layers = config.num_layers
        self.untouched_ratio = untouched_ratio
        self.hists = None

    def update(self, batch):
        """add the neighbour-count histograms of one un-limited batch (batch.neighbors[l] int64 [N_l, H_l])"""
        rows = []
        for mat in batch.neighbors[:self.num_layers]:
            counts = (mat < mat.shape[0]).sum(dim=1)                    # :1238 (shadow index = number of supports)
            h = torch.bincount(counts, minlength=self.hist_n)[:self.hist_n]
            rows.append(h)
        hists = torch.stack(rows)
        self.hists = hists if self.hists is None else self.hists + hists
        return self

    def limits(self):
        return limits_from_histograms(self.hists.cpu().numpy(), self.untouched_ratio)

"""Neighbourhood-limit calibration.

The reference's samplers iterate ~1000 batches WITHOUT limits, histogram the number of real
neighbours of every point per layer and keep, per layer, the smallest width that leaves
`untouched_ratio` (0.9) of the neighbourhoods uncropped
(datasets/DALES_PseudoLabel.py:1186-1189 histogram size, :1238-1240 counts, :1321-1324 percentile;
the crop itself is datasets/common.py:336-346).  The batch-size PID controller of the same routine
(:1190-1241) is `BatchLimitController` below: host arithmetic on the number of spheres per batch.

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


class BatchLimitController:
    """The PID loop that tunes `batch_limit` (the point budget of a stacked batch) until batches hold
    `target_b` spheres on average (datasets/DALES_PseudoLabel.py:1190-1241): call `update(b)` with the number
    of spheres of every calibration batch and use `.batch_limit` for the next one; `.converged` turns True when
    the low-passed batch size has stayed within `converge_threshold` of the target for 30 batches."""

    def __init__(self, target_b, batch_limit=1.0, expected_n=20000, converge_threshold=0.1):
        self.target_b = target_b
        self.batch_limit = float(batch_limit)
        self.kp = expected_n / 200                  # :1194
        self.ki = 0.001 * self.kp
        self.kd = 5 * self.kp
        self.low_pass_t = 100
        self.converge_threshold = converge_threshold
        self.estim_b = 0.0
        self.error_i = 0.0
        self.last_error = 0.0
        self.smooth_errors = []
        self.finer = False
        self.stabilized = False
        self.converged = False
        self.steps = 0

    def update(self, b):
        self.estim_b += (b - self.estim_b) / self.low_pass_t          # :1219
        error = self.target_b - b
        self.error_i += error
        error_d = error - self.last_error
        self.last_error = error
        self.smooth_errors.append(self.target_b - self.estim_b)
        if len(self.smooth_errors) > 30:
            self.smooth_errors = self.smooth_errors[1:]
        self.batch_limit += self.kp * error + self.ki * self.error_i + self.kd * error_d      # :1231
        if not self.stabilized and self.batch_limit < 0:              # unstable start: damp the gains once (:1234-1238)
            self.kp *= 0.1
            self.ki *= 0.1
            self.kd *= 0.1
            self.stabilized = True
        if not self.finer and abs(self.estim_b - self.target_b) < 1:
            self.low_pass_t = 100
            self.finer = True
        if self.finer and max(abs(e) for e in self.smooth_errors) < self.converge_threshold:
            self.converged = True
        self.steps += 1
        return self.batch_limit

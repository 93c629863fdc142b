"""TEST INFRASTRUCTURE — torch restatement of the three replica-merge kernels (csrc/n2v_merge.hip, declared in
include/n2v_hip.h "replica merges"), same arguments and semantics, on tensors of any device.  The CPU tests run
the merge PROTOCOL (n2v_hip.sgns.ReplicaMerger: tiers, one-interval delay of the cold rows, collectives over
gloo) with these ops injected; tests/test_gpu_sgns.py checks the HIP kernels against them bit for bit.
The product never imports this file: without a GPU its own ops raise."""
import torch


class TorchMergeOps:
    @staticmethod
    def _wire(t, like):
        return t.to(like.dtype)

    def snapshot(self, x, xs, base, w, hot_pos, sum_prev, cold_wire, hot_wire):
        d = x - xs
        n = x.shape[0]
        hot = (hot_pos >= 0) if hot_pos is not None else torch.zeros(n, dtype=torch.bool, device=x.device)
        cold = ~hot
        if hot.any():
            hot_wire[hot_pos[hot].long()] = d[hot].to(hot_wire.dtype)
            if cold_wire is not None:
                cold_wire[hot] = 0
        if cold.any():
            if sum_prev is not None:
                base[cold] = base[cold] + w[cold, None] * sum_prev[cold].float()
            cold_wire[cold] = d[cold].to(cold_wire.dtype)
            nx = base[cold] + d[cold]
            x[cold] = nx
            xs[cold] = nx

    def pack_rows(self, x, base, rows, wire):
        wire[: rows.numel()] = (x[rows] - base[rows]).to(wire.dtype)

    def hot_apply(self, x, xs, base, w, hot_rows, hot_sum):
        b = base[hot_rows] + w[hot_rows, None] * hot_sum.float()
        base[hot_rows] = b
        x[hot_rows] = b
        xs[hot_rows] = b

    def flush(self, x, xs, base, w, hot_pos, sum_last):
        n = x.shape[0]
        cold = (hot_pos < 0) if hot_pos is not None else torch.ones(n, dtype=torch.bool, device=x.device)
        if sum_last is not None and cold.any():
            base[cold] = base[cold] + w[cold, None] * sum_last[cold].float()
        x.copy_(base)
        xs.copy_(base)

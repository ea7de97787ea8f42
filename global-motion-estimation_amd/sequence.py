"""Whole-sequence (batch) form of the hot path and its multi-GPU sharding.

The reference processes a video as independent frame pairs (results.py:41-59): pair ``p``
is ``(frames[p], frames[p + fd])`` and no state is carried between pairs.  That is the
only parallel axis the path has, so:

* on one GPU the frames stay resident in HBM (``_gme_native.Sequence``) and every
  kernel launch covers all pairs of the shard;
* across GPUs (one process per GPU) contiguous pair ranges are dealt to the ranks, each
  rank needs its pairs' frames plus ``fd`` halo frames, and the only exchange is one
  all-gather of the per-pair ``float64[6]`` parameter vectors (48 B per pair) at the end
  -- RCCL over xGMI when the process group is ``nccl``, ``gloo`` in the CPU tests.
"""
import numpy as np

import _gme_native as _native
import motion


def shard_range(n_pairs, rank, world):
    """Contiguous pair range [start, stop) of `rank`."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return (rank * n_pairs) // world, ((rank + 1) * n_pairs) // world


def shard_frames(n_pairs, frame_distance, rank, world):
    """(first_frame, n_frames) a rank must hold: its pairs plus the `fd` halo."""
    start, stop = shard_range(n_pairs, rank, world)
    if stop == start:
        return start, 0
    return start, (stop - start) + frame_distance


def gather_parameters(local, n_pairs, rank, world, device=None):
    """All-gather per-pair parameter rows across ranks -> float64[n_pairs, k] on every rank.

    Uses torch.distributed when a process group is initialised (backend nccl = RCCL on
    ROCm: tensors are staged on ``device``; gloo: CPU tensors); with world == 1 it is
    the identity.  Rows are padded to the largest shard so one fixed-size collective
    serves ragged shards.
    """
    local = np.ascontiguousarray(local, dtype=np.float64)
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    k = local.shape[1] if local.ndim == 2 else 6
    sizes = [shard_range(n_pairs, r, world) for r in range(world)]
    longest = max(b - a for a, b in sizes)
    buf = torch.zeros((longest, k), dtype=torch.float64)
    if len(local):
        buf[:len(local)] = torch.from_numpy(local)
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = [o.cpu().numpy()[:b - a] for o, (a, b) in zip(out, sizes)]
    return np.concatenate(rows, axis=0)


class ShardedSequence:
    """This rank's slice of a video: frames in HBM, pair-level results on demand."""

    def __init__(self, height, width, n_frames, frame_distance=1, rank=0, world=1, ctx=None):
        self.ctx = ctx or _native.default_context()
        self.H, self.W, self.fd = int(height), int(width), int(frame_distance)
        self.n_frames_total = int(n_frames)
        self.n_pairs_total = max(0, self.n_frames_total - self.fd)
        self.rank, self.world = rank, world
        self.pair_start, self.pair_stop = shard_range(self.n_pairs_total, rank, world)
        self.first_frame, n_local = shard_frames(self.n_pairs_total, self.fd, rank, world)
        self.seq = _native.Sequence(self.ctx, max(n_local, self.fd + 1), self.H, self.W) if n_local else None

    @property
    def n_pairs(self):
        return self.pair_stop - self.pair_start

    def load(self, frames):
        """`frames` is the WHOLE video (uint8[N, H, W] or a list); only this rank's slice is uploaded."""
        if self.seq is None:
            return
        for k in range(self.seq.N):
            self.seq.upload(k, np.ascontiguousarray(frames[self.first_frame + k], dtype=np.uint8)[None])

    def synth(self, seed):
        """Generate this rank's slice of the synthetic sequence `seed` on its own GPU."""
        if self.seq is not None:
            self.seq.synth(seed, self.first_frame)

    def motion_fields(self, block_size, search_window, procedure, pnorm):
        """bbme.get_motion_field for every local pair -> int32[P_local, h, w, 2]."""
        if self.seq is None:
            return np.zeros((0, int(self.H / block_size), int(self.W / block_size), 2), np.int32)
        self.seq.bbme(self.fd, block_size, search_window, procedure, pnorm)
        return self.seq.read_mv()

    def estimate(self):
        """motion.global_motion_estimation for every local pair -> float64[P_local, 6]."""
        if self.seq is None:
            return np.zeros((0, 6))
        return motion.estimate_sequence(self.seq, self.fd)

    def compensate(self, params):
        """results.py:52-59,109 for every local pair -> PSNR(current, compensated) as float64[P_local]."""
        if self.seq is None:
            return np.zeros(0)
        from cmath import log10, sqrt
        sse = self.seq.compensate(self.fd, int(motion.BBME_BLOCK_SIZE), params)
        out = np.empty(len(sse))
        for k, s in enumerate(sse):
            mse = int(s) / (self.H * self.W)
            out[k] = -1 if mse == 0 else (20 * log10(255.0 / sqrt(mse))).real
        return out

    def gather(self, local_rows, device=None):
        return gather_parameters(local_rows, self.n_pairs_total, self.rank, self.world, device)

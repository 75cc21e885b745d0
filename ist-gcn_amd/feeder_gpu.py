"""Host side of the GPU input stage: the random draws and per-frame coefficients of the reference's augmentation
(`feeder/tools.py:31-101`, applied per sample by `feeder/feeder.py:79-86`), produced for a whole batch so that the
kernels of `csrc/input.hip` can apply them inside the data_bn prologue (or on their own, `augment`).

The reference runs `random_choose` / `auto_pading` and `random_move` in numpy per sample; `random_move` loops over the
T frames in Python (tools.py:94-99).  Here the host only draws the handful of numbers per clip -- with the SAME
generator calls in the same order (`random.randint`; `random.choice`, then four `np.random.choice`), so seeding
`random` / `np.random` reproduces the reference's draws -- and interpolates the per-frame 2x3 affine in float64 exactly
as tools.py:73-90 does; everything that touches clip data happens on the GPU.
"""
import random

import numpy as np
import torch

ANGLE = [-10., -5., 0., 5., 10.]
SCALE = [0.9, 1.0, 1.1]
TRANSFORM = [-0.2, -0.1, 0.0, 0.1, 0.2]
MOVE_TIME = [1]


def choose_shift(T_raw, size, random_pad=True):
    """Frame shift of tools.random_choose (:44-57) / tools.auto_pading (:31-41): source frame = t + shift.
    T_raw > size: random crop, shift = +begin; T_raw < size: zero padding with the clip at a random (random_choose) or
    zero (auto_pading) offset, shift = -begin; equal: 0.  Draws with `random.randint` exactly where the reference does."""
    if T_raw == size:
        return 0
    if T_raw < size:
        return -(random.randint(0, size - T_raw) if random_pad else 0)
    return random.randint(0, T_raw - size)


def draw_move_nodes(T, angle=ANGLE, scale=SCALE, transform=TRANSFORM, move_time=MOVE_TIME):
    """The draws of tools.random_move (:66-75) in its order -> (node frames, [4][num_node] values A, S, T_x, T_y)."""
    mt = random.choice(move_time)
    node = np.arange(0, T, T * 1.0 / mt).round().astype(int)
    node = np.append(node, T)
    n = len(node)
    A = np.random.choice(angle, n)
    S = np.random.choice(scale, n)
    Tx = np.random.choice(transform, n)
    Ty = np.random.choice(transform, n)
    return node, np.stack([A, S, Tx, Ty])


def move_coefficients(T, node, vals):
    """tools.py:77-90: per-frame angle / scale / shift by piecewise `np.linspace` between the nodes, then
    theta = [[cos a * s, -sin a * s], [sin a * s, cos a * s]] -> [T][6] float64 rows (m00, m01, tx, m10, m11, ty)."""
    a = np.zeros(T)
    s = np.zeros(T)
    tx = np.zeros(T)
    ty = np.zeros(T)
    A, S, Tx, Ty = vals
    for i in range(len(node) - 1):
        n0, n1 = node[i], node[i + 1]
        a[n0:n1] = np.linspace(A[i], A[i + 1], n1 - n0) * np.pi / 180
        s[n0:n1] = np.linspace(S[i], S[i + 1], n1 - n0)
        tx[n0:n1] = np.linspace(Tx[i], Tx[i + 1], n1 - n0)
        ty[n0:n1] = np.linspace(Ty[i], Ty[i + 1], n1 - n0)
    return np.stack([np.cos(a) * s, -np.sin(a) * s, tx, np.sin(a) * s, np.cos(a) * s, ty], axis=1)


class GpuAugment:
    """Batch-level counterpart of the feeder's per-sample processing (feeder.py:79-86):
        random_choose -> tools.random_choose(window_size), else window_size > 0 -> tools.auto_pading(window_size);
        random_move   -> tools.random_move.
    `draw(N, T_raw)` returns (shift int32 [N] or None, move float64 [N, T, 6] or None, T) as HOST tensors for
    ops.feeder_augment / functional.InputStageFn."""

    def __init__(self, window_size=-1, random_choose=False, random_move=False):
        self.window_size, self.random_choose, self.random_move = int(window_size), bool(random_choose), bool(random_move)

    def out_frames(self, T_raw):
        if self.random_choose or self.window_size > 0:
            if self.random_choose:
                return self.window_size if self.window_size > 0 else T_raw
            return max(T_raw, self.window_size)           # auto_pading only pads (tools.py:33,39-40)
        return T_raw

    def draw(self, N, T_raw):
        T = self.out_frames(T_raw)
        shifts, moves = [], []
        for _ in range(N):                                # per sample, in the feeder's order of calls
            if self.random_choose:
                shifts.append(choose_shift(T_raw, T, random_pad=True))
            elif self.window_size > 0:
                shifts.append(0)
            if self.random_move:
                node, vals = draw_move_nodes(T)
                moves.append(move_coefficients(T, node, vals))
        shift = torch.tensor(shifts, dtype=torch.int32) if shifts else None
        move = torch.from_numpy(np.stack(moves)) if moves else None
        return shift, move, T


def augment(raw, aug):
    """(N, C, T_raw, V, M) fp32 GPU clips -> augmented (N, C, T, V, M) clips (one launch): what `Feeder.__getitem__` with
    the same options would have returned for each sample."""
    from . import ops
    shift, move, T = aug.draw(raw.shape[0], raw.shape[2])
    dev = raw.device
    return ops.feeder_augment(raw.contiguous(), None if shift is None else shift.to(dev),
                              None if move is None else move.to(dev), T)

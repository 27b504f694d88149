"""Host-side occupancy-grid construction (setup, not hot path): `ObstacleMap`, `get_obst_preset`,
`generate_obstacle_map` with the behaviour of dust/utils/obstacle_map.py:13-43,101-220,249-361 and
dust/utils/obstacle.py:57-69 (pinned by tests/golden/maps.npz).  The device-side lookup is `collision()` in csrc/common.hpp."""
from math import ceil

import numpy as np
import torch


class ObstacleMap:
    def __init__(self, map_dim, cell_size):
        assert map_dim[0] % 2 == 0 and map_dim[1] % 2 == 0
        self.map = np.zeros([ceil(map_dim[0] / cell_size), ceil(map_dim[1] / cell_size)])
        self.cell_size = cell_size
        self.origin_xi, self.origin_yi = int(self.map.shape[0] / 2), int(self.map.shape[1] / 2)
        self.x_dim, self.y_dim = self.map.shape
        self.xlim = [-cell_size * self.x_dim / 2, cell_size * self.x_dim / 2]
        self.ylim = [-cell_size * self.y_dim / 2, cell_size * self.y_dim / 2]
        self.c_offset = torch.Tensor([self.origin_xi, self.origin_yi])
        self.map_torch = None

    def add_rectangle(self, cx, cy, width, height):
        cx, cy = int(cx), int(cy)  # Obstacle.__init__ truncates the centre (obstacle.py:14-15)
        w, h = ceil(width / self.cell_size), ceil(height / self.cell_size)
        c_x, c_y = ceil(cx / self.cell_size), ceil(cy / self.cell_size)
        xs, xe = c_x - ceil(w / 2.0) + self.origin_xi, c_x + ceil(w / 2.0) + self.origin_xi
        ys, ye = c_y - ceil(h / 2.0) + self.origin_yi, c_y + ceil(h / 2.0) + self.origin_yi
        self.map[xs:xe, ys:ye] = 1  # raw (possibly negative) indices: numpy slice semantics are part of the behaviour

    def convert_map(self):
        self.map_torch = torch.from_numpy(self.map).type(torch.float)
        return self.map_torch

    def get_collisions(self, X):
        """Plant-side lookup on a handful of points (obstacle_map.py:64-93); rollouts use the HIP kernel instead."""
        X = torch.as_tensor(X, dtype=torch.float)
        occ = (X * (1 / self.cell_size) + self.c_offset).floor().type(torch.LongTensor)
        occ[..., 0] = occ[..., 0].clamp(0, self.map.shape[0] - 1)
        occ[..., 1] = occ[..., 1].clamp(0, self.map.shape[1] - 1)
        return self.map_torch[occ[..., 0], occ[..., 1]]


def _grid(n, s, w):
    half = (n - 1) / 2.0
    return [[(i - half) * s, (half - j) * s, w, w] for j in range(n) for i in range(n)]


def get_obst_preset(preset_name, obst_width=2):
    w = obst_width
    if preset_name == "grid_3x3":
        return _grid(3, 5, w)
    if preset_name == "grid_4x4":
        return _grid(4, 4, w)
    if preset_name == "grid_6x6":
        return _grid(6, 3, w)
    if preset_name == "single_centred":
        return [[0, 0, w, w]]
    if preset_name == "staggered_3-2-3":
        return [[x, y, w, w] for y, xs in ((4.0, (-4.0, 0.0, 4.0)), (0, (-6, -2, 2, 6)), (-4.0, (-4.0, 0.0, 4.0))) for x in xs]
    if preset_name == "staggered_4-3-4-3-4":
        rows = ((6, (-6, -2.0, 2.0, 6)), (3, (-4.0, 0.0, 4.0)), (0, (-6, -2.0, 2.0, 6)), (-3, (-4, 0.0, 4)), (-6, (-6, -2, 2, 6)))
        return [[x, y, w, w] for y, xs in rows for x in xs]
    raise IOError("Obstacle preset not supported: ", preset_name)


def generate_obstacle_map(map_dim=(10, 10), obst_list=(), cell_size=1.0, map_type=None, **_unused):
    m = ObstacleMap(map_dim, cell_size)
    for cx, cy, width, height in obst_list:
        m.add_rectangle(cx, cy, width, height)
    for limit in m.xlim:  # border walls (obstacle_map.py:311-321)
        m.add_rectangle(limit, 0, 4 * m.cell_size, m.ylim[1] - m.ylim[0])
    for limit in m.ylim:
        m.add_rectangle(0, limit, m.xlim[1] - m.xlim[0], 4 * m.cell_size)
    m.convert_map()
    if map_type == "direct":
        return m
    raise IOError('Map type "{}" not recognized'.format(map_type))

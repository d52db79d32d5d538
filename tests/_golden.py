"""Helpers shared by the golden-vector tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synth(N, d, seed):
    """Same generator as oracle/gen_golden.py::synth (inputs of cases stored by seed only)."""
    import math
    torch.manual_seed(seed)
    x = torch.rand(N, d, dtype=torch.float64) * 2 - 1
    if d == 1:
        f = torch.sin(3 * x[:, 0]) + 0.5 * torch.exp(-((x[:, 0] - 0.3) ** 2) / 0.3) + 0.7 * torch.sin(2 * math.pi * x[:, 0] ** 2)
    else:
        f = (torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1])
             + 0.5 * torch.exp(-((x[:, 0] - 0.3) ** 2 + (x[:, 1] + 0.3) ** 2) / 0.3)
             + 0.7 * torch.sin(2 * math.pi * (x[:, 0] ** 2 + x[:, 1] ** 2)))
        if d == 3:
            f = f * torch.cos(2 * x[:, 2])
    y = f + torch.randn(N, dtype=torch.float64) * math.sqrt(0.2)
    return x, y


_SEEDED = {"c4_se2d_hard_n100000": (100000, 2, 0), "c5_matern32_3d_n20000": (20000, 3, 1),
           "c5b_matern32_3d_l02_n20000": (20000, 3, 1)}


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    if "x" in g:
        x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    elif name == "c2_se2d_n100000":
        dat = np.load(os.path.join(GOLDEN, "data", "gp_samples_100000_0.2_2_0.2.npz"))
        x, y = torch.from_numpy(dat["x"]), torch.from_numpy(dat["y"])
    else:
        x, y = synth(*_SEEDED[name])
    assert abs(float(x.sum()) - float(g["x_checksum"])) < 1e-6 * max(1.0, abs(float(g["x_checksum"])))
    assert abs(float(y.sum()) - float(g["y_checksum"])) < 1e-6 * max(1.0, abs(float(g["y_checksum"])))
    T = g["V"].shape[0]
    N = x.shape[0]
    Z = torch.from_numpy(np.unpackbits(g["Z_bits"], axis=1)[:, :N].astype(np.float64) * 2 - 1)
    g["Z"] = Z
    return g, x, y


def oracle_kernel(g):
    from oracle import efgp_oracle as O
    return O.KernelSpec(str(g["kind"]), int(g["d"]), float(g["lengthscale"]), float(g["variance"]), float(g["nu"]))


def product_kernel(g):
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    d = int(g["d"])
    # nominal (un-rounded) values that gen_golden.py passed to the reference
    nominal = {"c1_se1d_n5000": (0.1, 2.0), "c2_se2d_n100000": (0.2, 2.0), "c3_matern52_usatemp": (0.1, 1.0),
               "c4_se2d_hard_n100000": (0.05, 3.0), "c5_matern32_3d_n20000": (0.3, 1.5), "s1_se2d_n100": (0.5, 2.0),
               "s2_matern12_1d_n200": (0.3, 1.2)}
    return nominal


def rel(a, b):
    a = torch.as_tensor(a).detach().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().cpu().reshape(-1)
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))

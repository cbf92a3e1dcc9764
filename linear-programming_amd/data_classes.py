"""Result record of every solver (reference: linprog/data_classes.py:6-12), field for field."""
from dataclasses import dataclass

import numpy as np


@dataclass
class LinProgResult:
    x: np.ndarray
    basis: np.ndarray
    cost: float
    iters: int
    optimum: bool

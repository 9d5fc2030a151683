"""Named workloads of BASELINE.json / SURVEY.md section 8(d) and the multi-GPU sharding rule."""
from dataclasses import dataclass

from . import _ffi


@dataclass(frozen=True)
class RectWorkload:
    name: str
    size: int            # square image edge
    x_begin: float
    x_end: float
    hs: int
    vs: int
    py_stop: int
    model: int = _ffi.FM_UVUXUYVXVY
    truth: tuple = (1.3, -0.7, 0.002, 0.0, 0.0, -0.001)


# config 2 of BASELINE.json: 2048^2 pair, 100x100 sectors of 19x19 = 361 samples, affine,
# pyramid 0/1/2 (the configuration the headline metric is quoted on)
C2 = RectWorkload("C2: 2048x2048 speckle pair, 100x100 sectors of 19x19 samples, affine 6-DOF, bicubic, pyramid 0/1/2",
                  2048, 24.0, 2023.0, 100, 100, 2)
# config 4's sector geometry (one pair of the sequence): 224x224 sectors of 9x9
C4 = RectWorkload("C4: 2048x2048, 224x224 sectors of 9x9 samples, affine, pyramid 0/1/2",
                  2048, 24.0, 2023.0, 224, 224, 2)
# config 5: 8192^2, 447x447 sectors of 17x17, 4 levels
C5 = RectWorkload("C5: 8192x8192, 447x447 sectors of 17x17 samples, affine, pyramid 0/1/2/3",
                  8192, 32.0, 8159.0, 447, 447, 3)


def shard_range(n_units, rank, world):
    """Contiguous block partition of the sector index: rank r owns
    [r*S/G, (r+1)*S/G).  Returns (first, count)."""
    first = (n_units * rank) // world
    last = (n_units * (rank + 1)) // world
    return first, last - first

"""Build-owned, seeded, platform-independent weight initialiser.

No pretrained weights ship with the reference, so parity and
performance are measured on seeded random-init weights.  To make the weights
bit-identical in the fixture generator (which fills the *reference* modules via
``load_state_dict``), in the CPU oracle and on the GPU box, every tensor is
produced from integer arithmetic only: a 64-bit splitmix counter stream keyed
by FNV-1a(parameter name) ^ seed, mapped to uniform floats with 24 mantissa
bits.  Nothing here depends on torch's initialisers or RNG.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix(ctr: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = ctr + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, seed: int = 0, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 uniform in [lo, hi) for tensor `name` (deterministic everywhere)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
    bits = _splitmix(ctr) >> np.uint64(40)                      # 24 random bits
    u = bits.astype(np.float32) * np.float32(1.0 / (1 << 24))    # exact in fp32
    out = np.float32(lo) + u * np.float32(hi - lo)
    return out.astype(np.float32).reshape(shape)


def _fan_in(shape):
    if len(shape) <= 1:
        return shape[0] if shape else 1
    return int(np.prod(shape[1:]))


_DEFERRED = [False]


class deferred:
    """`with init_weights.deferred():` -- modules built inside get zero tensors instead of the seeded values: ranks >= 1 of a multi-GPU job
    receive rank 0's weights by one broadcast (lds/shard.py broadcast_state) instead of generating them again."""

    def __enter__(self):
        self.prev = _DEFERRED[0]
        _DEFERRED[0] = True

    def __exit__(self, *a):
        _DEFERRED[0] = self.prev


def init_tensor(name: str, shape, seed: int = 0) -> np.ndarray:
    if _DEFERRED[0]:
        return np.zeros(shape, dtype=np.float32)
    return _init_tensor(name, shape, seed)


def _init_tensor(name: str, shape, seed: int = 0) -> np.ndarray:
    """Role-based scale: norm gains ~1, norm/linear biases small, matrices/conv kernels
    uniform(+-1/sqrt(fan_in)) (the familiar default scale), embeddings unit variance."""
    leaf = name.rsplit(".", 2)
    is_norm = any(t in name for t in (".norm", "conv_norm_out")) and "time_emb" not in name
    if name.endswith("spk_embed.weight"):
        return uniform(name, shape, seed, -1.7320508, 1.7320508)
    if is_norm and name.endswith(".weight"):
        return uniform(name, shape, seed, 0.8, 1.2)
    if is_norm and name.endswith(".bias"):
        return uniform(name, shape, seed, -0.1, 0.1)
    if name.endswith("weight_g"):
        return uniform(name, shape, seed, 0.5, 1.5)
    if name.endswith(".bias"):
        return uniform(name, shape, seed, -0.05, 0.05)
    b = 1.0 / np.sqrt(max(1, _fan_in(shape)))
    return uniform(name, shape, seed, -b, b)


def init_state(shapes: dict, seed: int = 0, skip_prefixes=()) -> dict:
    """name -> float32 ndarray for every key of `shapes` (an arch.*_param_shapes dict)."""
    out = {}
    for k, s in shapes.items():
        if any(k.startswith(p) for p in skip_prefixes):
            continue
        out[k] = init_tensor(k, tuple(s), seed)
    return out

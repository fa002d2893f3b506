"""Descriptor sets for the vocabulary tests: images that share visual words (prototypes with bit noise)."""
import numpy as np


def noisy_descriptor_images(n_images=40, per_image=300, n_proto=200, flip=0.03, seed=0):
    rng = np.random.default_rng(seed)
    proto = rng.integers(0, 2 ** 32, (n_proto, 8), dtype=np.uint64).astype(np.uint32)
    imgs = []
    for _ in range(n_images):
        sel = rng.integers(0, n_proto, per_image)
        bits = rng.random((per_image, 8, 32)) < flip
        noise = (bits * (1 << np.arange(32, dtype=np.uint64))).sum(axis=2).astype(np.uint32)
        imgs.append(np.ascontiguousarray(proto[sel] ^ noise))
    return imgs

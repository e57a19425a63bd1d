"""The reference's SafeEmbeddingLookupSparse known answers (T/dynamic_embedding_ops_test.py:187-250,
1007-1165, 1205-1324) as data: `run(lookup)` checks a `lookup(indices, ids, dense_shape, weights, default_id)`
callable that reads from a table holding E[k] for the valid keys (misses -> zeros)."""
import numpy as np

DIM = 4
IDS = [0, 1, -100, -100, 2, 0, 1]
WEIGHTS = [1.0, 2.0, 1.0, 1.0, 3.0, 0.0, -0.5]
IDX_2D = [[0, 0], [0, 1], [0, 2], [1, 0], [3, 0], [4, 0], [4, 1]]
SHAPE_2D = [5, DIM]
IDX_3D = [[0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 1, 0], [1, 0, 0], [1, 1, 0], [1, 1, 1]]
SHAPE_3D = [2, 3, DIM]


def embeddings(rng):
  return {k: rng.standard_normal(DIM).astype(np.float32) for k in (0, 1, 2, 3, -100)}


def expected(E, weighted, default_id):
  z = np.zeros(DIM, np.float32)
  d = z if default_id is None else E[default_id]
  if weighted:      # :1033-1049 (zero vector) / :1076-1088 (special vector)
    rows = [(1.0 * E[0] + 2.0 * E[1] + 1.0 * E[-100]) / 4.0, E[-100] * 1.0, d, E[2], d]
  else:             # :1114-1126 no weights
    rows = [(E[0] + E[1] + E[-100]) / 3.0, E[-100], d, E[2], (E[0] + E[1]) / 2.0]
  return np.stack(rows).astype(np.float32)


def run(lookup, E):
  for weighted in (True, False):
    for default_id in (None, 3):
      w = WEIGHTS if weighted else None
      want = expected(E, weighted, default_id)
      got2 = lookup(IDX_2D, IDS, SHAPE_2D, w, default_id)
      np.testing.assert_allclose(got2, want, rtol=1e-6, atol=1e-6)
      got3 = lookup(IDX_3D, IDS, SHAPE_3D, w, default_id)        # :1205-1324: same cases laid out as [2, 3]
      want3 = np.concatenate([want[:2], want[2:3], want[3:5], (want[2:3])]).reshape(2, 3, DIM)
      np.testing.assert_allclose(got3, want3, rtol=1e-6, atol=1e-6)

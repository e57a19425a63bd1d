"""TEST INFRASTRUCTURE ONLY — NumPy fp32 restatement of the sparse-optimizer write-back.

The reference does not own the optimizer arithmetic: ``DynamicEmbeddingOptimizer`` reads
param + slot rows for the batch's unique ids, runs the STOCK TensorFlow dense apply kernel
on the local ``[U, dim]`` buffers and upserts param + every slot back
(PY/dynamic_embedding_optimizer.py:165-204, PY/embedding_weights.py:434-444,
create_slots :870-958).  TensorFlow (pinned 2.16.2, R/README.md:109) is a third-party
dependency absent from ``/root/reference`` and from this image, so the update rules below
restate TF's PUBLISHED op definitions (``ResourceApplyAdam`` / ``ResourceApplyAdagrad[V2]``
/ ``ResourceApplyFtrl`` / ``ResourceApplyGradientDescent``).

Parity status of THIS file: the reference's tests pin it only differentially
(T/dynamic_embedding_optimizer_test.py:347-440: de.Variable + DynamicEmbeddingOptimizer ==
dense ResourceVariable + the same TF optimizer) and hold no golden numbers => "pinned
differentially, no golden vectors".

All arithmetic is float32, one operation per NumPy call, so the sequence of roundings is
the one a straightforward fp32 kernel performs.  Duplicate ids: gradients of duplicates are
summed first, then ONE update per key (``_resource_apply_sparse_duplicate_indices``,
PY/dynamic_embedding_optimizer.py:177-190; pinned by K13).
"""
import numpy as np

f32 = np.float32


def segment_sum_by_key(keys, grads):
  """unique (first-occurrence order, like tf.unique) + unsorted_segment_sum in index order."""
  keys = np.asarray(keys, dtype=np.int64).reshape(-1)
  grads = np.asarray(grads, dtype=f32).reshape(keys.size, -1)
  order = {}
  for k in keys.tolist():
    if k not in order:
      order[k] = len(order)
  idx = np.fromiter((order[k] for k in keys.tolist()), dtype=np.int64, count=keys.size)
  uniq = np.fromiter(order.keys(), dtype=np.int64, count=len(order))
  out = np.zeros((uniq.size, grads.shape[1]), dtype=f32)
  np.add.at(out, idx, grads)  # sequential in index order
  return uniq, out, idx


def sgd(p, g, lr):
  """ResourceApplyGradientDescent: p -= lr * g."""
  return (p - f32(lr) * g).astype(f32)


def adam(p, m, v, g, lr, beta1, beta2, eps, t):
  """ResourceApplyAdam (TF1 AdamOptimizer / Keras-legacy Adam, epsilon-hat form), t 1-based:
  lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
  p -= lr_t*m/(sqrt(v)+eps)."""
  b1p = f32(np.power(f32(beta1), f32(t)))
  b2p = f32(np.power(f32(beta2), f32(t)))
  lr_t = f32(f32(lr) * np.sqrt(f32(1) - b2p) / (f32(1) - b1p))
  m = (m + (g - m) * (f32(1) - f32(beta1))).astype(f32)
  v = (v + (g * g - v) * (f32(1) - f32(beta2))).astype(f32)
  p = (p - (m * lr_t) / (np.sqrt(v) + f32(eps))).astype(f32)
  return p, m, v


def adam_lr_t(lr, beta1, beta2, t):
  b1p = f32(np.power(f32(beta1), f32(t)))
  b2p = f32(np.power(f32(beta2), f32(t)))
  return f32(f32(lr) * np.sqrt(f32(1) - b2p) / (f32(1) - b1p))


def adagrad(p, a, g, lr, eps=None):
  """ResourceApplyAdagrad: a += g^2; p -= lr*g/sqrt(a)   (TF1, a0 = 0.1)
  ResourceApplyAdagradV2 (Keras): p -= lr*g/(sqrt(a)+eps), eps = 1e-7."""
  a = (a + g * g).astype(f32)
  if eps is None:
    p = (p - f32(lr) * g / np.sqrt(a)).astype(f32)
  else:
    p = (p - f32(lr) * g / (np.sqrt(a) + f32(eps))).astype(f32)
  return p, a


def ftrl(p, a, z, g, lr, l1, l2, lr_power=-0.5):
  """ResourceApplyFtrl (no shrinkage): a' = a+g^2; sigma = (a'^-lp - a^-lp)/lr;
  z += g - sigma*p; q = a'^-lp/lr + 2*l2; p = |z|>l1 ? (sign(z)*l1 - z)/q : 0; a = a'.
  (a0 = 0.1, z0 = 0).  lr_power = -0.5 uses sqrt like TF's fast path."""
  lr, l1, l2 = f32(lr), f32(l1), f32(l2)
  a_new = (a + g * g).astype(f32)
  if lr_power == -0.5:
    pa_new, pa = np.sqrt(a_new), np.sqrt(a)
  else:
    pa_new, pa = np.power(a_new, f32(-lr_power)), np.power(a, f32(-lr_power))
  sigma = ((pa_new - pa) / lr).astype(f32)
  z = (z + g - sigma * p).astype(f32)
  q = (pa_new / lr + f32(2) * l2).astype(f32)
  p = np.where(np.abs(z) > l1, (np.sign(z) * l1 - z) / q, f32(0)).astype(f32)
  return p, a_new, z


def momentum(p, accum, g, lr, momentum_, use_nesterov=False):
  """ResourceApplyMomentum (TF core training_ops.cc): accum = accum*momentum + g; var -= lr*accum, or with
  use_nesterov var -= g*lr + accum*momentum*lr."""
  accum = (accum * f32(momentum_) + g).astype(f32)
  if use_nesterov:
    p = (p - (g * f32(lr) + accum * f32(f32(momentum_) * f32(lr)))).astype(f32)
  else:
    p = (p - accum * f32(lr)).astype(f32)
  return p, accum


def rmsprop(p, ms, mom, g, lr, rho, momentum_, eps):
  """ResourceApplyRMSProp: ms += (g*g - ms)*(1-rho); mom = mom*momentum + lr*g/sqrt(ms+eps); var -= mom."""
  ms = (ms + (g * g - ms) * f32(1.0 - rho)).astype(f32)
  mom = (mom * f32(momentum_) + (g * f32(lr)) / np.sqrt(ms + f32(eps))).astype(f32)
  return (p - mom).astype(f32), ms, mom


class SparseOptimizerOracle:
  """The reference's write-back sequence over CPU tables (one table per slot, like
  ``create_slots``): (1+S) finds -> dense apply on [U,dim] -> (1+S) upserts.

  ``param``/``slots`` are ``oracle.CpuTable``; missing rows take the Variable's initializer
  (param) or the slot initial value (zeros; Adagrad/FTRL accumulator 0.1)."""

  def __init__(self, kind, param, slots, hyper, param_default=0.0):
    self.kind, self.param, self.slots, self.h = kind, param, slots, dict(hyper)
    self.param_default = param_default
    self.step = 0

  def apply(self, keys, grads, param_defaults=None):
    uniq, g, _ = segment_sum_by_key(keys, grads)
    dim = self.param.dim
    self.step += 1
    pd = param_defaults if param_defaults is not None else np.full(dim, self.param_default, f32)
    p = self.param.find(uniq, pd)
    h = self.h
    if self.kind == "sgd":
      p = sgd(p, g, h["lr"])
      new_slots = []
    elif self.kind == "adam":
      m = self.slots[0].find(uniq, np.zeros(dim, f32))
      v = self.slots[1].find(uniq, np.zeros(dim, f32))
      p, m, v = adam(p, m, v, g, h["lr"], h["beta1"], h["beta2"], h["eps"], self.step)
      new_slots = [m, v]
    elif self.kind == "adagrad":
      a = self.slots[0].find(uniq, np.full(dim, h.get("init_acc", 0.1), f32))
      p, a = adagrad(p, a, g, h["lr"], h.get("eps"))
      new_slots = [a]
    elif self.kind == "ftrl":
      a = self.slots[0].find(uniq, np.full(dim, h.get("init_acc", 0.1), f32))
      z = self.slots[1].find(uniq, np.zeros(dim, f32))
      p, a, z = ftrl(p, a, z, g, h["lr"], h["l1"], h["l2"], h.get("lr_power", -0.5))
      new_slots = [a, z]
    elif self.kind == "momentum":
      a = self.slots[0].find(uniq, np.zeros(dim, f32))
      p, a = momentum(p, a, g, h["lr"], h["momentum"], h.get("nesterov", False))
      new_slots = [a]
    elif self.kind == "rmsprop":
      ms = self.slots[0].find(uniq, np.full(dim, h.get("initial_rms", 1.0), f32))
      mom = self.slots[1].find(uniq, np.zeros(dim, f32))
      p, ms, mom = rmsprop(p, ms, mom, g, h["lr"], h["rho"], h["momentum"], h["eps"])
      new_slots = [ms, mom]
    else:
      raise ValueError(self.kind)
    self.param.insert(uniq, p)
    for t, s in zip(self.slots, new_slots):
      t.insert(uniq, s)
    return uniq

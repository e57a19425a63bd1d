"""Pins oracle/optimizers.py (the NumPy fp32 restatement of TF's ResourceApply* rules that every fused-optimizer parity
test checks the HIP kernels against) to an implementation that is NOT itself: torch.optim on CPU, fp32, single-tensor
code path.  TensorFlow is absent from /root/reference and from this image, so this is the independent
reference available here:

  ResourceApplyGradientDescent  == torch.optim.SGD(lr)
  ResourceApplyMomentum         == torch.optim.SGD(lr, momentum[, nesterov])     (first step: buf = g, like accum = 0*m + g)
  ResourceApplyAdagrad[V2]      == torch.optim.Adagrad(lr, initial_accumulator_value=a0, eps = 0 | 1e-7)
  ResourceApplyAdam (eps-hat)   == torch.optim.Adam with eps_torch(t) = eps_tf / sqrt(1 - beta2^t):
                                   torch: p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps_torch)
                                        = lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps_torch*sqrt(1-b2^t))   = TF's rule
  ResourceApplyFtrl             == the known answers TensorFlow publishes in its own test-suite
                                (tensorflow/python/training/ftrl_test.py: testFtrlwithoutRegularization[2], testFtrlWithL1,
                                testFtrlWithL1_L2 — FTRL_KATS below; the HIP kernels run the same four in
                                tests/test_gpu_frontend.py::test_ftrl_tensorflow_known_answers)
Tolerance 2e-6 relative (+1e-7 absolute): both sides are fp32 with a different association of the same operations."""
import math

import numpy as np
import pytest
import torch

from oracle import optimizers as O

RTOL, ATOL = 2e-6, 1e-7
STEPS, N, DIM = 12, 257, 24


def _data(seed):
  rng = np.random.default_rng(seed)
  p0 = rng.standard_normal((N, DIM)).astype(np.float32)
  grads = [(rng.standard_normal((N, DIM)) * (0.3 if s % 3 else 3.0)).astype(np.float32) for s in range(STEPS)]
  grads[5][::7] = 0.0   # rows whose gradient is exactly zero in a step
  return p0, grads


def _torch_run(make_opt, p0, grads, before_step=None):
  p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
  opt = make_opt([p])
  for t, g in enumerate(grads, start=1):
    if before_step:
      before_step(opt, t)
    p.grad = torch.from_numpy(g.copy())
    opt.step()
  return p.detach().numpy(), opt.state[p]


def test_sgd_matches_torch():
  p0, grads = _data(1)
  ref, _ = _torch_run(lambda ps: torch.optim.SGD(ps, lr=0.1, foreach=False), p0, grads)
  p = p0.copy()
  for g in grads:
    p = O.sgd(p, g, 0.1)
  np.testing.assert_allclose(p, ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("nesterov", [False, True])
def test_momentum_matches_torch(nesterov):
  p0, grads = _data(2)
  ref, st = _torch_run(lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=nesterov, foreach=False), p0, grads)
  p, acc = p0.copy(), np.zeros_like(p0)
  for g in grads:
    p, acc = O.momentum(p, acc, g, 0.05, 0.9, nesterov)
  np.testing.assert_allclose(p, ref, rtol=5e-6, atol=1e-6)
  np.testing.assert_allclose(acc, st["momentum_buffer"].numpy(), rtol=5e-6, atol=1e-6)


@pytest.mark.parametrize("eps", [None, 1e-7])
def test_adagrad_matches_torch(eps):
  p0, grads = _data(3)
  ref, st = _torch_run(lambda ps: torch.optim.Adagrad(ps, lr=0.05, lr_decay=0.0, initial_accumulator_value=0.1,
                                                      eps=0.0 if eps is None else eps, foreach=False), p0, grads)
  p, a = p0.copy(), np.full_like(p0, 0.1)
  for g in grads:
    p, a = O.adagrad(p, a, g, 0.05, eps)
  np.testing.assert_allclose(a, st["sum"].numpy(), rtol=RTOL, atol=ATOL)
  np.testing.assert_allclose(p, ref, rtol=RTOL, atol=ATOL)


# betas exactly representable in fp32 (1 - 2^-10, 1 - 2^-3, ...): TF's kernel forms (1 - beta) in fp32 (T(1) - beta2, as the oracle
# does), torch forms it in double and rounds once — with beta2 = 0.999 the two factors differ by 4.7e-5 relative
# (f32(0.999) = 0.99900001287), which is a property of the two frameworks, not of the rule; that case gets the looser bound.
@pytest.mark.parametrize("lr,b1,b2,eps,rtol", [(1e-3, 0.875, 1 - 2.0 ** -10, 1e-8, 5e-6), (0.01, 0.75, 0.984375, 1e-5, 5e-6),
                                               (1e-3, 0.9, 0.999, 1e-8, 6e-5)])
def test_adam_matches_torch_with_the_epsilon_hat_mapping(lr, b1, b2, eps, rtol):
  p0, grads = _data(4)

  def eps_of_step(opt, t):
    for gr in opt.param_groups:
      gr["eps"] = eps / math.sqrt(1.0 - b2 ** t)

  ref, st = _torch_run(lambda ps: torch.optim.Adam(ps, lr=lr, betas=(b1, b2), eps=eps, foreach=False, fused=False), p0, grads,
                       before_step=eps_of_step)
  p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
  for t, g in enumerate(grads, start=1):
    p, m, v = O.adam(p, m, v, g, lr, b1, b2, eps, t)
  np.testing.assert_allclose(m, st["exp_avg"].numpy(), rtol=rtol, atol=1e-7)
  np.testing.assert_allclose(v, st["exp_avg_sq"].numpy(), rtol=rtol, atol=1e-9)
  np.testing.assert_allclose(p, ref, rtol=rtol, atol=2e-7)


def test_adam_lr_t_is_the_bias_corrected_step():
  for t in (1, 2, 10, 1000):
    want = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    assert abs(float(O.adam_lr_t(1e-3, 0.9, 0.999, t)) - want) <= 5e-5 * want   # (1 - f32(0.999)) alone is 4.7e-5 off the double value


def test_ftrl_closed_form_properties():
  """Besides TensorFlow's known answers (FTRL_KATS below), the closed form's fixed points hold: |z| <= l1 -> p = 0; l1 = l2 = 0 and
  lr_power = -0.5 reduce the rule to p = -z * lr / sqrt(a) (per-coordinate adaptive step)."""
  rng = np.random.default_rng(5)
  p = rng.standard_normal((8, 4)).astype(np.float32)
  a = np.full_like(p, 0.1)
  z = np.zeros_like(p)
  g = (rng.standard_normal((8, 4)) * 1e-3).astype(np.float32)
  p1, a1, z1 = O.ftrl(p, a, z, g, 0.05, l1=10.0, l2=0.0)
  assert np.all(p1 == 0)
  p2, a2, z2 = O.ftrl(p, a, z, g, 0.05, l1=0.0, l2=0.0)
  np.testing.assert_allclose(p2, -z2 * np.float32(0.05) / np.sqrt(a2), rtol=1e-6)
  np.testing.assert_allclose(a2, a + g * g, rtol=1e-7)


# TensorFlow's own known answers for FtrlOptimizer(3.0, initial_accumulator_value=0.1, l1, l2) — tensorflow/python/training/
# ftrl_test.py (TF 2.16): two variables, constant gradients [0.1, 0.2] and [0.01, 0.02], `steps` applications.
#   (name, var0, var1, l1, l2, steps, expected var0, expected var1)
FTRL_KATS = [
    ("testFtrlwithoutRegularization", [0.0, 0.0], [0.0, 0.0], 0.0, 0.0, 3, [-2.60260963, -4.29698515], [-0.28432083, -0.56694895]),
    ("testFtrlwithoutRegularization2", [1.0, 2.0], [4.0, 3.0], 0.0, 0.0, 3, [-2.55607247, -3.98729396], [-0.28232238, -0.56096673]),
    ("testFtrlWithL1", [1.0, 2.0], [4.0, 3.0], 0.001, 0.0, 10, [-7.66718769, -10.91273689], [-0.93460727, -1.86147261]),
    ("testFtrlWithL1_L2", [1.0, 2.0], [4.0, 3.0], 0.001, 2.0, 10, [-0.24059935, -0.46829352], [-0.02406147, -0.04830509]),
]
FTRL_KAT_GRADS = ([0.1, 0.2], [0.01, 0.02])


@pytest.mark.parametrize("kat", FTRL_KATS, ids=[k[0] for k in FTRL_KATS])
def test_ftrl_matches_tensorflow_known_answers(kat):
  """oracle.optimizers.ftrl against the numbers TensorFlow's ftrl_test.py pins (rtol 1e-5: TF's own assertAllCloseAccordingToType
  bound for float32 is 1e-6 absolute on values of this size after rounding the literals to 8 digits)."""
  _, v0, v1, l1, l2, steps, e0, e1 = kat
  for p0, g, want in ((v0, FTRL_KAT_GRADS[0], e0), (v1, FTRL_KAT_GRADS[1], e1)):
    p, a, z = np.array(p0, np.float32), np.full(2, 0.1, np.float32), np.zeros(2, np.float32)
    g = np.array(g, np.float32)
    for _ in range(steps):
      p, a, z = O.ftrl(p, a, z, g, 3.0, l1, l2)
    np.testing.assert_allclose(p, np.array(want, np.float32), rtol=1e-5, atol=0)

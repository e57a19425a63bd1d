"""Known-answer cases of the reference's own DynamicPartition / DynamicStitch tests,
transcribed as data so that the CPU restatement (oracle/frontends.py) and the device ops run the same list.
Sources: T/dynamic_partition_op_test.py:37-283,331-345, T/dynamic_stitch_op_test.py:37-169."""
import numpy as np


def partition_cases():
  cases = []
  # testSimpleOneDimensional :37-55
  cases.append(("simple_1d", np.array([0, 13, 2, 39, 4, 17], np.float32), np.array([0, 0, 2, 3, 2, 1], np.int32), 4,
                [[0, 13], [17], [2, 4], [39]]))
  # testSimpleTwoDimensional :57-76
  d = np.arange(18, dtype=np.float32).reshape(6, 3)
  cases.append(("simple_2d", d, np.array([0, 0, 2, 3, 2, 1], np.int32), 4, [d[[0, 1]], d[[5]], d[[2, 4]], d[[3]]]))
  # testLargeOneDimensional :78-92
  num = 100000
  x = np.arange(num, dtype=np.float32)
  cases.append(("large_1d", x, (np.arange(num) % 2).astype(np.int32), 2, [x[0::2], x[1::2]]))
  # testLargeTwoDimensional :94-117
  rows, cols, parts = 100000, 100, 97
  d = np.repeat(np.arange(rows, dtype=np.float32)[:, None], cols, axis=1)
  idx = ((np.arange(rows, dtype=np.int64) ** 2) % parts).astype(np.int32)
  cases.append(("large_2d", d, idx, parts, [d[idx == p] for p in range(parts)]))
  # testHigherRank :152-175 (partitions of rank 2 over data of rank 3)
  rng = np.random.default_rng(0)
  for n in (2, 5, 7):
    shape = (4, 3)
    d = rng.standard_normal(shape + (5,)).astype(np.float32)
    p = rng.integers(0, n, size=shape).astype(np.int32)
    cases.append(("higher_rank_%d" % n, d, p, n, [d[p == q] for q in range(n)]))
  # testEmptyParts :177-190
  cases.append(("empty_parts", np.array([1, 2, 3, 4], np.float32), np.array([1, 3, 1, 3], np.int32), 4,
                [[], [1, 3], [], [2, 4]]))
  # testEmptyDataTwoDimensional :192-205
  cases.append(("empty_data_2d", np.zeros((2, 0), np.float32), np.array([0, 1], np.int32), 3,
                [np.zeros((1, 0)), np.zeros((1, 0)), np.zeros((0, 0))]))
  # testEmptyPartitions :207-219
  cases.append(("empty_partitions", np.zeros((0,), np.float32), np.zeros((0,), np.int32), 2, [[], []]))
  # GPU kernel: out-of-range partition ids are discarded  (testGPUTooManyParts :221-239,
  # testGPUPartsTooLarge :241-262, testGPUAllIndicesBig :264-283)
  cases.append(("gpu_too_many_parts", np.array([1, 2, 3, 4, 5, 6], np.float32), np.array([6, 5, 4, 3, 1, 0], np.int32), 2,
                [[6], [5]]))
  cases.append(("gpu_parts_too_large", np.array([1, 2, 3, 4, 5, 6], np.float32),
                np.array([10, 11, 2, 12, 0, 1000], np.int32), 5, [[5], [], [3], [], []]))
  cases.append(("gpu_all_indices_big", np.array([1.1, 2.1, 3.1, 4.1, 5.1, 6.1], np.float32),
                np.array([90, 70, 60, 100, 110, 40], np.int32), 40, [[] for _ in range(40)]))
  # testCUBBug :331-345 (regression: 4 partitions, 1024+ elements)
  x = np.arange(2048, dtype=np.float32)
  p = (np.arange(2048) % 4).astype(np.int32)
  cases.append(("cub_bug", x, p, 4, [x[p == q] for q in range(4)]))
  return cases


def stitch_cases():
  cases = []
  # testScalar :37-47 (both orders)
  cases.append(("scalar", [np.array(0), np.array(1)], [np.array(40, np.int32), np.array(60, np.int32)], [40, 60]))
  cases.append(("scalar_rev", [np.array(1), np.array(0)], [np.array(40, np.int32), np.array(60, np.int32)], [60, 40]))
  # testSimpleOneDimensional :64-83
  cases.append(("simple_1d", [np.array([0, 4, 7]), np.array([1, 6, 2, 3, 5])],
                [np.array([0, 40, 70], np.float32), np.array([10, 60, 20, 30, 50], np.float32)],
                [0, 10, 20, 30, 40, 50, 60, 70]))
  # testOneListOneDimensional :85-92
  cases.append(("one_list", [np.array([1, 6, 2, 3, 5, 0, 4, 7])], [np.array([10, 60, 20, 30, 50, 0, 40, 70], np.int32)],
                [0, 10, 20, 30, 40, 50, 60, 70]))
  # testSimpleTwoDimensional :94-110
  want2d = [[0, 1], [10, 11], [20, 21], [30, 31], [40, 41], [50, 51], [60, 61], [70, 71]]
  idx3 = [np.array([0, 4, 7]), np.array([1, 6]), np.array([2, 3, 5])]
  dat3 = [np.array([[0, 1], [40, 41], [70, 71]], np.int32), np.array([[10, 11], [60, 61]], np.int32),
          np.array([[20, 21], [30, 31], [50, 51]], np.int32)]
  cases.append(("simple_2d", idx3, dat3, want2d))
  # testZeroSizeTensor :112-130
  cases.append(("zero_size", idx3 + [np.zeros([0], np.int32)], dat3 + [np.zeros([0, 2], np.int32)], want2d))
  # testAllZeroSizeTensor :132-145
  cases.append(("all_zero_size", [np.zeros([0], np.int32)] * 2, [np.zeros([0, 2], np.int32)] * 2, np.zeros((0, 2))))
  # testHigherRank :147-163
  cases.append(("higher_rank", [np.array(6), np.array([4, 1]), np.array([[5, 2], [0, 3]])],
                [np.array([61., 62.], np.float32), np.array([[41., 42.], [11., 12.]], np.float32),
                 np.array([[[51., 52.], [21., 22.]], [[1., 2.], [31., 32.]]], np.float32)],
                10. * np.arange(7)[:, None] + [1., 2.]))
  return cases

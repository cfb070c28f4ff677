"""CPU tier (host logic): a Python model of the wave-parallel std::sort emulation used by k_cells
(lvt_amd/csrc/k_features.hip: wave_partition_pivot / wave_introsort_partitions + stable ranking) checked against the
oracle's real std::sort (libstdc++ introsort) on tie-heavy inputs, including the heap-sort fallback."""
import numpy as np
import pytest


def comp(a, b):           # handler.cpp:38-41: lhs.response > rhs.response
    return a[2] > b[2]


def partition_pivot(arr, first, last):
    mid = first + (last - first) // 2
    a, b, c = first + 1, mid, last - 1
    if comp(arr[a], arr[b]):
        pick = b if comp(arr[b], arr[c]) else (c if comp(arr[a], arr[c]) else a)
    elif comp(arr[a], arr[c]):
        pick = a
    elif comp(arr[b], arr[c]):
        pick = c
    else:
        pick = b
    arr[first], arr[pick] = arr[pick], arr[first]
    piv = arr[first]
    lo, hi = first + 1, last
    posL = [p for p in range(lo, hi) if not comp(arr[p], piv)]                 # ballot pass 1 (ascending)
    posR = [q for q in range(hi - 1, lo - 1, -1) if not comp(piv, arr[q])]     # ballot pass 2 (descending)
    K = min(len(posL), len(posR))
    m = 0
    while m < K and posL[m] < posR[m]:
        m += 1
    for k in range(m):                                                          # independent swaps
        p, q = posL[k], posR[k]
        arr[p], arr[q] = arr[q], arr[p]
    pm = posL[m] if m < len(posL) else 1 << 30
    qm1 = posR[m - 1] if m > 0 else last
    return min(pm, qm1)


def heap_sort(f):
    def adjust(hole, length, value):
        top = hole; child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if comp(f[child], f[child - 1]):
                child -= 1
            f[hole] = f[child]; hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            f[hole] = f[child - 1]; hole = child - 1
        parent = (hole - 1) // 2
        while hole > top and comp(f[parent], value):
            f[hole] = f[parent]; hole = parent; parent = (hole - 1) // 2
        f[hole] = value
    n = len(f)
    if n >= 2:
        parent = (n - 2) // 2
        while True:
            adjust(parent, n, f[parent])
            if parent == 0:
                break
            parent -= 1
    last = n
    while last > 1:
        last -= 1
        v = f[last]; f[last] = f[0]
        adjust(0, last, v)


def emulated_sort(items, depth_override=None):
    arr = list(items)
    n = len(arr)
    if n > 16:
        depth0 = 2 * (n.bit_length() - 1) if depth_override is None else depth_override
        stack = [(0, n, depth0)]
        while stack:
            first, last, depth = stack.pop()
            while last - first > 16:
                if depth == 0:
                    seg = arr[first:last]; heap_sort(seg); arr[first:last] = seg
                    break
                depth -= 1
                cut = partition_pivot(arr, first, last)
                if last - cut > 16:
                    stack.append((cut, last, depth))
                last = cut
    # final insertion sort == stable sort by key
    order = sorted(range(n), key=lambda i: (-arr[i][2], i))
    return [arr[i] for i in order]


@pytest.mark.parametrize("n,levels,seed", [(5, 3, 0), (16, 4, 1), (17, 2, 2), (64, 3, 3), (151, 8, 4), (300, 30, 5), (700, 60, 6),
                                           (1500, 200, 7), (3000, 5, 8), (999, 1, 9)])
def test_model_equals_std_sort(oracle_lib, n, levels, seed):
    rng = np.random.default_rng(seed)
    kp = np.column_stack([np.arange(n), rng.integers(0, 250, n), rng.integers(25, 25 + levels, n)]).astype(np.float32)
    exp = oracle_lib.sort_by_response(kp)
    got = np.array(emulated_sort([tuple(k) for k in kp]), np.float32)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("pattern", ["ascending", "descending", "organ", "sawtooth"])
def test_model_structured_inputs(oracle_lib, pattern):
    n = 1200
    i = np.arange(n)
    r = {"ascending": i // 7, "descending": (n - i) // 5, "organ": np.minimum(i, n - i) // 3, "sawtooth": i % 17}[pattern] + 25
    kp = np.column_stack([i, i, r]).astype(np.float32)
    exp = oracle_lib.sort_by_response(kp)
    got = np.array(emulated_sort([tuple(k) for k in kp]), np.float32)
    assert np.array_equal(got, exp)


def test_heap_fallback_is_a_valid_sort():
    """the depth-limit branch (std::__partial_sort) cannot be forced through std::sort from outside; check that the
    restated heap sort orders correctly and is only reached when the depth budget is exhausted"""
    rng = np.random.default_rng(3)
    items = [(i, 0, int(r)) for i, r in enumerate(rng.integers(25, 40, 500))]
    out = emulated_sort(items, depth_override=1)
    assert [o[2] for o in out] == sorted([it[2] for it in items], reverse=True)
    assert sorted(out) == sorted(items)

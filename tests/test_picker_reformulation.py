"""The peak picker's reformulation, position-level, in numpy — against the reference's sequential
scan (decode.rs:239-253) on random correlation signals with ties, plateaus, NaN runs and all-equal stretches.

What the kernels compute, restated without their group / chunk layout:
  terminals (k_sync_words)        T[i]: nothing in (i, i+md] exceeds corr[i]   (NaN read as -inf, corr[0] clamped to >= 0)
  node terminals (k_sync_slots)   heads of runs of terminals, terminals md+1 behind a terminal or a NaN, terminals on
                                  the grid; NaN positions a phase can start on travel with a tag
  orbit (k_sync_orbit_global)     successor of EVERY possible start (root, grid nodes, md+1 behind every node terminal),
                                  the root's orbit by doubling the known prefix, the peak list from the path
No GPU; the kernels themselves are compared with the oracle bit for bit in tests/test_gpu_*.py.
"""
import numpy as np
import pytest

f32 = np.float32


def reference_scan(corr, spr, md):
    """decode.rs:239-253, literally."""
    peaks = [(0, f32(0.0))]
    for i in range(len(corr)):
        c = corr[i]
        if i - peaks[-1][0] > md:
            while i // spr > len(peaks):
                peaks.append((i, c))
        elif c > peaks[-1][1]:
            peaks[-1] = (i, c)
    return [p[0] for p in peaks]


def terminal_masks(corr, md):
    c = np.array(corr, f32, copy=True)
    n = c.size
    nan = np.isnan(c)
    if n and not (c[0] > 0):
        c[0] = f32(0)
        nan[0] = False
    c[nan] = -np.inf
    t = np.zeros(n, bool)
    for i in range(n):
        w = c[i + 1:i + md + 1]
        t[i] = not (w.size and w.max() > c[i])
    return t, nan


def node_terminal_list(t, nan, spr, md):
    n = t.size
    idx = np.arange(n)
    prev_t = np.concatenate([[False], t[:-1]])
    back = idx - md - 1
    behind_t = np.where(back >= 0, t[np.maximum(back, 0)], False)
    behind_nan = np.where(back >= 0, nan[np.maximum(back, 0)], False)
    starts = behind_t | behind_nan | (idx % spr == 0)
    nw = t & ((~prev_t) | starts)
    ns = nan & starts & ~t
    pos = np.flatnonzero(nw | ns)
    return pos, ns[pos]


def orbit_all_nodes(pos, tagged, n_corr, spr, md):
    """Peak list through successors of all nodes + doubling from the root (the orbit kernel's default form)."""
    if n_corr == 0:
        return [0]
    kc = (n_corr - 1) // spr
    n_grid = max(kc - 1, 0)
    base = 1 + n_grid
    n_all = base + len(pos)
    END = n_all

    def start_of(i):
        if i == 0:
            return 0, 1
        if i < base:
            return (i + 1) * spr, i + 1
        sv = int(pos[i - base]) + md + 1
        return sv, sv // spr

    def first_nt(sv):
        j = int(np.searchsorted(pos, sv, side="left"))
        while j < len(pos) and tagged[j] and pos[j] != sv:
            j += 1
        assert j < len(pos), "fact 3: the last position is a node terminal"
        return j, int(pos[j])

    succ = np.full(n_all + 1, END, np.int64)
    reach = np.zeros(n_all + 1, np.int64)
    for i in range(n_all):
        sv, cell = start_of(i)
        if sv < n_corr:
            j, u = first_nt(sv)
            reach[i] = u
            a, b = u + md + 1, (cell + 1) * spr
            if max(a, b) < n_corr:
                succ[i] = base + j if a >= b else cell
    path_cap = kc + 2
    path = np.full(path_cap, END, np.int64)
    path[0] = 0
    jump = succ.copy()
    span = 1
    while span < path_cap:
        m = np.arange(min(span, path_cap - span))
        path[m + span] = jump[path[m]]
        jump = jump[jump]
        span *= 2
    peaks = {}
    c_prev = 1
    length = 1
    for k in range(path_cap):
        v = int(path[k])
        if v == END:
            break
        sv, cell = start_of(v)
        if k == 0:
            peaks[0] = int(reach[v])
            continue
        for q in range(c_prev, cell - 1):
            peaks[q] = sv
        peaks[cell - 1] = int(reach[v])
        c_prev = cell
        length = cell
    assert sorted(peaks) == list(range(length))
    return [peaks[q] for q in range(length)]


def _signals(rng, n, spr):
    kind = rng.integers(0, 6)
    x = rng.standard_normal(n).astype(f32)
    if kind == 0:    # sync-like: a peak per row, jittered, over noise
        for r in range(0, n, spr):
            p = r + int(rng.integers(0, spr))
            if p < n:
                x[p] += f32(rng.uniform(3, 9))
    elif kind == 1:  # quantised: many exact ties and plateaus
        x = np.round(x * 2).astype(f32)
    elif kind == 2:  # long constant stretches (every position a terminal there)
        for _ in range(3):
            a = int(rng.integers(0, n))
            x[a:a + int(rng.integers(1, 3 * spr))] = f32(rng.integers(-1, 2))
    elif kind == 3:  # NaN runs and isolated NaNs, some on the grid, some md+1 behind each other
        for _ in range(int(rng.integers(1, 12))):
            a = int(rng.integers(0, n))
            x[a:a + int(rng.integers(1, 8))] = np.nan
        x[::spr][rng.random(x[::spr].size) < 0.3] = np.nan
    elif kind == 4:  # decaying / rising ramps: long runs of terminals, or none
        x = (np.linspace(1, -1, n) * rng.choice([-1, 1]) * 5 + x * 0.01).astype(f32)
    else:            # +-inf samples
        x[rng.random(n) < 0.01] = np.inf
        x[rng.random(n) < 0.01] = -np.inf
    return x


@pytest.mark.parametrize("seed", range(40))
def test_all_nodes_orbit_equals_the_sequential_scan(seed):
    rng = np.random.default_rng(1000 + seed)
    spr = int(rng.choice([20, 40, 65, 130]))
    md = spr * 8 // 10
    n = int(rng.integers(1, 14 * spr))
    corr = _signals(rng, n, spr)
    want = reference_scan(corr, spr, md)
    t, nan = terminal_masks(corr, md)
    pos, tagged = node_terminal_list(t, nan, spr, md)
    got = orbit_all_nodes(pos, tagged, n, spr, md)
    assert got == want, (seed, spr, md, n)


def test_edge_cases():
    for corr, spr in (([], 20), ([1.0], 20), ([np.nan], 20), ([0.0] * 100, 20), ([np.nan] * 100, 20),
                      (list(range(100)), 20), (list(range(100, 0, -1)), 20)):
        corr = np.array(corr, f32)
        md = spr * 8 // 10
        want = reference_scan(corr, spr, md)
        if corr.size == 0:
            assert want == [0]
            continue
        t, nan = terminal_masks(corr, md)
        pos, tagged = node_terminal_list(t, nan, spr, md)
        assert orbit_all_nodes(pos, tagged, corr.size, spr, md) == want, (corr[:5], spr)

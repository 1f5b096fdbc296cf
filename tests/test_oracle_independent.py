"""A second, independent restatement of the motion search -- a plain NumPy loop
written directly from SURVEY.md 2.1 -- against the C oracle on random small
cases (hypothesis).  Guards the oracle beyond the fixed golden vectors."""
import numpy as np
from hypothesis import given, settings, strategies as st


def me_numpy(cur, ref, bs, win, step, thresh):
    H, W = cur.shape[:2]
    out, n_static = [], 0
    for y in range(0, H - bs + 1, bs):
        for x in range(0, W - bs + 1, bs):
            c = cur[y:y + bs, x:x + bs].astype(np.int64)
            r0 = ref[y:y + bs, x:x + bs].astype(np.int64)
            if np.maximum(r0 - c, 0).sum() <= thresh:           # saturating subtract
                out.append((0, 0)); n_static += 1
                continue
            best, bxy = 9999999999, (0, 0)
            i_min, j_min = max(y - win, 0), max(x - win, 0)
            i_max, j_max = min(y + win, H), min(x + win, W)
            for i in range(i_min, i_max, step):
                if i + bs >= i_max:
                    continue
                for j in range(j_min, j_max, step):
                    if j + bs >= j_max:
                        continue
                    d = ((ref[i:i + bs, j:j + bs].astype(np.int64) - c) % 256).sum()  # wrap-around
                    if d < best:
                        best, bxy = d, (j, i)
            out.append((bxy[0] - x, bxy[1] - y))
    return np.array(out, np.int16).reshape(-1, 2), n_static


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**31 - 1), st.sampled_from([4, 6, 8, 16]), st.integers(16, 56),
       st.integers(16, 72), st.sampled_from([0, 300, 2000, 10**6]))
def test_oracle_me_matches_numpy_loop(oracle, seed, bs, H, W, thresh):
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    kind = seed % 3
    if kind == 0:
        cur = np.roll(ref, (seed % 5 - 2, seed % 7 - 3), axis=(0, 1))
    elif kind == 1:
        cur = np.clip(ref.astype(np.int16) + rng.integers(-3, 4, ref.shape), 0, 255).astype(np.uint8)
    else:
        cur = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    win, step = 2 * bs, round(bs / 3)
    mv, ns = oracle.me(cur, ref, bs, thresh=thresh)
    want, wns = me_numpy(cur, ref, bs, win, step, thresh)
    assert np.array_equal(mv, want) and ns == wns

# numpy model of k_quads_hash's candidate enumeration against the all-pairs test of k_quads (cell adjacency part)
import numpy as np
rng = np.random.default_rng(0)
SLOTS = 4096
def h(cx, cy, cz):
    return ((np.uint32(cx) * np.uint32(73856093)) ^ (np.uint32(cy) * np.uint32(19349663)) ^ (np.uint32(cz) * np.uint32(83492791))) & np.uint32(SLOTS - 1)
for trial in range(20):
    eg = int(rng.integers(5, 40))
    n1, n2 = int(rng.integers(1, 3000)), int(rng.integers(1, 800))
    E = rng.integers(-2, eg + 2, (n1, 3)); nid = rng.integers(-3, 350, n1)
    Q = rng.integers(-2, eg + 2, (n2, 3))
    evalid = ((E >= 0) & (E < eg)).all(1) & (nid >= 0) & (nid < 343)
    qvalid = ((Q >= 0) & (Q < eg)).all(1)
    brute = set()
    for j in range(n2):
        if not qvalid[j]: continue
        d = E - Q[j]
        ok = evalid & (np.abs(d) <= 1).all(1)
        brute |= {(int(i), j) for i in np.nonzero(ok)[0]}
    head = -np.ones(SLOTS, int); nxt = -np.ones(n1, int)
    with np.errstate(over="ignore"):
        for i in range(n1):
            if evalid[i]:
                s = int(h(*E[i])); nxt[i] = head[s]; head[s] = i
        got = set()
        for j in range(n2):
            if not qvalid[j]: continue
            for nb in range(27):
                c = Q[j] + np.array([nb % 3 - 1, (nb // 3) % 3 - 1, nb // 9 - 1])
                if not ((c >= 0) & (c < eg)).all(): continue
                cur = head[int(h(*c))]
                while cur >= 0:
                    if (E[cur] == c).all(): got.add((int(cur), j))
                    cur = nxt[cur]
    assert got == brute, (trial, len(got), len(brute))
print("hash enumeration == all-pairs adjacency on 20 random cases")

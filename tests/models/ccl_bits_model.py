"""CPU model of the bit-plane tile kernel k_ccl_bits (moving_object_detector_amd/csrc/cluster.hip): the per-tile steps on
Python integers, word for word what the kernel does on (h, a, b) register triples with lane = grid row.

A tile whose dynamic cells all pass each other's depth gate has the connectivity of its MASK alone
(clusterer_nodelet.cpp:56-83,186-219 with `depthDiff` never firing): p ~ q iff both dynamic and q lies in p's up-left
(n+1)x(n+1) window, n = neighbor_distance (1 .. 10; configure(n)).  Grid = n halo rows above + 16 tile rows, n halo columns left + 64;
a row is one 96-bit integer, column c (c = -n .. 63) at bit c + 32.

TEST INFRASTRUCTURE (tests/test_ccl_bits_model.py): checked against brute-force components of the same graph."""
import random

TH = 16
BITS = 96
FULL = (1 << BITS) - 1
INTERIOR = ((1 << 64) - 1) << 32      # columns 0 .. 63


def configure(n):
    """neighbor_distance n: n halo rows above, n halo columns to the left (the kernel is a template on it, 1 .. 10)."""
    global N, ROWS, GRID, HALO_COLS
    N = n
    ROWS = TH + n
    GRID = ((1 << (64 + n)) - 1) << (32 - n)      # columns -n .. 63
    HALO_COLS = ((1 << n) - 1) << (32 - n)


configure(4)


def shl(v, k):                        # towards larger x
    return (v << k) & FULL


def shr(v, k):                        # towards smaller x; bits left of column -4 are not part of the grid
    return (v >> k) & GRID


def _grow(v, sh):                     # v | sh(v, 1) | .. | sh(v, N - 1) by doubling: the covered range grows by min(covered + 1, rest)
    y, cov = v, 0
    while cov < N - 1:
        s = min(cov + 1, N - 1 - cov)
        y = y | sh(y, s)
        cov += s
    return y


def dil_r(v, lo=0):                   # v | v<<1 | .. | v<<N (lo = 1: without v itself)
    x = shl(_grow(v, shl), 1)         # 1 .. N
    return x if lo else v | x


def dil_l(v, lo=0):
    x = shr(_grow(v, shr), 1)
    return x if lo else v | x


def row_dn(rows):                     # value of the row above arrives (wave_shr:1, zero into row 0)
    return [0] + rows[:-1]


def row_up(rows):                     # value of the row below arrives (wave_shl:1, zero into the last row)
    return rows[1:] + [0]


def orr(a, b):
    return [x | y for x, y in zip(a, b)]


def vert4(rows, dn):                  # OR over dv = 1..N of the rows dv above (dn) / below (not dn), by doubling
    sh1 = row_dn if dn else row_up

    def sh(r, k):
        for _ in range(k):
            r = sh1(r)
        return r
    y, cov = rows, 0
    while cov < N - 1:
        s = min(cov + 1, N - 1 - cov)
        y = orr(y, sh(y, s))
        cov += s
    return sh1(y)                                 # dv 1..N


def brev(v):
    return int(format(v, "096b")[::-1], 2)


def fill(c, s):
    """All bits of the runs of c that hold a bit of s (s subset of c): carry trick upwards, the same on the reversed words
    downwards."""
    def up(c, s):
        t = (c + s) & FULL
        return ((t ^ c) & c) | s
    return up(c, s) | brev(up(brev(c), brev(s)))


def closed(m):
    """m with the gaps of <= 3 cells between two dynamic cells filled: the cells of one run of the result are chained by
    same-row links (dv = 0, k <= 4).  A cell is in the result iff a dynamic cell lies i to its left and one j to its right
    with i + j <= 4."""
    l = [m]                                       # l[i]: a dynamic cell within i to the left
    for i in range(1, N + 1):
        l.append(l[-1] | shl(m, i))
    c = 0
    for j in range(N + 1):
        c |= l[N - j] & (shr(m, j) if j else m)
    return c & GRID


def analyse(M):
    DR = [dil_r(m) for m in M]
    DL = [dil_l(m) for m in M]
    ul = [m & (dil_r(m, 1) | v) for m, v in zip(M, vert4(DR, True))]      # has an up-left edge
    dr = [m & (dil_l(m, 1) | v) for m, v in zip(M, vert4(DL, False))]     # is somebody's up-left neighbour
    C = [closed(m) for m in M]
    return ul, dr, C


def flood(M, C, seed_row, seed_bit):
    S = [0] * ROWS
    S[seed_row] = seed_bit
    sweeps = 0
    while True:
        DR = [dil_r(s) for s in S]
        DL = [dil_l(s) for s in S]
        new = [m & (a | b | c | d) for m, a, b, c, d in zip(M, DR, DL, vert4(DR, True), vert4(DL, False))]
        S2 = [fill(c, x) & m for c, x, m in zip(C, new, M)]
        sweeps += 1
        if S2 == S:
            return S, sweeps
        S = S2


def single_component_check(M, ul, dr, C):
    """Sufficient test for 'the dynamic cells of the grid are ONE component': every non-empty row is one closed run of cells that all
    have an edge, and every non-empty row but the first has a link to a row above."""
    ne = [m != 0 for m in M]
    if not any(ne):
        return False
    first = ne.index(True)
    vd = vert4([dil_r(m) for m in M], True)
    for r in range(ROWS):
        if not ne[r]:
            continue
        starts = C[r] & ~shl(C[r], 1)
        if bin(starts).count("1") != 1:
            return False
        if (ul[r] | dr[r]) != M[r]:
            return False
        if r != first and (M[r] & vd[r]) == 0:
            return False
    return True


def tile(M, use_check=True):
    """M: 20 row integers.  Returns (parent {(r, c) -> (r, c)} over interior cells, size / key per root, requests set of
    (halo cell, root), stats)."""
    M = [m & GRID for m in M]
    ul, dr, C = analyse(M)
    parent, size, key, reqs = {}, {}, {}, set()
    stats = {"flood": 0, "sweeps": 0, "check": False}
    inter_rows = range(ROWS - TH, ROWS)

    def cells(v):
        return [b - 32 for b in range(32 - N, 96) if (v >> b) & 1]

    # singletons: interior cells without any edge
    for r in inter_rows:
        z = M[r] & INTERIOR & ~ul[r] & ~dr[r]
        for c in cells(z):
            parent[(r, c)] = (r, c); size[(r, c)] = 1; key[(r, c)] = None
    R = [(M[r] & INTERIOR & (ul[r] | dr[r])) if r >= ROWS - TH else 0 for r in range(ROWS)]

    def emit(S):
        rr = next(r for r in inter_rows if S[r] & INTERIOR)
        rc = cells(S[rr] & INTERIOR)[0]
        root = (rr, rc)
        n = 0; k = None
        for r in inter_rows:
            for c in cells(S[r] & INTERIOR):
                parent[(r, c)] = root; n += 1
            if k is None and S[r] & INTERIOR & ul[r]:
                k = (r, cells(S[r] & INTERIOR & ul[r])[0])
        size[root] = n; key[root] = k
        HS = [S[r] if r < ROWS - TH else S[r] & HALO_COLS for r in range(ROWS)]
        up = row_dn(HS)
        for r in range(ROWS):
            cov = HS[r] & (shl(HS[r], 1) | up[r])
            for c in cells(HS[r] & ~cov):
                reqs.add(((r, c), root))

    if use_check and any(R) and single_component_check(M, ul, dr, C):
        stats["check"] = True
        emit(M)
        return parent, size, key, reqs, stats
    while any(R):
        rr = next(r for r in range(ROWS) if R[r])
        sb = R[rr] & -R[rr]
        S, sw = flood(M, C, rr, sb)
        stats["flood"] += 1; stats["sweeps"] += sw
        emit(S)
        R = [x & ~s for x, s in zip(R, S)]
    return parent, size, key, reqs, stats


# ---- brute force ------------------------------------------------------------------------------------------------------------
def brute(M):
    dyn = {(r, c) for r in range(ROWS) for c in range(-N, 64) if (M[r] >> (c + 32)) & 1}
    par = {p: p for p in dyn}

    def find(p):
        while par[p] != p:
            par[p] = par[par[p]]; p = par[p]
        return p
    hasul = set()
    for (r, c) in dyn:
        for dv in range(N + 1):
            for k in range(N + 1):
                if dv == 0 and k == 0:
                    continue
                q = (r - dv, c - k)
                if q in dyn:
                    hasul.add((r, c))
                    a, b = find((r, c)), find(q)
                    if a != b:
                        par[max(a, b)] = min(a, b)
    comps = {}
    for p in dyn:
        comps.setdefault(find(p), []).append(p)
    return dyn, comps, hasul


def check(M):
    M = [m & GRID for m in M]
    dyn, comps, hasul = brute(M)
    for uc in (False, True):
        parent, size, key, reqs, stats = tile(M, uc)
        want_parent, want_size, want_key, want_req = {}, {}, {}, set()
        for members in comps.values():
            inter = sorted(p for p in members if p[0] >= ROWS - TH and p[1] >= 0)
            if not inter:
                continue
            root = inter[0]
            for p in inter:
                want_parent[p] = root
            want_size[root] = len(inter)
            ks = [p for p in inter if p in hasul]
            want_key[root] = ks[0] if ks else None
            ms = set(members)
            for (r, c) in members:
                if r >= ROWS - TH and c >= 0:
                    continue
                if (r, c - 1) in ms and not (r >= ROWS - TH and c - 1 >= 0):
                    continue
                if (r - 1, c) in ms:
                    continue
                want_req.add(((r, c), root))
        assert parent == want_parent, ("parent", uc)
        assert size == want_size, ("size", uc)
        assert key == want_key, ("key", uc, key, want_key)
        assert reqs == want_req, ("requests", uc, sorted(reqs ^ want_req))
    return stats


def random_mask(rng, kind):
    M = [0] * ROWS
    if kind == "noise":
        d = rng.choice([0.02, 0.1, 0.3, 0.6, 0.95])
        for r in range(ROWS):
            for c in range(-N, 64):
                if rng.random() < d:
                    M[r] |= 1 << (c + 32)
    elif kind in ("blobs", "oneblob"):
        for _ in range(rng.randint(1, 4) if kind == "blobs" else 1):
            x0, x1 = sorted(rng.sample(range(-N, 64), 2)); y0, y1 = sorted(rng.sample(range(0, ROWS), 2))
            for r in range(y0, y1 + 1):
                for c in range(x0, x1 + 1):
                    if rng.random() < 0.94:
                        M[r] |= 1 << (c + 32)
        for _ in range(rng.randint(0, 3) if kind == "blobs" else 0):
            M[rng.randrange(ROWS)] |= 1 << (rng.randrange(-N, 64) + 32)
    else:   # stripes: vertical / horizontal bars with gaps around the window size
        gap = rng.randint(max(N - 1, 2), N + 2)
        for r in range(ROWS):
            for c in range(-N, 64):
                if (kind == "vbars" and (c + N) % gap == 0) or (kind == "hbars" and r % gap == 0):
                    if rng.random() < 0.9:
                        M[r] |= 1 << (c + 32)
    if rng.random() < 0.3:
        for r in range(N):
            M[r] = 0                      # image top
    if rng.random() < 0.3:
        M = [m & ~HALO_COLS for m in M]   # image left
    return M


def run(seed, cases=150, n=4):
    configure(n)
    rng = random.Random(seed)
    agg = {"check": 0, "flood": 0, "sweeps": 0, "tiles": 0}
    for i in range(cases):
        M = random_mask(rng, rng.choice(["noise", "blobs", "oneblob", "oneblob", "vbars", "hbars"]))
        if not any(m & INTERIOR for m in M[ROWS - TH:]):
            continue
        st = check(M)
        agg["tiles"] += 1; agg["check"] += st["check"]; agg["flood"] += st["flood"]; agg["sweeps"] += st["sweeps"]
    return agg


if __name__ == "__main__":
    import sys
    print(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 300, int(sys.argv[3]) if len(sys.argv) > 3 else 4))

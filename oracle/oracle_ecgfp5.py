"""TEST ORACLE (not product code): ecGFp5 in textbook affine arithmetic, pure Python integers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The reference takes its curve from the `pod2` crate (pinned rev cdf227e3, Cargo.toml:14), which is NOT in the reference
tree: ecgfp5/src/lib.rs:12 imports `Point` and `GROUP_ORDER`, and nothing in-tree holds curve constants, generator
coordinates or test vectors.  This file therefore restates the PUBLISHED curve ("ecGFp5", T. Pornin,
2022; "Double-odd elliptic curves", 2020):

    K  = GF(p)[z] / (z^5 - 3),  p = 2^64 - 2^32 + 1
    E  : y^2 = x (x^2 + a x + b),  a = 2,  b = 263 z          (|E| = 2 n, n an odd 319-bit prime)
    group = the points of E that are NOT of n-torsion, i.e. N + E[n] with N = (0, 0); the neutral element is N and the
            law is  P (+) Q = P + Q + N  (ordinary curve additions).  A group element is written (x, u) with u = x / y
            (u = 0 for N); it compresses to the single field element u and x is recovered from
            x^2 + (a - 1/u^2) x + b = 0, the root that is a non-square.
    generator: u = 1/4  (w = y/x = 4)

Parity with pod2's byte-level conventions (which coordinate `compress_from_subgroup` returns, how
`new_rand_from_subgroup` samples) is UNPINNED: no fixture in the reference covers them.  What IS checked here and in
tests/test_ecgfp5.py: n is prime, n * G = neutral, every formula of the product (projective, complete) against the
chord-and-tangent law below on random points, and the reference's own round-trip tests
(ecgfp5/src/lib.rs:101-112, elgamal.rs:33-46, hashed_elgamal.rs:49-63).
"""
P = 2**64 - 2**32 + 1
A = (2, 0, 0, 0, 0)
B = (0, 263, 0, 0, 0)
ZERO = (0, 0, 0, 0, 0)
ONE = (1, 0, 0, 0, 0)
GROUP_ORDER = 1067993516717146951041484916571792702745057740581727230159139685185762082554198619328292418486241


def f_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def f_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def f_neg(a):
    return tuple((-x) % P for x in a)


def f_mul(a, b):
    c = [0] * 9
    for i in range(5):
        for j in range(5):
            c[i + j] += a[i] * b[j]
    return tuple((c[k] + 3 * (c[k + 5] if k + 5 < 9 else 0)) % P for k in range(5))


def f_small(a, k):
    return tuple(x * k % P for x in a)


def f_pow(a, e):
    r = ONE
    while e:
        if e & 1:
            r = f_mul(r, a)
        a = f_mul(a, a)
        e >>= 1
    return r


Q5 = P**5
GAMMA = pow(3, (P - 1) // 5, P)  # z^p = GAMMA * z


def f_frob(a, k=1):
    """a^(p^k): coefficient i is scaled by GAMMA^(i k)."""
    g = pow(GAMMA, k, P)
    return tuple(a[i] * pow(g, i, P) % P for i in range(5))


def f_inv(a):
    """a^-1 = a^(r-1) / Norm(a),  r = 1 + p + p^2 + p^3 + p^4,  Norm(a) = a^r in GF(p)."""
    if a == ZERO:
        raise ZeroDivisionError("inverse of zero in GF(p^5)")
    t = f_mul(f_frob(a, 1), f_frob(a, 2))  # a^(p + p^2)
    t = f_mul(t, f_frob(t, 2))  # a^(p + p^2 + p^3 + p^4)
    norm = f_mul(a, t)
    assert norm[1:] == (0, 0, 0, 0)
    return f_small(t, pow(norm[0], P - 2, P))


def f_is_square(a):
    return a == ZERO or f_pow(a, (Q5 - 1) // 2) == ONE


def f_sqrt(a):
    """Tonelli-Shanks in GF(p^5) (2-adicity 32); None when a is not a square."""
    if a == ZERO:
        return ZERO
    if not f_is_square(a):
        return None
    s, q = 32, (Q5 - 1) >> 32
    zz = (0, 1, 0, 0, 0)
    # a non-residue: try small elements
    g = None
    for k in range(1, 50):
        cand = (k, 1, 0, 0, 0)
        if not f_is_square(cand):
            g = cand
            break
    del zz
    c = f_pow(g, q)
    x = f_pow(a, (q + 1) // 2)
    t = f_pow(a, q)
    m = s
    while t != ONE:
        i, t2 = 0, t
        while t2 != ONE:
            t2 = f_mul(t2, t2)
            i += 1
        bb = c
        for _ in range(m - i - 1):
            bb = f_mul(bb, bb)
        x = f_mul(x, bb)
        c = f_mul(bb, bb)
        t = f_mul(t, c)
        m = i
    return x


# ---- curve points of E in affine (x, y); None is the point at infinity ---------------------------------------------
def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    rhs = f_mul(x, f_add(f_add(f_mul(x, x), f_mul(A, x)), B))
    return f_mul(y, y) == rhs


def e_neg(pt):
    return None if pt is None else (pt[0], f_neg(pt[1]))


def e_add(p1, p2):
    """Chord and tangent on y^2 = x^3 + a x^2 + b x."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if f_add(y1, y2) == ZERO:
            return None
        lam = f_mul(f_add(f_add(f_small(f_mul(x1, x1), 3), f_small(f_mul(A, x1), 2)), B), f_inv(f_small(y1, 2)))
    else:
        lam = f_mul(f_sub(y2, y1), f_inv(f_sub(x2, x1)))
    x3 = f_sub(f_sub(f_sub(f_mul(lam, lam), A), x1), x2)
    y3 = f_sub(f_mul(lam, f_sub(x1, x3)), y1)
    return (x3, y3)


N_PT = (ZERO, ZERO)  # the point of order two; neutral element of the group


# ---- group elements (x, u) --------------------------------------------------------------------------------------
NEUTRAL = (ZERO, ZERO)


def to_curve(g):
    x, u = g
    if u == ZERO:
        return N_PT
    return (x, f_mul(x, f_inv(u)))


def from_curve(pt):
    if pt == N_PT:
        return NEUTRAL
    x, y = pt
    return (x, f_mul(x, f_inv(y)))


def g_add(g1, g2):
    return from_curve(e_add(e_add(to_curve(g1), to_curve(g2)), N_PT))


def g_neg(g):
    return (g[0], f_neg(g[1]))


def g_mul(k, g):
    acc = NEUTRAL
    for bit in bin(k)[2:] if k else "":
        acc = g_add(acc, acc)
        if bit == "1":
            acc = g_add(acc, g)
    return acc


def in_group(g):
    """On the curve and not of n-torsion: x is a non-square (or the neutral)."""
    x, u = g
    if g == NEUTRAL:
        return True
    if u == ZERO or x == ZERO:
        return False
    # x = u^2 (x^2 + a x + b)
    if f_mul(f_mul(u, u), f_add(f_add(f_mul(x, x), f_mul(A, x)), B)) != x:
        return False
    return not f_is_square(x)


def compress(g):
    return g[1]


def decompress(u):
    """Group element with the given u, or None."""
    if u == ZERO:
        return NEUTRAL
    iu2 = f_inv(f_mul(u, u))
    bcoef = f_sub(A, iu2)  # x^2 + bcoef x + b = 0
    disc = f_sub(f_mul(bcoef, bcoef), f_small(B, 4))
    r = f_sqrt(disc)
    if r is None:
        return None
    half = (P + 1) // 2
    for s in (r, f_neg(r)):
        x = f_small(f_sub(s, bcoef), half)
        if in_group((x, u)):
            return (x, u)
    return None


GENERATOR = None


def generator():
    global GENERATOR
    if GENERATOR is None:
        GENERATOR = decompress((pow(4, P - 2, P), 0, 0, 0, 0))
    return GENERATOR


# ---- the reference's schemes ------------------------------------------------------------------------------------
def elgamal_encrypt(pk, nonce, msg):  # ecgfp5/src/elgamal.rs:11-16
    return g_mul(nonce, generator()), g_add(msg, g_mul(nonce, pk))


def elgamal_decrypt(sk, ct):  # ecgfp5/src/elgamal.rs:19-22
    return g_add(ct[1], g_neg(g_mul(sk, ct[0])))


def as_fields(g):
    return list(g[0]) + list(g[1])

"""Parameter sets of the reference, regenerated rather than copied.

* ``Pi60`` / ``Qi60`` (ring/params.go:28-69): the first hundred primes = 1 mod 2^17 upward from
  2^59, and the first hundred primes = 1 mod 2^18 downward from 2^60.
* ``DefaultParamsQi/Pi[logN]`` (ring/params.go:10-25): the benchmark rings R12..R16.
* CKKS / BFV default moduli: ``GenerateNTTPrimes`` (ring/utils.go:133-175) handed out per
  bit-size in the order Qi, Pi (, QiMul) as ``GenModuli`` does (ckks/utils.go:150-193,
  bfv/utils.go:26-85); parameter tables ckks/params.go:36-87, bfv/params.go:47-88.
"""
from functools import lru_cache


def is_prime(n):
    """Deterministic Miller-Rabin for n < 2^64."""
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _primes_from(start, step, count):
    out, x = [], start
    while len(out) < count:
        if is_prime(x):
            out.append(x)
        x += step
    return out


@lru_cache(maxsize=None)
def Pi60():
    return tuple(_primes_from((1 << 59) + 1, 1 << 17, 100))


@lru_cache(maxsize=None)
def Qi60():
    return tuple(_primes_from((1 << 60) + 1 - (1 << 18), -(1 << 18), 100))


_RING_LIMBS = {12: 2, 13: 4, 14: 8, 15: 16, 16: 32}


def DefaultParamsQi(logN):
    """(N, moduli) of ring.DefaultParamsQi[logN] (ring/params.go:10-16)."""
    k = _RING_LIMBS[logN]
    return 1 << logN, list(Qi60()[-k:])


def DefaultParamsPi(logN):
    k = _RING_LIMBS[logN]
    return 1 << logN, list(Pi60()[-k:])


def GenerateNTTPrimes(logQ, logN, levels):
    """ring.GenerateNTTPrimes (ring/utils.go:133-175); the downward branch (:161) is dead for real sizes."""
    if logQ > 60:
        raise ValueError("logQ must be between 1 and 60")
    two_n = 2 << logN
    x = y = (1 << logQ) + 1
    primes = []
    while True:
        if is_prime(x):
            primes.append(x)
            if len(primes) == levels:
                return primes
        x += two_n
        if two_n > y:
            y -= two_n
            if is_prime(y):
                primes.append(y)
                if len(primes) == levels:
                    return primes


def _gen_moduli(logN, *bit_lists):
    """GenModuli: count primes per bit-size over all lists, generate once per size, hand out in list order."""
    need = {}
    for bits in bit_lists:
        for b in bits:
            need[b] = need.get(b, 0) + 1
    pool = {b: GenerateNTTPrimes(b, logN, n) for b, n in need.items()}
    out = []
    for bits in bit_lists:
        cur = []
        for b in bits:
            cur.append(pool[b].pop(0))
        out.append(cur)
    return out


# ckks/params.go:36-87 (LogQi, LogPi)
CKKS_DEFAULT = {
    "PN12QP109": (12, [37, 32], [38]),
    "PN13QP218": (13, [33, 30, 30, 30, 30, 30], [35]),
    "PN14QP438": (14, [45] + [34] * 9, [43, 43]),
    "PN15QP880": (15, [50] + [40] * 17, [50, 50, 50]),
    "PN16QP1761": (16, [55] + [45] * 33, [55, 55, 55, 55]),
}

# bfv/params.go:47-88 (LogQi, LogPi, LogQiMul), t = 65537
BFV_DEFAULT = {
    "PN12QP109": (12, [39, 39], [30], [60, 60]),
    "PN13QP218": (13, [54, 54, 54], [55], [60, 60, 60]),
    "PN14QP438": (14, [56, 55, 55, 54, 54, 54], [55, 55], [60] * 6),
    "PN15QP880": (15, [59, 59, 59, 58, 58, 58, 58, 58, 58, 58, 58, 58], [60, 60, 60], [60] * 12),
}


@lru_cache(maxsize=None)
def ckks_moduli(name):
    """(N, Q, P) of ckks.DefaultParams[name]."""
    logN, lq, lp = CKKS_DEFAULT[name]
    Q, P = _gen_moduli(logN, lq, lp)
    return 1 << logN, Q, P


@lru_cache(maxsize=None)
def bfv_moduli(name):
    """(N, Q, P, QMul) of bfv.DefaultParams[name]."""
    logN, lq, lp, lm = BFV_DEFAULT[name]
    Q, P, QMul = _gen_moduli(logN, lq, lp, lm)
    return 1 << logN, Q, P, QMul

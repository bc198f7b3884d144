"""Philox4x32-10 in plain Python (test infrastructure): the statement pg_uniform_f32 is compared with."""


def _philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in plain Python."""
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xffffffff, p1 & 0xffffffff, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xffffffff, p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return c



"""Counter-based random streams of the build, restated in NumPy.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference draws from Julia RNGs (`randn(rng)` for the exploration noise, src/PDEagent.jl:201; `rand(rng, 1:n, k)` for
the minibatch indices, :317-321; `rand(Uniform(-1, 1), n)` for the random initial conditions, KSSetup.jl:288-298) whose
streams cannot be reproduced outside Julia -- only their distributions.  The build replaces them by Philox4x32-10
(Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the published algorithm, constants
0xD2511F53 / 0xCD9E8D57 and Weyl keys 0x9E3779B9 / 0xBB67AE85) so that the device needs no generator state; this file
restates that stream bit for bit, which is what the GPU sampling / noise / initialiser kernels are checked against.
PINNED by the known-answer vectors of the Random123 distribution (tests/test_oracle.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32(counter, seed, rounds=10):
    """counter: uint64 array [n] (low 64 bits of the 128-bit counter, the high words are 0); seed: python int (64 bit).
    Returns uint32 [n, 4]."""
    ctr = np.asarray(counter, dtype=np.uint64)
    c0, c1 = ctr & MASK, ctr >> np.uint64(32)
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(rounds):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def words(seed, offset, n):
    """the first n 32-bit words of the stream (seed, offset): word k = word k % 4 of counter offset + k // 4"""
    nb = (n + 3) // 4
    w = philox4x32(np.uint64(offset) + np.arange(nb, dtype=np.uint64), seed)
    return w.reshape(-1)[:n]


def randn(seed, offset, n):
    """pdec_randn / the acting kernels: Box-Muller on pairs of words, u = (word + 0.5) / 2^32, in float64"""
    nb = (n + 3) // 4
    w = philox4x32(np.uint64(offset) + np.arange(nb, dtype=np.uint64), seed).astype(np.float64)
    u = (w + 0.5) / 4294967296.0
    out = np.empty((nb, 4))
    for h in range(2):
        rad = np.sqrt(-2.0 * np.log(u[:, 2 * h]))
        ang = 6.283185307179586 * u[:, 2 * h + 1]
        out[:, 2 * h] = rad * np.cos(ang)
        out[:, 2 * h + 1] = rad * np.sin(ang)
    return out.reshape(-1)[:n]


def sample_slots(seed, offset, count, n_valid, n_rt, capacity, stride):
    """pde_sample (src/PDEagent.jl:317-321) as the device draws it: ind = (word * (n_valid - stride)) >> 32 in
    [0, n_valid - stride); logical index max(0, n_rt - capacity) + ind; returns int64 [3, count] = slots of (s, a),
    of (r, t) and of s' (row + stride)."""
    hi = n_valid - stride
    w = words(seed, offset, count).astype(np.uint64)
    ind = ((w * np.uint64(hi)) >> np.uint64(32)).astype(np.int64)
    lg = max(0, n_rt - capacity) + ind
    cap1 = capacity + stride
    return np.stack([lg % cap1, lg % capacity, (lg + stride) % cap1])


def random_init_coefficients(seed, offset, B, nc):
    """pdec_env_random_init: nc uniforms in (-1, 1) per trajectory from ceil(nc / 4) counters each, normalised to |a| = 1"""
    nblk = (nc + 3) // 4
    ctr = np.uint64(offset) + np.arange(B * nblk, dtype=np.uint64)
    w = philox4x32(ctr, seed).astype(np.float64).reshape(B, nblk * 4)[:, :nc]
    a = 2.0 * ((w + 0.5) / 4294967296.0) - 1.0
    return a / np.linalg.norm(a, axis=1, keepdims=True)

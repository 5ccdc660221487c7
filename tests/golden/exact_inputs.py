"""Loop inputs (W, Xxt_damped, L, diag(Hinv), T0) that every box rebuilds BIT FOR BIT from a seed.  TEST INFRASTRUCTURE.

The large golden cases (tests/golden/large/*.npz, made by make_golden_large.py) record what the reference's own
`GANQ._perform_quantization_loop` (ganq.py:456-646) returns at sizes whose inputs are too big to commit (L and H are
n x n fp32: 2 x 16 MB at n = 2048, 2 x 64 MB at 4096).  So the inputs are generated, here and on the GPU box, by
arithmetic whose result does not depend on the machine:

  * calibration activations are small integers times a power of two per feature: X[t,f] = (Z[t,f] + Z[t,f-1]) * 2^e_f,
    Z in [-7, 7], e_f in {-2..1}; every sum of products is an integer below 2^24 times a power of two, so
    H = (2/N) X^T X is EXACT in fp32 (and in the fp64 BLAS product that forms it) whatever the summation order;
  * act_sort="asc" permutation by a stable sort of diag(H); ganq-style offset sum_j |H_ij| - 2 H_ii and the damping
    0.01 * mean(diag H) are exact fp64 sums rounded once (gptq.py:281-300);
  * L = factor of H + diag(offset), and diag(Hinv) = flipped 1 / diag(factor of the index-reversed damped H), by
    oracle.c_oracle.det_cholesky: one ascending fp64 sum per entry (a LAPACK factor changes with blocking / threads /
    instruction set and could not be rebuilt bit for bit);
  * W = fp16-rounded sums of four uniform integers * 2^-16 (bell-shaped, std 0.018; numpy's PCG64 integer stream only:
    no libm call anywhere); T0 = the oracle's exact k-means (stored in the fixture as well).
The fixture stores the sha256 of every input; `check()` fails loudly if a box ever disagrees.
"""
import hashlib

import numpy as np

from oracle import c_oracle


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


_base_cache = {}


def _base(m: int, n: int, seed: int, tokens: int):
    """everything but T0 (which depends on `bits`); the last result is kept: cases that differ only in `bits` share it"""
    key = (m, n, seed, tokens)
    if key in _base_cache:
        return _base_cache[key]
    _base_cache.clear()
    rng = np.random.default_rng(seed)
    Z = rng.integers(-7, 8, size=(tokens, n)).astype(np.float64)
    X = Z + np.roll(Z, 1, axis=1)                       # neighbouring features correlate (0.5)
    del Z
    e = rng.integers(-2, 2, size=n)
    G = X.T @ X                                          # integers < 2^22: exact in fp64 whatever the BLAS does
    del X
    nseq = max(1, tokens // 2048) * 4                    # the 2/N of gptq.py:122-131 as a power of two
    G *= np.exp2(e)[:, None]
    G *= np.exp2(e)[None, :] * ((2.0 / nseq) / 1024.0)   # powers of two: every step exact
    H = G
    H32 = H.astype(np.float32)
    assert np.array_equal(H, H32)
    # bell-shaped weights from integers only (sum of four uniform integers, std 0.018), rounded to fp16 like a module's
    Wi = rng.integers(-1023, 1024, size=(4, m, n)).sum(axis=0)
    W = (Wi.astype(np.float32) * np.float32(2.0 ** -16)).astype(np.float16).astype(np.float32)
    d = np.diag(H).copy()
    perm = np.argsort(d, kind="stable")                  # act_sort = "asc" (gptq.py:281-286)
    W = np.ascontiguousarray(W[:, perm])
    H32 = np.ascontiguousarray(H32[perm][:, perm])
    del H, G
    idx = np.arange(n)
    rowabs = np.array([np.abs(H32[i].astype(np.float64)).sum() for i in range(n)])  # exact: multiples of one power of two
    dg = H32[idx, idx].astype(np.float64)
    offset = np.clip(rowabs - 2.0 * dg, 1e-8, None).astype(np.float32)  # gptq.py:289-291
    damp = np.float32(0.01 * dg.mean())                  # gptq.py:296-298
    A1 = H32.copy()
    A1[idx, idx] += offset
    L = c_oracle.det_cholesky(A1)
    del A1
    Hd = H32
    Hd[idx, idx] += damp
    Lr = c_oracle.det_cholesky(np.ascontiguousarray(Hd[::-1, ::-1]))
    hinv_diag = (np.float32(1.0) / np.diag(Lr)[::-1]).astype(np.float32)
    del Lr
    out = dict(W=W, H=Hd, L=L, hinv_diag=hinv_diag, perm=perm.astype(np.int64), damp=float(damp))
    _base_cache[key] = out
    return out


def make(m: int, n: int, bits: int, seed: int, tokens: int):
    """-> dict(W [m,n] f32 (permuted columns), H = Xxt_damped [n,n] f32, L [n,n] f32 lower, hinv_diag [n] f32,
    T0 [m,V] f32, perm [n] int64, damp float)"""
    out = dict(_base(m, n, seed, tokens))
    hinv_diag = out["hinv_diag"]
    h2 = hinv_diag * hinv_diag                            # ganq.py:427-429: diag(Hinv)^-4 in fp32, by exact elementwise
    weights = np.float32(1.0) / (h2 * h2)                 # operations (a library pow() may round differently per CPU)
    out["T0"] = c_oracle.kmeans_init(out["W"], weights.astype(np.float64), 2 ** bits)
    return out


def hashes(inp) -> dict:
    return {k: sha(inp[k]) for k in ("W", "H", "L", "hinv_diag", "T0")}


def check(inp, fixture) -> None:
    """the rebuilt inputs must be the ones the reference ran on"""
    for k, h in hashes(inp).items():
        want = str(fixture["sha_" + k])
        if h != want:
            raise AssertionError(f"exact_inputs: `{k}` rebuilt on this host differs from the fixture's ({h[:12]} vs {want[:12]}): "
                                 f"the large golden case cannot be compared bit for bit here")


def pack_q_trace(Qs: np.ndarray) -> dict:
    """[K,m,n] uint8 indices -> first iteration + XOR with the previous one (almost all zeros: compresses well)"""
    out = {"Q_first": Qs[0]}
    if Qs.shape[0] > 1:
        out["Q_xor"] = Qs[1:] ^ Qs[:-1]
    return out


def unpack_q_trace(fx) -> np.ndarray:
    Q = [np.asarray(fx["Q_first"])]
    if "Q_xor" in fx:
        for d in np.asarray(fx["Q_xor"]):
            Q.append(Q[-1] ^ d)
    return np.stack(Q)


def row_digest(Q: np.ndarray) -> np.ndarray:
    """one 64-bit digest per row of an index matrix (so a hash-only case can still count the rows that differ)"""
    return np.array([int.from_bytes(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest()[:8], "little")
                     for r in Q], dtype=np.uint64)

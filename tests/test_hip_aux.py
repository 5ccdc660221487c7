"""GPU parity tests for the kernels either side of the loop: Hessian accumulation (a1), codebook
initialisation (a3), index packing and the LUT-dequant linear forward (a9) -- through the C-ABI, against the
CPU oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, rel_fro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ganq_amd import _lib

    assert torch.cuda.is_available()
    _lib.lib()
    return _lib


@pytest.fixture(scope="module")
def oracle():
    from oracle import c_oracle

    return c_oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------------ Hessian
@pytest.mark.parametrize("name", golden_names())
def test_hessian_golden(hip, name):
    g = load_golden(name)
    X = g["X"]  # [nb, bsz, seq, n] fp16
    n = X.shape[-1]
    H = torch.zeros(n, n, device="cuda")
    N = 0
    for b in range(X.shape[0]):
        xb = dev(X[b].reshape(-1, n))
        hip.hessian_accum(H, xb, N, X.shape[1])
        N += X.shape[1]
    Hn = H.cpu().numpy()
    assert np.array_equal(Hn, Hn.T)
    assert rel_fro(Hn, g["H_raw"]) < 1e-6


def test_hessian_batch_beyond_the_offset_window_is_split(hip):
    # the kernel's fast path reaches X through 32-bit buffer offsets; a batch of more than 2 GB is accumulated in pieces
    # (first piece with the batch's decay, the others adding to it): same result as the batch in two separate halves
    n, rows = 2048, 540_000  # 2.2 GB of fp16
    g = torch.Generator(device="cuda").manual_seed(5)
    X = (0.5 * torch.randn(rows, n, device="cuda", generator=g)).half()
    H0 = torch.randn(n, n, device="cuda", generator=g)
    H0 = (H0 + H0.T).contiguous()
    H = H0.clone()
    hip.hessian_accum(H, X, 3, 5)  # H <- H * 3/8 + (2/8) X^T X
    ref = H0.double() * (3.0 / 8.0)
    for a in range(0, rows, 60_000):  # fp64 reference in slices (the fp64 copy of all of X would be 8.8 GB)
        xs = X[a:a + 60_000].double()
        ref += (2.0 / 8.0) * (xs.T @ xs)
    err = float((H.double() - ref).norm() / ref.norm())
    assert err < 3e-5, err  # fp32 accumulators over 540 000 tokens (6e-6 measured); a lost or doubled piece would be 0.4
    assert torch.equal(H, H.T)
    # the same batch handed over in two calls (each inside the window): fp32 sums in the same token order
    H2 = H0.clone()
    half = rows // 2 // 32 * 32
    hip.hessian_accum(H2, X[:half], 3, 5)                 # H * 3/8 + (2/8) X1^T X1
    H3 = torch.zeros_like(H2)
    hip.hessian_accum(H3, X[half:], 0, 8)                 # (2/8) X2^T X2
    err2 = float((H.double() - (H2.double() + H3.double())).norm() / ref.norm())
    assert err2 < 2e-5, err2  # different grouping of the fp32 sums: 4e-6 measured


@pytest.mark.parametrize("rows,n,dtype", [(100, 72, torch.float16), (333, 264, torch.bfloat16), (2048, 512, torch.float16)])
def test_hessian_vs_oracle_shapes(hip, oracle, rows, n, dtype):
    g = torch.Generator().manual_seed(rows + n)
    X1 = torch.randn(rows, n, generator=g).to(dtype)
    X2 = torch.randn(rows // 2, n, generator=g).to(dtype)
    H = torch.full((n, n), 7.0, device="cuda")  # stale contents must be ignored on the first batch
    hip.hessian_accum(H, X1.cuda(), 0, 3)
    hip.hessian_accum(H, X2.cuda(), 3, 2)
    ref = (2.0 / 5.0) * (X1.double().T @ X1.double() + X2.double().T @ X2.double())
    assert rel_fro(H.cpu().numpy(), ref.numpy()) < 1e-6
    if dtype == torch.float16:
        Ho = np.zeros((n, n), dtype=np.float32)
        oracle.hessian_accum(Ho, X1.numpy(), 0, 3)
        oracle.hessian_accum(Ho, X2.numpy(), 3, 2)
        assert rel_fro(H.cpu().numpy(), Ho) < 1e-6


@pytest.mark.parametrize("rows,n,dtype", [(4096, 768, torch.float16), (5000, 1280, torch.bfloat16), (16384, 2048, torch.float16),
                                          (4101, 3072, torch.float16), (4096, 5120, torch.float16)])
def test_hessian_token_split_launches(hip, rows, n, dtype, lib_options):
    """Layers with fewer 128 x 128 tiles than workgroup slots cut the tokens of some tiles into parts (hessian_sk_kernel) and add
    the parts in token order (hessian_fix_kernel).  Same products, another grouping of the fp32 sums: equal to the whole-tile
    kernel to fp32 rounding, exactly symmetric, run-to-run identical, the running average's decay applied once; ragged token
    counts (last part shorter, last slab partial) and a layer with more tiles than slots (n = 5120: only the tiles behind the
    last full round are cut) included."""
    g = torch.Generator().manual_seed(rows + n)
    X1 = (torch.randn(rows, n, generator=g) * 0.5).to(dtype).cuda()
    X2 = (torch.randn(rows, n, generator=g) * 0.5).to(dtype).cuda()
    outs = {}
    # (wide, split): 128 x 128 whole tiles (the reference point); 128 x 128 cut; 256 x 128 tiles (where in_features is a multiple
    # of 256: every shape here) whole / cut / cut into at most 3 parts
    configs = [(0, 0), (0, 1), (0, 3), (2, 0), (2, 1), (2, 3)]  # (GANQ_HESS_WIDE = 2: wide tiles whatever in_features)
    for cfg in configs:
        lib_options(GANQ_HESS_WIDE=cfg[0], GANQ_HESS_SPLIT=cfg[1])
        H = torch.full((n, n), 7.0, device="cuda")  # stale contents must be ignored on the first batch
        hip.hessian_accum(H, X1, 0, 2)
        hip.hessian_accum(H, X2, 2, 3)              # H * 2/5 + (2/5) X2^T X2
        outs[cfg] = H
    ref = (2.0 / 5.0) * (X1.double().T @ X1.double() + X2.double().T @ X2.double())
    for cfg in configs[1:]:
        assert torch.equal(outs[cfg], outs[cfg].T), cfg
        assert float((outs[cfg].double() - ref).norm() / ref.norm()) < 1e-6, cfg
        assert float((outs[cfg] - outs[(0, 0)]).abs().max() / outs[(0, 0)].abs().max()) < 1e-5, cfg
    # whole tiles of either shape add the same products in the same order: the same bits
    assert torch.equal(outs[(2, 0)], outs[(0, 0)])
    for cfg in ((0, 1), (2, 1)):  # deterministic: no atomics, fixed order
        lib_options(GANQ_HESS_WIDE=cfg[0], GANQ_HESS_SPLIT=cfg[1])
        H = torch.full((n, n), 7.0, device="cuda")
        hip.hessian_accum(H, X1, 0, 2)
        hip.hessian_accum(H, X2, 2, 3)
        assert torch.equal(H, outs[cfg]), cfg


@pytest.mark.parametrize("rows,n,dtype", [(4096, 1024, torch.float16), (8192, 2048, torch.bfloat16), (16384, 4096, torch.float16),
                                          (4096, 1288, torch.float16), (6144, 3072, torch.bfloat16), (4128, 2048, torch.float16),
                                          (2048, 4096, torch.float16), (40, 1024, torch.float16)])
def test_hessian_transposed_staging_stream_k(hip, rows, n, dtype, lib_options):
    """hessian_w4.hip: the batches of a group staged TRANSPOSED (hessian_stage_t, one call per batch) and multiplied as Xt Xt^T with
    256 x 256 tiles, the (tile, token slice) pairs cut into equal shares, one per CU; partial tiles meet through a ticket and are
    summed in range order.  Against the fp64 product and the row-major kernels (same products, another grouping of the fp32 sums);
    exactly symmetric; the same bits from run to run whichever workgroup finishes a tile; the running average's decay applied
    once; in_features that are not a multiple of the tile (1288), token counts that are not a multiple of the slice (4128 -> the
    host zero-fills, 40), a group that fills only part of the staging buffer."""
    g = torch.Generator().manual_seed(rows + n)
    X1 = (torch.randn(rows, n, generator=g) * 0.5).to(dtype).cuda()
    X2 = (torch.randn(rows, n, generator=g) * 0.5).to(dtype).cuda()
    cap = -(-rows // 32) * 32 + 64  # (the buffer is larger than the group)
    lib_options(GANQ_HESS_W4=2)     # (the product takes this path from 3072 in_features on)
    assert hip.hessian_t_supported(n, cap)

    def run():
        H = torch.full((n, n), 7.0, device="cuda")  # stale contents must be ignored on the first batch
        Xt = torch.full((n, cap), 3.0, dtype=dtype, device="cuda")  # stale tokens behind the group must not count
        pad = -(-rows // 32) * 32
        for X, before, batch in ((X1, 0, 2), (X2, 2, 3)):
            step = max(8, rows // 3 // 8 * 8)  # three or four batches per group
            for t0 in range(0, rows, step):
                hip.hessian_stage_t(Xt, X[t0:t0 + step], t0)
            Xt[:, rows:pad].zero_()
            hip.hessian_accum_t(H, Xt, pad, before, batch)
        return H

    H = run()
    Hr = torch.full((n, n), 7.0, device="cuda")
    hip.hessian_accum(Hr, X1, 0, 2)
    hip.hessian_accum(Hr, X2, 2, 3)              # H * 2/5 + (2/5) X2^T X2
    ref = (2.0 / 5.0) * (X1.double().T @ X1.double() + X2.double().T @ X2.double())
    assert torch.equal(H, H.T)
    assert float((H.double() - ref).norm() / ref.norm()) < 1e-6
    assert float((H - Hr).abs().max() / Hr.abs().max()) < 1e-5
    for _ in range(3):  # deterministic
        assert torch.equal(run(), H)


def test_hessian_transposed_staging_in_the_quantizer(hip, lib_options):
    """GPTQ.add_batch stages transposed where the layer is served (n >= 1024) and row-major elsewhere: the same Hessian as batch-by-batch
    accumulation, ragged batches and a dtype change inside the calibration included"""
    from ganq_amd.quantization import GANQ, QuantizeConfig
    from ganq_amd.looper.named_module import NamedModule

    n = 1024
    lib_options(GANQ_HESS_W4=2)
    lin = torch.nn.Linear(n, 32, bias=False).half().cuda()
    g = torch.Generator().manual_seed(3)
    xs = [(torch.randn(1, t, n, generator=g) * 0.5).half().cuda() for t in (256, 256, 200, 256, 4, 256, 36)]
    Hs = {}
    for stage in (0, 1024):
        q = GANQ(NamedModule(lin, "fc", "layers.0.fc", 0), QuantizeConfig(bits=4, ganq_hessian_stage_tokens=stage))
        q.quantizer.configure(perchannel=True)
        for x in xs:
            q.add_batch(x, None)
        Hs[stage] = q.hessian.clone()
        assert q.nsamples == len(xs)
        assert stage == 0 or q._stage_t
        q.free() if hasattr(q, "free") else None
    assert torch.equal(Hs[1024], Hs[1024].T)
    assert float((Hs[1024] - Hs[0]).abs().max() / Hs[0].abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------ Cholesky (prologue)
@pytest.mark.parametrize("n", [1, 5, 127, 128, 129, 300, 1000, 2048])
def test_cholesky_vs_fp64(hip, n):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((2 * n + 3, n)) * (0.1 + rng.random(n))
    H = (X.T @ X / X.shape[0] + 0.01 * np.eye(n)).astype(np.float32)
    L = hip.cholesky(dev(H)).cpu().numpy()
    assert np.all(np.triu(L, 1) == 0)
    ref = np.linalg.cholesky(H.astype(np.float64))
    assert rel_fro(L, ref) < 2e-6
    assert rel_fro(L.astype(np.float64) @ L.astype(np.float64).T, H) < 2e-6
    Lt = torch.linalg.cholesky(dev(H)).cpu().numpy()  # rocSOLVER, fp32: same size of error against fp64
    assert rel_fro(L, ref) < 4 * max(rel_fro(Lt, ref), 1e-7)


@pytest.mark.parametrize("n", [4096, 4230])
def test_cholesky_lookahead_path(hip, n, lib_options):
    """from 4096 columns on the factorisation runs its trailing updates on a second stream and the diagonal kernel applies the previous
    step's update to its own block: against fp64, against the single-stream order (GANQ_CHOL_LOOKAHEAD=0: the same additions per
    element in another order of the two K = 128 products -- fp32 rounding apart), twice in a row and two at once on two streams
    (the prologue does that), a ragged last block included"""
    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn(2 * n, n, device="cuda", generator=g) * (0.1 + torch.rand(n, device="cuda", generator=g))
    H = (X.T @ X) / X.shape[0]
    H += 0.01 * H.diag().mean() * torch.eye(n, device="cuda")
    L = hip.cholesky(H)
    assert torch.equal(hip.cholesky(H), L)
    Hd = H.double()
    assert float((L.double() @ L.double().T - Hd).norm() / Hd.norm()) < 2e-6
    ref = torch.linalg.cholesky(Hd)
    assert float((L.double() - ref).norm() / ref.norm()) < 2e-6
    assert torch.all(torch.triu(L, 1) == 0)
    lib_options(GANQ_CHOL_LOOKAHEAD=0)
    L0 = hip.cholesky(H)
    lib_options(GANQ_CHOL_LOOKAHEAD=1)
    assert float((L - L0).abs().max() / L0.abs().max()) < 1e-5
    H2 = H + 0.5 * torch.eye(n, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        La = hip.cholesky(H)
    with torch.cuda.stream(s2):
        Lb = hip.cholesky(H2)
    torch.cuda.synchronize()
    assert torch.equal(La, L)
    assert float((Lb.double() @ Lb.double().T - H2.double()).norm() / H2.double().norm()) < 2e-6


def test_cholesky_not_positive_definite_raises(hip):
    H = np.eye(200, dtype=np.float32)
    H[150, 150] = -1.0
    with pytest.raises(torch.linalg.LinAlgError):
        hip.cholesky(dev(H))
    L, info = hip.cholesky(dev(H), check=False)
    assert int(info) == 151


# ------------------------------------------------------------------------------------------ k-means
@pytest.mark.parametrize("m,n,V,seed", [(8, 64, 4, 1), (32, 300, 16, 2), (16, 1024, 8, 3), (5, 4096, 16, 4), (3, 17, 16, 5),
                                        (700, 256, 16, 6), (2, 4700, 16, 7), (2, 4800, 16, 8), (3, 1, 4, 9), (2, 2, 4, 10),
                                        # every CU busy (one / two workgroups per CU): the ordering bugs of a workgroup-wide sort or
                                        # level scheme are timing-dependent and do not show on a handful of rows
                                        (384, 4096, 16, 11), (640, 2048, 16, 12)])
def test_kmeans_vs_oracle(hip, oracle, m, n, V, seed):
    rng = np.random.default_rng(seed)
    W = (0.02 * rng.standard_normal((m, n))).astype(np.float16).astype(np.float32)
    wts = rng.uniform(0.2, 3.0, n) ** -4
    T0 = hip.kmeans_init(dev(W), torch.from_numpy(wts), V).cpu().numpy()
    ref = oracle.kmeans_init(W, wts, V)
    assert rel_fro(T0, ref) < 1e-6, np.abs(T0 - ref).max()
    assert np.all(np.diff(T0, axis=1) >= 0)


@pytest.mark.parametrize("wcap", [64, 200, 1024])
def test_kmeans_windowed_kernel_small_windows(hip, oracle, wcap, lib_options):
    # the large-n kernel (prefix sums in L2, LDS window per segment of nodes) forced onto small rows with a tiny window:
    # exercises multi-segment levels and nodes wider than the window
    lib_options(GANQ_KMEANS_WCAP=wcap)
    rng = np.random.default_rng(wcap)
    for m, n, V in [(6, 777, 16), (3, 2048, 8), (300, 130, 4)]:
        W = (0.02 * rng.standard_normal((m, n))).astype(np.float16).astype(np.float32)
        wts = rng.uniform(0.2, 3.0, n) ** -4
        T0 = hip.kmeans_init(dev(W), torch.from_numpy(wts), V).cpu().numpy()
        ref = oracle.kmeans_init(W, wts, V)
        assert rel_fro(T0, ref) < 1e-6, (m, n, V, np.abs(T0 - ref).max())


def test_kmeans_golden_T0(hip):
    for name in golden_names():
        g = load_golden(name)
        V = 2 ** int(g["bits"])
        wts = g["Hinv_diag"].astype(np.float32) ** -4  # ganq.py:427-429 (float32 power, then numpy)
        T0 = hip.kmeans_init(dev(g["W_perm"]), torch.from_numpy(wts.astype(np.float64)), V).cpu().numpy()
        assert rel_fro(T0, g["T"][0]) < 1e-6, name


# ------------------------------------------------------------------------------------------ packing / LUT linear
@pytest.mark.parametrize("bits", [2, 3, 4])
def test_pack_roundtrip_and_layout(hip, bits):
    rng = np.random.default_rng(bits)
    m, n = 48, 160
    Q = rng.integers(0, 2 ** bits, size=(m, n), dtype=np.uint8)
    qw = hip.pack_indices(dev(Q), bits)
    assert qw.shape == (n * bits // 32, m) and qw.dtype == torch.int32
    assert np.array_equal(hip.unpack_indices(qw, n, bits).cpu().numpy(), Q)
    # the GPTQ int32 layout (qlinear/__init__.py:508-538): a little-endian bit stream along in_features
    ref = np.zeros((n * bits // 32, m), dtype=np.uint64)
    for e in range(n):
        pos = e * bits
        w, sh = pos // 32, pos % 32
        v = Q[:, e].astype(np.uint64) << np.uint64(sh)
        ref[w] |= v & np.uint64(0xFFFFFFFF)
        if sh + bits > 32:
            ref[w + 1] |= v >> np.uint64(32)
    assert np.array_equal(qw.cpu().numpy().view(np.uint32), ref.astype(np.uint32))


@pytest.mark.parametrize("bits,M,dtype,m,n", [(4, 1, torch.float16, 200, 512), (4, 5, torch.float16, 200, 512),
                                              (3, 2, torch.bfloat16, 200, 512), (2, 8, torch.float16, 200, 512),
                                              (4, 16, torch.bfloat16, 200, 512), (3, 17, torch.float16, 77, 2080),
                                              (4, 33, torch.bfloat16, 45, 4128), (2, 64, torch.float16, 130, 1056),
                                              (3, 1, torch.float16, 1, 32), (4, 64, torch.float16, 4096, 11008),
                                              # decode kernel (out_features >= 128; the cases above with 200 / 130 features too): one and two row tiles, 16 and 32 features
                                              # per workgroup (from 8192), ragged and odd out_features, every bit width
                                              (4, 1, torch.float16, 1024, 512), (4, 20, torch.float16, 1024, 512),
                                              (3, 32, torch.bfloat16, 2048, 1024), (2, 24, torch.float16, 1500, 256),
                                              (4, 1, torch.bfloat16, 8192, 256), (4, 16, torch.float16, 8200, 512),
                                              (3, 5, torch.float16, 8192, 320), (4, 3, torch.float16, 1025, 256),
                                              (2, 1, torch.float16, 9001, 64), (4, 7, torch.float16, 4096, 14336),
                                              # decode kernel, fewer feature blocks than CUs: in_features split across
                                              # workgroups too (2 and 4 ways), partial tiles + ticket
                                              (4, 1, torch.float16, 2048, 8192), (4, 16, torch.bfloat16, 1024, 4096),
                                              (3, 20, torch.float16, 1500, 2048), (4, 3, torch.float16, 1025, 8192),
                                              (2, 1, torch.float16, 2048, 4096)])
def test_lut_linear_vs_dense(hip, oracle, bits, M, dtype, m, n):
    rng = np.random.default_rng(bits * 100 + M)
    V = 2 ** bits
    Q = rng.integers(0, V, size=(m, n), dtype=np.uint8)
    lut = torch.from_numpy((0.02 * rng.standard_normal((m, V))).astype(np.float32)).to(dtype)
    x = torch.from_numpy(rng.standard_normal((M, n)).astype(np.float32)).to(dtype)
    bias = torch.from_numpy((0.1 * rng.standard_normal(m)).astype(np.float32)).to(dtype)
    qw = hip.pack_indices(dev(Q), bits)
    y = hip.lut_linear(x.cuda(), qw, lut.cuda(), bias.cuda(), bits).cpu()
    Wq = torch.gather(lut.float(), 1, torch.from_numpy(Q.astype(np.int64)))
    ref = x.double() @ Wq.double().T + bias.double()
    # one rounding to the activation dtype on top of an fp32 accumulation
    eps = 2 ** -10 if dtype == torch.float16 else 2 ** -7
    assert torch.allclose(y.double(), ref, rtol=eps, atol=eps * float(ref.abs().max()) * 0.05 + 1e-6)
    assert torch.equal(hip.lut_dequant(qw, lut.cuda(), n, bits).cpu(), Wq.to(dtype))
    if dtype == torch.float16:
        yo = oracle.lut_linear(x.numpy(), Q, lut.numpy(), bias.numpy())
        assert np.allclose(y.float().numpy(), yo, rtol=eps, atol=eps * 0.05 * np.abs(yo).max() + 1e-6)


def test_lut_linear_repeat_is_bitwise_stable(hip):
    # split-K through ticket counters in a persistent workspace: every call must leave them clean, and the
    # reduction order is fixed -> many calls, interleaved shapes, identical bits
    rng = np.random.default_rng(7)
    outs = {}
    cases = [(4, 1, 4096, 4096), (3, 16, 1000, 2048), (4, 40, 300, 8192), (2, 3, 129, 64), (4, 1, 2048, 8192), (4, 9, 1024, 4096)]
    data = {}
    for bits, M, m, n in cases:
        Q = rng.integers(0, 2 ** bits, size=(m, n), dtype=np.uint8)
        data[(bits, M, m, n)] = (hip.pack_indices(dev(Q), bits),
                                 torch.from_numpy((0.02 * rng.standard_normal((m, 2 ** bits))).astype(np.float32)).half().cuda(),
                                 torch.from_numpy(rng.standard_normal((M, n)).astype(np.float32)).half().cuda())
    for rep in range(25):
        for key in cases:
            qw, lut, x = data[key]
            y = hip.lut_linear(x, qw, lut, None, key[0])
            if rep == 0:
                outs[key] = y.clone()
            else:
                assert torch.equal(y, outs[key]), (key, rep)


@pytest.mark.parametrize("with_outliers", [False, True])
def test_lut_linear_one_workspace_serves_alternating_shapes(hip, with_outliers):
    # M = 48 takes the split-K-through-memory variant (KS = 2).  One persistent workspace serves every layer of a model:
    # a layer with <= 8192 output features followed by one with more (Llama gate / up_proj, 14336) must not find the
    # smaller layer's partial tiles where its own ticket counters live (the counters have a fixed region).
    rng = np.random.default_rng(48)
    M, bits = 48, 4
    layers = []
    for m, n in [(4096, 1024), (14336, 1024), (4096, 1024), (14336, 1024)]:
        Q = rng.integers(0, 16, size=(m, n), dtype=np.uint8)
        lut = torch.from_numpy((0.02 * rng.standard_normal((m, 16))).astype(np.float32)).half().cuda()
        x = torch.from_numpy(rng.standard_normal((M, n)).astype(np.float32)).half().cuda()
        qw = hip.pack_indices(dev(Q), bits)
        Wq = torch.gather(lut.float(), 1, torch.from_numpy(Q.astype(np.int64)).cuda())
        ref = x.double() @ Wq.double().T
        sparse = None
        if with_outliers:
            nnz_rows = torch.arange(0, m, 7, device="cuda")
            rowptr = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
            cnt = torch.zeros(m, dtype=torch.int32, device="cuda")
            cnt[nnz_rows] = 1
            rowptr[1:] = torch.cumsum(cnt, 0)
            cols = ((nnz_rows * 13) % n).to(torch.int32)
            vals = torch.full((nnz_rows.numel(),), 0.5, dtype=torch.float16, device="cuda")
            sparse = (rowptr, cols, vals)
            ref[:, nnz_rows] += 0.5 * x.double()[:, cols.long()]
        layers.append((x, qw, lut, sparse, ref))
    for rep in range(3):
        for x, qw, lut, sparse, ref in layers:
            if sparse is None:
                y = hip.lut_linear(x, qw, lut, None, bits)
            else:
                y = hip.lut_linear_outliers(x, qw, lut, None, bits, *sparse)
            assert torch.isfinite(y).all()
            assert torch.allclose(y.double(), ref, rtol=2 ** -9, atol=2 ** -9 * float(ref.abs().max()) * 0.05 + 1e-6), (rep, tuple(lut.shape))


@pytest.mark.parametrize("ks", [2, 3, 5])
def test_lut_decode_forced_split_matches_unsplit(hip, lib_options, ks):
    # the cross-workgroup split of the decode kernel with ragged shares (128 column groups in 3 / 5 parts) against the
    # unsplit launch: same fp32 partial sums in a different association -> equal to rounding, and stable over repeats
    rng = np.random.default_rng(ks)
    m, n, bits = 4096, 4096, 4
    Q = rng.integers(0, 16, size=(m, n), dtype=np.uint8)
    lut = torch.from_numpy((0.02 * rng.standard_normal((m, 16))).astype(np.float32)).half().cuda()
    qw = hip.pack_indices(dev(Q), bits)
    Wq = torch.gather(lut.float(), 1, torch.from_numpy(Q.astype(np.int64)).cuda())
    for M in (1, 16, 24):
        x = torch.from_numpy(rng.standard_normal((M, n)).astype(np.float32)).half().cuda()
        ref = x.double() @ Wq.double().T
        lib_options(GANQ_LUT_KS=ks)
        y = hip.lut_linear(x, qw, lut, None, bits)
        y2 = hip.lut_linear(x, qw, lut, None, bits)
        assert torch.equal(y, y2)
        assert torch.allclose(y.double(), ref, rtol=2 ** -10, atol=2 ** -10 * float(ref.abs().max()) * 0.05 + 1e-6), (ks, M)


def test_lut_linear_golden_forward(hip):
    # G7: FakeQuantLinear.forward (fake.py:88-89) on the reference's own quantized weight
    for name in golden_names():
        g = load_golden(name)
        if int(g["n"]) % 32 or not (bool(g["desc_act"]) or str(g["act_sort"]) == "none"):
            continue
        K, bits = int(g["K"]), int(g["bits"])
        best_k = int(np.argmin(g["dists"]))
        Q = g["Q"][K - 1][:, np.argsort(g["perm"])]  # back to the original column order (gptq.py:341-343)
        lut = torch.from_numpy(g["T"][best_k + 1]).half()
        qw = hip.pack_indices(dev(Q), bits)
        y = hip.lut_linear(dev(g["x_fwd"]), qw, lut.cuda(), dev(g["bias"]), bits).cpu().float().numpy()
        ref = g["y_fwd"].astype(np.float32)
        assert np.allclose(y, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max()), name


# ------------------------------------------------------------------------------------------ prologue passes
@pytest.mark.parametrize("n,m,act_sort,dead_mode", [(96, 40, "asc", "mean"), (300, 64, "desc", "zero"), (1030, 33, "none", "mean"),
                                                     (2048, 128, "asc", "mean")])
def test_prologue_passes_match_the_reference_op_sequence(hip, n, m, act_sort, dead_mode):
    """csrc/prologue.hip against the torch ops of gptq.py:267-300 they replace: dead columns, permutation gathers, ganq-style
    offset, damping, index reversal -- everything but the row sums is data movement and must be bit-identical"""
    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn(2 * n, n, device="cuda", generator=g) * (0.1 + torch.rand(n, device="cuda", generator=g))
    X[:, 5] = 0
    X[:, n - 3] = 0  # two dead columns
    H = (X.T @ X) / n
    W = torch.randn(m, n, device="cuda", generator=g)
    diag, rowabs = hip.prologue_rowstats(H)
    assert torch.equal(diag, torch.diag(H))
    assert torch.allclose(rowabs, torch.sum(torch.abs(H), dim=1), rtol=1e-6)
    dead = diag == 0
    assert int(dead.sum()) == 2
    # the reference's sequence
    Hr, Wr = H.clone(), W.clone()
    Hr[dead, dead] = 1
    Wr[:, dead] = 0 if dead_mode == "zero" else torch.mean(Wr[:, ~dead], dim=1, keepdim=True)
    perm = None
    if act_sort != "none":
        perm = torch.argsort(torch.diag(Hr), descending=act_sort == "desc")
        Wr = Wr[:, perm]
        Hr = Hr[perm][:, perm]
    damp = 0.01 * torch.mean(torch.diag(Hr))
    offset = (torch.sum(torch.abs(Hr), dim=1) - 2 * torch.diag(Hr)).clamp(min=1e-8)
    # the passes
    diag_fixed = torch.where(dead, torch.ones_like(diag), diag)
    Wp = hip.prologue_weights(W, perm, dead, mean_fill=dead_mode == "mean")
    if dead_mode == "zero":
        assert torch.equal(Wp, Wr)
    else:
        live = ~(dead if perm is None else dead[perm])
        assert torch.equal(Wp[:, live], Wr[:, live]) and torch.allclose(Wp, Wr, rtol=1e-6, atol=1e-7)
    add = damp.expand(n).contiguous()
    Xd, Hf, A1 = hip.prologue_gather(H, perm, diag_fixed, [(add, False), (add, True), (offset, False)])
    eye = torch.eye(n, device="cuda", dtype=torch.bool)
    want = Hr.clone()
    want[eye] += damp
    assert torch.equal(Xd, want)
    assert torch.equal(Hf, torch.flip(want, dims=(0, 1)))
    assert torch.equal(A1, Hr + torch.diag(offset))
    # in-place factorisation == the copying one
    L1 = hip.cholesky(A1)
    L2 = hip.cholesky_inplace(A1.clone())
    assert torch.equal(L1, L2)

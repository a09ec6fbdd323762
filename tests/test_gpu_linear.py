"""GPU parity of the q/k/v producers (second half) through the C ABI:
nvk_batch_rms_norm_device, nvk_batch_matmul_device, nvk_qkv_project_device against the checker
(oracle_batch_rms_norm / oracle_batch_matmul, pinned bit-exact to Ops::rms_norm / Ops::matmul and the BatchOps
forms) and the committed fixture F7.

Tolerance: the kernels add the same fp32 products in another order than the reference's serial loops, so the
bar is the path's 1e-4 absolute + 1e-5 relative (BASELINE.json north_star), written in `close`."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from golden_defs import QKV_CASES, qkv_case_inputs
from nano_vllm_amd.workloads import uniform

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-4, 1e-5


def close(got, want):
    return bool(np.all(np.abs(got - want) <= ATOL + RTOL * np.abs(want)))


@pytest.fixture(scope="module")
def nv():
    import nano_vllm_amd
    return nano_vllm_amd


@pytest.fixture(scope="module")
def DA():
    from hipmem import DeviceArray
    return DeviceArray


def project(nv, DA, x, g, wq, wk, wv):
    B, D = x.shape
    QD, KD = wq.shape[0], wk.shape[0]
    dx, dg, dq_w, dk_w, dv_w = DA(x), (DA(g) if g is not None else None), DA(wq), DA(wk), DA(wv)
    xb = DA(np.full((B, D), np.nan, np.float32))
    q, k, v = (DA(np.full((B, n), np.nan, np.float32)) for n in (QD, KD, KD))
    nv.qkv_project_device(q, k, v, dx, xb if g is not None else None, dg, dq_w, dk_w, dv_w, B, D, QD, KD)
    return xb.numpy(), q.numpy(), k.numpy(), v.numpy()


@pytest.mark.parametrize("case", QKV_CASES, ids=[c["name"] for c in QKV_CASES])
def test_golden_qkv_cases(case, nv, DA, golden_dir):
    gold = np.load(os.path.join(golden_dir, "qkv_cases.npz"))
    x, g, wq, wk, wv = qkv_case_inputs(case)
    xb, q, k, v = project(nv, DA, x, g, wq, wk, wv)
    n = case["name"]
    assert close(xb, gold[n + "/xb"]), float(np.abs(xb - gold[n + "/xb"]).max())
    for got, key in ((q, "q"), (k, "k"), (v, "v")):
        assert close(got, gold[n + "/" + key]), (key, float(np.abs(got - gold[n + "/" + key]).max()))


def test_rms_norm_rows_and_in_place(nv, DA):
    for seed, (B, D) in enumerate([(1, 4096), (7, 288), (300, 1024), (3, 6), (2, 4100), (2, 9000), (3, 8192)]):
        x = (uniform(800 + seed, 0, B * D) * np.float32(3)).reshape(B, D)
        g = uniform(810 + seed, 0, D) + np.float32(1.25)
        want = ol.batch_rms_norm(ol.ORACLE.oracle_batch_rms_norm, x, g)
        dx, dg, out = DA(x), DA(g), DA(np.zeros_like(x))
        nv.batch_rms_norm_device(out, dx, dg, B, D)
        assert close(out.numpy(), want)
        # relative error of a normalised row stays at the rounding level, far inside the bar
        assert float(np.abs(out.numpy() - want).max()) <= 4e-6 * float(np.abs(want).max())
        nv.batch_rms_norm_device(dx, dx, dg, B, D)  # out may alias in (model.hpp:243 normalises x in place)
        assert np.array_equal(dx.numpy(), out.numpy())


def test_residual_add_fused_into_the_norm(nv, DA):
    """nvk_batch_add_rms_norm_device: x += addend (bit-exact: one fp32 add per element, model.hpp:227-228), then the norm
    of the sum -- the same bits as the add followed by nvk_batch_rms_norm_device."""
    for seed, (B, D) in enumerate([(1, 288), (5, 4096), (3, 9000), (2, 6)]):
        x = uniform(1200 + seed, 0, B * D).reshape(B, D)
        y = uniform(1210 + seed, 0, B * D).reshape(B, D) * np.float32(0.3)
        g = uniform(1220 + seed, 0, D) + np.float32(1.5)
        dx, out = DA(x), DA(np.zeros_like(x))
        nv.batch_add_rms_norm_device(out, dx, DA(y), DA(g), B, D)
        assert np.array_equal(dx.numpy(), x + y)
        want = ol.batch_rms_norm(ol.ORACLE.oracle_batch_rms_norm, x + y, g)
        assert close(out.numpy(), want)
        ref = DA(np.zeros_like(x))
        nv.batch_rms_norm_device(ref, DA(x + y), DA(g), B, D)
        assert np.array_equal(out.numpy(), ref.numpy())
    with pytest.raises(nv.NvkError):
        nv.batch_add_rms_norm_device(dx, dx, dx, dx, 1, 4)  # out must not alias x


@pytest.mark.parametrize("B,D,N", [(1, 4096, 4096), (4, 288, 288), (32, 1024, 100), (33, 1024, 100), (17, 100, 33),
                                    (128, 512, 384), (129, 260, 130), (300, 64, 1), (2, 4, 3),
                                    # 17-32 rows over >= 256 column tiles: two 16-row tiles per weight tile (<16, 2, 32>)
                                    (17, 512, 4100), (24, 1028, 4096), (32, 260, 5000), (31, 4096, 6144),
                                    # 3-16 rows: output columns dealt evenly over the CUs, 8 .. 48 to a workgroup
                                    # (nvk_gemm_skinny16_kernel<1 | 2 | 3>), ragged last workgroups and k tails
                                    (16, 4096, 6144), (3, 2048, 2048), (9, 260, 11264), (16, 64, 12), (5, 132, 4104),
                                    (13, 1024, 20), (16, 2048, 32000), (20, 256, 9000)])
def test_matmul_shapes_against_the_checker(B, D, N, nv, DA):
    """Both kernels (<= 32 rows: weights streamed once; more: 128x128 MFMA tiles), ragged edges in every dimension."""
    x = uniform(900 + B, 0, B * D).reshape(B, D)
    w = (uniform(901 + N, 0, N * D) * np.float32(1.0 / np.sqrt(D))).reshape(N, D)
    want = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x, w)
    out = DA(np.full((B, N), np.nan, np.float32))
    nv.batch_matmul_device(out, DA(x), DA(w), B, D, N)
    got = out.numpy()
    assert close(got, want), float(np.abs(got - want).max())


@pytest.mark.parametrize("B,D,N", [(128, 5632, 2048), (100, 2052, 200), (256, 4096, 1000), (97, 1024, 65),
                                    (200, 8192, 64), (130, 516, 2048),
                                    # up to 16 rows over at most 128 column tiles (nvk_gemm_skinny_kernel<16, 1, 64, .., SPLIT>)
                                    (16, 5632, 2048), (3, 2048, 2048), (9, 4100, 200), (16, 8192, 20), (5, 2052, 1028),
                                    # 33+ rows, at least 16 k-steps to a CU and tiles that would not fill the chip evenly: runs
                                    # of k-steps across tile boundaries (nvk_gemm_nt_kernel<64, 64, 64, kNtStream>)
                                    (64, 2048, 11264), (40, 4100, 6000), (130, 2052, 6000),
                                    # 17-32 rows of a large product: the same over 32 x 128 tiles (<32, 128, 32, kNtStream, 1>)
                                    (24, 2052, 11000), (32, 4096, 6144), (400, 2052, 5000)])
def test_split_k_tiles_against_the_checker_with_changing_inputs(B, D, N, nv, DA):
    """More rows than the weight-streaming kernel takes and too few 64 x 64 tiles for the chip -- or up to 16 rows and
    too few 16-column tiles: the contraction is cut into slices (nvk_gemm_nt_kernel<64, 64, 64, kNtSplit>,
    nvk_gemm_skinny_kernel<16, 1, 64, plain, false, true>), a workgroup per (tile, slice) -- or, stream-K, per run of
    k-steps that crosses tile boundaries -- partial tiles handed over
    through an arrival counter and added in slice order by whoever comes last.  A wrong hand-off shows as stale or
    missing partials, so: three calls in a row with different activations, each against the checker; the same call
    twice gives the same bits (the order of the sums does not depend on who arrives last); and a row's result does
    not depend on its place in the batch.  k tails that are not multiples of 64, ragged rows and columns."""
    w = (uniform(1701 + N, 0, N * D) * np.float32(1.0 / np.sqrt(D))).reshape(N, D)
    dw = DA(w)
    out = DA(np.full((B, N), np.nan, np.float32))
    first = None
    for rnd in range(3):
        x = (uniform(1700 + 7 * rnd + B, 0, B * D) * np.float32(1 + rnd)).reshape(B, D)
        want = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x, w)
        nv.batch_matmul_device(out, DA(x), dw, B, D, N)
        got = out.numpy()
        assert close(got, want), (rnd, float(np.abs(got - want).max()))
        if rnd == 0:
            first = (x, got)
    x, got = first
    nv.batch_matmul_device(out, DA(x), dw, B, D, N)
    assert np.array_equal(out.numpy(), got)
    perm = np.random.default_rng(B).permutation(B)
    nv.batch_matmul_device(out, DA(x[perm]), dw, B, D, N)
    assert np.array_equal(out.numpy(), got[perm])
    nv.batch_matmul_device(out, DA(x * np.float32(2)), dw, B, D, N)
    assert np.array_equal(out.numpy(), got * np.float32(2))


def test_llama_size_projection_sampled_rows_and_exact_properties(nv, DA):
    """dim 4096 -> 32 q-heads + 8 kv-heads x 128 at 512 rows: sampled rows against the checker, and two properties
    that hold bit for bit at any size: scaling the input by 2 doubles every output (power-of-two scaling commutes
    with fp32 rounding), and an output row does not depend on which rows share its batch (a permutation of the
    rows permutes the outputs) -- in the tiled kernel and in the skinny one."""
    B, D, QD, KD = 512, 4096, 4096, 1024
    x = uniform(1000, 0, B * D).reshape(B, D)
    sc = np.float32(1.0 / np.sqrt(D))
    wq = (uniform(1001, 0, QD * D) * sc).reshape(QD, D)
    wk = (uniform(1002, 0, KD * D) * sc).reshape(KD, D)
    wv = (uniform(1003, 0, KD * D) * sc).reshape(KD, D)
    _, q, k, v = project(nv, DA, x, None, wq, wk, wv)
    rows = [0, 1, 31, 127, 128, 300, 511]
    for got, w in ((q, wq), (k, wk), (v, wv)):
        want = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x[rows], w)
        assert close(got[rows], want), float(np.abs(got[rows] - want).max())
    _, q2, k2, v2 = project(nv, DA, x * np.float32(2), None, wq, wk, wv)
    assert np.array_equal(q2, q * np.float32(2)) and np.array_equal(k2, k * np.float32(2)) and np.array_equal(v2, v * np.float32(2))
    perm = np.random.default_rng(5).permutation(B)
    _, qp, kp, vp = project(nv, DA, x[perm], None, wq, wk, wv)
    assert np.array_equal(qp, q[perm]) and np.array_equal(kp, k[perm]) and np.array_equal(vp, v[perm])
    # the skinny kernel on a decode-sized batch of the same rows: same properties, and tolerance-equal to the tiled one
    xs = x[:8]
    _, qs, ks, vs = project(nv, DA, xs, None, wq, wk, wv)
    assert close(qs, q[:8]) and close(ks, k[:8]) and close(vs, v[:8])
    _, qs2, _, _ = project(nv, DA, xs[::-1] * np.float32(2), None, wq, wk, wv)
    assert np.array_equal(qs2, qs[::-1] * np.float32(2))


def test_qkv_feeds_rope_append_and_decode(nv, DA):
    """The head of a layer on the device end to end (model.hpp:161-221): x -> rms_norm -> q/k/v -> RoPE + append ->
    paged attention, against the checker doing the same on the host."""
    H, KV, hd, bs, D = 8, 2, 64, 16, 256
    n, nb, max_blocks = 5, 64, 8
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n, D), dtype=np.float32)
    g = rng.random(D, dtype=np.float32) + np.float32(0.5)
    sc = np.float32(1.0 / np.sqrt(D))
    wq, wk, wv = (rng.standard_normal((m, D), dtype=np.float32) * sc for m in (H * hd, KV * hd, KV * hd))
    pos = np.array([0, 5, 16, 40, 99], np.int32)
    tabs = rng.permutation(nb)[: n * max_blocks].reshape(n, max_blocks).astype(np.int32)
    host_k = rng.standard_normal((nb, bs, KV, hd), dtype=np.float32)
    host_v = rng.standard_normal((nb, bs, KV, hd), dtype=np.float32)
    with nv.PagedKVPool(1, nb, bs, KV, hd) as pool:
        for b in range(nb):
            pool.upload_block(0, b, host_k[b], host_v[b])
        q, k, v = (DA(np.zeros((n, m), np.float32)) for m in (H * hd, KV * hd, KV * hd))
        xb = DA(np.zeros((n, D), np.float32))
        # on the pool's stream, like the calls that consume q/k/v (it does not synchronise with the null stream)
        nv.qkv_project_device(q, k, v, DA(x), xb, DA(g), DA(wq), DA(wk), DA(wv), n, D, H * hd, KV * hd,
                              stream=pool.stream)
        dt, dp = DA(tabs), DA(pos)
        pool.rope_append_batch_device(0, n, H, q, k, v, dt, max_blocks, dp, 10000.0)
        out = DA(np.zeros((n, H, hd), np.float32))
        nv.decode_device(pool, 0, n, H, q, dt, max_blocks, DA(pos + 1), int(pos.max()) + 1, out)
        pool.synchronize()
        got = out.numpy()
    xb_h = ol.batch_rms_norm(ol.ORACLE.oracle_batch_rms_norm, x, g)
    q_h, k_h, v_h = (ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, xb_h, w) for w in (wq, wk, wv))
    for s in range(n):
        rq, rk = ol.apply_rope(ol.ORACLE.oracle_apply_rope, q_h[s], k_h[s], int(pos[s]), hd, H, KV, 10000.0)
        host_k[tabs[s, pos[s] // bs], pos[s] % bs] = rk.reshape(KV, hd)
        host_v[tabs[s, pos[s] // bs], pos[s] % bs] = v_h[s].reshape(KV, hd)
        want = ol.oracle_paged_attention(rq.reshape(H, hd), host_k, host_v, tabs[s], int(pos[s]) + 1, bs, hd, H, KV)
        assert close(got[s], want), (s, float(np.abs(got[s] - want).max()))


@pytest.mark.parametrize("B", [1, 7, 40, 150])
def test_two_products_in_one_launch(B, nv, DA):
    """nvk_batch_matmul2_device (w_gate | w_up, model.hpp:232-233): both outputs equal the single-product calls bit for
    bit (same kernel, same column tiles' summation order) and the checker within the bar."""
    D, NA, NB = 96, 200, 72
    x = uniform(1100 + B, 0, B * D).reshape(B, D)
    wa = (uniform(1101, 0, NA * D) * np.float32(0.1)).reshape(NA, D)
    wb = (uniform(1102, 0, NB * D) * np.float32(0.1)).reshape(NB, D)
    dx, dwa, dwb = DA(x), DA(wa), DA(wb)
    oa, ob = DA(np.full((B, NA), np.nan, np.float32)), DA(np.full((B, NB), np.nan, np.float32))
    nv.batch_matmul2_device(oa, ob, dx, dwa, dwb, B, D, NA, NB)
    sa, sb = DA(np.zeros((B, NA), np.float32)), DA(np.zeros((B, NB), np.float32))
    nv.batch_matmul_device(sa, dx, dwa, B, D, NA)
    nv.batch_matmul_device(sb, dx, dwb, B, D, NB)
    assert close(oa.numpy(), ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x, wa))
    assert close(ob.numpy(), ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x, wb))
    assert np.array_equal(oa.numpy(), sa.numpy())
    assert close(ob.numpy(), sb.numpy())  # the second segment's tiles start elsewhere in k than in a launch of its own


@pytest.mark.parametrize("B", [1, 2, 5])
@pytest.mark.parametrize("H", [200, 201, 2100])
def test_gated_ffn_half_with_swiglu_in_the_epilogue(B, H, nv, DA):
    """nvk_batch_swiglu_ffn_device (model.hpp:232-234): with one or two rows SwiGLU runs in the GEMV's epilogue (two
    gate rows and two up rows per workgroup, each walking k as the plain kernel does for its column), otherwise
    after the two products.  Either way the result is silu(gate) * up of nvk_batch_matmul2_device's own outputs --
    compared here to 2 ulp (numpy's float32 exp against the device's expf) -- and the checker's within the bar."""
    D = 96
    x = uniform(1200 + B, 0, B * D).reshape(B, D)
    wg = (uniform(1201, 0, H * D) * np.float32(0.2)).reshape(H, D)
    wu = (uniform(1202, 0, H * D) * np.float32(0.2)).reshape(H, D)
    dx, dwg, dwu = DA(x), DA(wg), DA(wu)
    hb, hb2 = DA(np.full((B, H), np.nan, np.float32)), DA(np.full((B, H), np.nan, np.float32))
    nv.batch_swiglu_ffn_device(hb, hb2, dx, dwg, dwu, B, D, H)
    g, u = DA(np.zeros((B, H), np.float32)), DA(np.zeros((B, H), np.float32))
    nv.batch_matmul2_device(g, u, dx, dwg, dwu, B, D, H, H)
    gate, up = g.numpy(), u.numpy()
    want = (gate / (np.float32(1) + np.exp(-gate, dtype=np.float32))) * up
    got = hb.numpy()
    assert np.all(np.abs(got - want) <= 3e-7 * np.abs(want) + 1e-30), float(np.abs(got - want).max())
    og = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x, wg)
    ou = ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x, wu)
    assert close(got, (og / (np.float32(1) + np.exp(-og, dtype=np.float32))) * ou)
    with pytest.raises(nv.NvkError):
        nv.batch_swiglu_ffn_device(hb, hb, dx, dwg, dwu, B, D, H)  # the scratch must be a buffer of its own


def test_argument_errors(nv, DA):
    a = DA(np.zeros((4, 8), np.float32))
    with pytest.raises(nv.NvkError):
        nv.batch_matmul_device(a, a, a, 4, 6, 4)  # in_dim not a multiple of 4
    with pytest.raises(nv.NvkError):
        nv.batch_matmul_device(a, a, a, 4, 8, 4, device=99)
    with pytest.raises(nv.NvkError):
        nv.batch_matmul_device(a, None, a, 4, 8, 4)
    with pytest.raises(nv.NvkError):
        nv.qkv_project_device(a, a, a, a, None, a, a, a, a, 4, 8, 8, 8)  # rms weight given, no xb scratch
    nv.batch_matmul_device(a, a, a, 0, 8, 4)  # empty batch: nothing to do


@pytest.mark.parametrize("seed", range(8))
def test_runs_of_k_steps_over_several_segments(seed, nv, DA):
    """Stream-K launches of two and three weight segments (w_gate | w_up, q | k | v without the norm) at 17-200 rows, ragged
    in every dimension: the runs cross tile AND segment boundaries.  Every output against the checker; the same call
    twice gives the same bits."""
    d = [int(v) for v in np.random.default_rng(4400 + seed).integers(0, 1 << 30, 8)]
    B = [17, 24, 32, 33, 48, 64, 100, 200][seed]
    D = 4 * (500 + d[0] % 600)          # 2000 .. 4396, a multiple of 4 and mostly not of 64
    NA = 4 * (900 + d[1] % 900)         # 3600 .. 7196
    NB = 4 * (200 + d[2] % 400)
    x = uniform(4500 + seed, 0, B * D).reshape(B, D)
    sc = np.float32(1.0 / np.sqrt(D))
    wa = (uniform(4510 + seed, 0, NA * D) * sc).reshape(NA, D)
    wb = (uniform(4520 + seed, 0, NB * D) * sc).reshape(NB, D)
    rows = sorted({0, B // 2, B - 1})
    dx, dwa, dwb = DA(x), DA(wa), DA(wb)
    oa, ob = DA(np.full((B, NA), np.nan, np.float32)), DA(np.full((B, NB), np.nan, np.float32))
    nv.batch_matmul2_device(oa, ob, dx, dwa, dwb, B, D, NA, NB)
    ga, gb = oa.numpy(), ob.numpy()
    assert np.isfinite(ga).all() and np.isfinite(gb).all()
    assert close(ga[rows], ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x[rows], wa))
    assert close(gb[rows], ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x[rows], wb))
    nv.batch_matmul2_device(oa, ob, dx, dwa, dwb, B, D, NA, NB)
    assert np.array_equal(oa.numpy(), ga) and np.array_equal(ob.numpy(), gb)
    # three segments: q | k | v of the same activations, no norm
    q, k, v = (DA(np.full((B, n), np.nan, np.float32)) for n in (NA, NB, NB))
    nv.qkv_project_device(q, k, v, dx, None, None, dwa, dwb, dwb, B, D, NA, NB)
    assert close(q.numpy()[rows], ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x[rows], wa))
    assert close(k.numpy()[rows], ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x[rows], wb))
    assert close(v.numpy()[rows], ol.batch_matmul(ol.ORACLE.oracle_batch_matmul, x[rows], wb))
    # (k and v have the same weights here but sit at different places of the run order: other cuts, other last bits)

"""GPU parity of each reference operator against the CPU oracle,
through the C ABI.  Tolerances: products/sums are fp32 with different (but fixed)
summation orders on the two sides -> norm-wise 2e-6 * sqrt-ish growth; element-wise
IEEE ops (divide, multiply, clamp) must be bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand(shape, seed, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.uniform(lo, hi, size=shape).astype(np.float32))


def _dev(ng, a):
    return ng.Matrix(a).to_device()


def _tol64(k):
    """fp32 fmaf-chain error against fp64 grows with the chain length (1e-6 up to K = 4096 on uniform data)"""
    return 1e-6 if k <= 4096 else 3e-6


GEMM_SHAPES = [(128, 128, 64), (1024, 4096, 64), (4096, 352, 128), (70, 45, 33), (1, 1, 1), (129, 257, 17),
               (256, 96, 300), (300, 100, 9000),   # takes the split-K path of A*B'
               (256, 384, 144), (128, 128, 16), (384, 256, 4112),   # tile-aligned: the 16-B-load fast path (+ split-K)
               (2048, 80, 320), (1024, 48, 512), (4096, 1040, 256)]  # tall A, K <= 512: A*B through the 16-column kernel's product 1


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
def test_matrix_multiply_nn(ng, oracle, m, n, k):
    """cuda/matrix.cu:97-105; A = I check with an asymmetric B guards the C/D lane map."""
    A, B = _rand((m, k), 1), _rand((k, n), 2)
    a, b, c = _dev(ng, A), _dev(ng, B), ng.Matrix(rows=m, cols=n).to_device()
    ng.matrix_multiply(a, b, c)
    got = c.from_device().mat
    want = oracle.sgemm("nn", A, B)
    assert oracle.relF(got, want) < 2e-6
    ref64 = A.astype(np.float64) @ B.astype(np.float64)
    assert oracle.relF(got, ref64) < _tol64(k)


def test_matrix_multiply_identity_asymmetric(ng):
    n = 96
    I = np.asfortranarray(np.eye(n, dtype=np.float32))
    B = np.asfortranarray((np.arange(n * 80, dtype=np.float32).reshape(n, 80) % 251) + 1.0)
    c = ng.Matrix(rows=n, cols=80).to_device()
    ng.matrix_multiply(_dev(ng, I), _dev(ng, B), c)
    assert np.array_equal(c.from_device().mat, B)
    c2 = ng.Matrix(rows=n, cols=80).to_device()
    ng.matrix_multiply_AtB(_dev(ng, I), _dev(ng, B), c2)
    assert np.array_equal(c2.from_device().mat, B)
    Bt = np.asfortranarray(B.T)
    c3 = ng.Matrix(rows=n, cols=80).to_device()
    ng.matrix_multiply_ABt(_dev(ng, I), _dev(ng, Bt), c3)
    assert np.array_equal(c3.from_device().mat, B)


@pytest.mark.parametrize("k", [64, 96, 128, 160, 224, 256, 288, 384, 512])
def test_matrix_multiply_wh_shape_exact_lane_map(ng, k):
    """The W*H-shaped product (tall A, K <= 512) runs on the 16-column kernel's product 1 with B held in registers:
    with B a 0/1 selection matrix every output column must be an exact copy of one column of A."""
    m, n = 1024, 80
    A = np.asfortranarray((np.arange(m * k, dtype=np.float32).reshape(m, k) % 8191) * np.float32(0.25) + 1.0)
    sel = (np.arange(n) * 37 + 5) % k
    B = np.zeros((k, n), dtype=np.float32, order="F")
    B[sel, np.arange(n)] = 1.0
    c = ng.Matrix(rows=m, cols=n).to_device()
    ng.matrix_multiply(_dev(ng, A), _dev(ng, B), c)
    assert np.array_equal(c.from_device().mat, A[:, sel])


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
def test_matrix_multiply_AtB(ng, oracle, m, n, k):
    """c(m x n) = a'(m x k) b(k x n), a stored k x m  (cuda/matrix.cu:107-115)"""
    A, B = _rand((k, m), 3), _rand((k, n), 4)
    c = ng.Matrix(rows=m, cols=n).to_device()
    ng.matrix_multiply_AtB(_dev(ng, A), _dev(ng, B), c)
    got = c.from_device().mat
    assert oracle.relF(got, oracle.sgemm("tn", A, B)) < 2e-6
    assert oracle.relF(got, A.astype(np.float64).T @ B.astype(np.float64)) < _tol64(k)


@pytest.mark.parametrize("m,n,k", GEMM_SHAPES)
def test_matrix_multiply_ABt(ng, oracle, m, n, k):
    """c(m x n) = a(m x k) b'(k x n), b stored n x k  (cuda/matrix.cu:117-125)"""
    A, B = _rand((m, k), 5), _rand((n, k), 6)
    c = ng.Matrix(rows=m, cols=n).to_device()
    ng.matrix_multiply_ABt(_dev(ng, A), _dev(ng, B), c)
    got = c.from_device().mat
    assert oracle.relF(got, oracle.sgemm("nt", A, B)) < 2e-6
    assert oracle.relF(got, A.astype(np.float64) @ B.astype(np.float64).T) < _tol64(k)


def test_gemm_shape_errors(ng):
    a, b, c = ng.Matrix(rows=4, cols=3).to_device(), ng.Matrix(rows=5, cols=2).to_device(), ng.Matrix(rows=4, cols=2).to_device()
    for f in (ng.matrix_multiply, ng.matrix_multiply_AtB, ng.matrix_multiply_ABt):
        with pytest.raises(ng.NmfError) as e:
            f(a, b, c)
        assert e.value.status == 2


def test_elementwise_bit_exact(ng, oracle):
    """vec_div / vec_mul / set_epsilon: IEEE fp32, must match the CPU bit for bit
    (cuda/matrix.cu:146-188)."""
    A, B = _rand((333, 77), 7, 0.0, 2.0), _rand((333, 77), 8, 1e-3, 3.0)
    A[0, 0] = 0.0; A[1, 0] = 1e-20; A[2, 0] = np.float32(2.2204e-16); A[3, 0] = -1.0
    a, b = _dev(ng, A), _dev(ng, B)
    c = ng.Matrix(rows=333, cols=77).to_device()
    ng.element_divide(a, b, c)
    assert np.array_equal(c.from_device().mat, A / B)
    ng.element_multiply(a, b, c)
    assert np.array_equal(c.from_device().mat, A * B)
    ng.set_epsilon(a)
    want = oracle.clamp(A)
    got = a.from_device().mat
    assert np.array_equal(got, want)
    assert got[0, 0] == np.float32(2.2204e-16) and got[3, 0] == np.float32(2.2204e-16)
    # NaN passes through the clamp unchanged (NaN < EPS is false, cuda/matrix.cu:185)
    N = np.asfortranarray(np.array([[np.nan, 0.5]], dtype=np.float32))
    n = _dev(ng, N)
    ng.set_epsilon(n)
    out = n.from_device().mat
    assert np.isnan(out[0, 0]) and out[0, 1] == 0.5


def test_row_col_divide(ng):
    """row_div: column j of a divided by b[j] (cuda/matrix.cu:220-224);
    col_div: row i of a divided by b[i] (cuda/matrix.cu:244-250).  Bit-exact."""
    A = _rand((130, 37), 9, 0.1, 5.0)
    bc = _rand((37, 1), 10, 0.5, 2.0)   # column vector, length = cols(a)
    br = _rand((1, 130), 11, 0.5, 2.0)  # row vector, length = rows(a)
    a = _dev(ng, A)
    c = ng.Matrix(rows=130, cols=37).to_device()
    ng.row_divide(a, _dev(ng, bc), c)
    assert np.array_equal(c.from_device().mat, A / bc.reshape(1, 37))
    ng.col_divide(a, _dev(ng, br), c)
    assert np.array_equal(c.from_device().mat, A / br.reshape(130, 1))
    with pytest.raises(ng.NmfError):
        ng.row_divide(a, _dev(ng, br), c)
    with pytest.raises(ng.NmfError):
        ng.col_divide(a, _dev(ng, bc), c)


@pytest.mark.parametrize("rows,cols", [(4096, 128), (1024, 64), (33, 7), (1, 5), (70000, 3), (256, 300)])
def test_sum_cols_rows(ng, oracle, rows, cols):
    """sum_cols / sum_rows (cuda/matrix.cu:261-503, 642-735), intended result (all partials)."""
    A = _rand((rows, cols), 12)
    a = _dev(ng, A)
    oc = ng.Matrix(rows=1, cols=cols).to_device()
    ng.sum_cols(a, oc)
    got = oc.from_device().mat.ravel()
    want64 = A.astype(np.float64).sum(axis=0)
    assert np.allclose(got, want64, rtol=3e-6, atol=0)
    assert np.allclose(got, oracle.sum_cols(A), rtol=3e-6, atol=0)
    orr = ng.Matrix(rows=rows, cols=1).to_device()
    ng.sum_rows(a, orr)
    got = orr.from_device().mat.ravel()
    assert np.allclose(got, A.astype(np.float64).sum(axis=1), rtol=3e-6, atol=0)
    assert np.allclose(got, oracle.sum_rows(A), rtol=3e-6, atol=0)


def test_kl_and_diff(ng, oracle):
    """reduce1d_div / reduce1d_diff (cuda/matrix.cu:505-640)."""
    X, Y = _rand((700, 333), 13, 1e-3, 1.0), _rand((700, 333), 14, 1e-3, 1.0)
    x, y = _dev(ng, X), _dev(ng, Y)
    kl = ng.kl_divergence(x, y)
    assert abs(kl - oracle.kl_div(X, Y)) / oracle.kl_div(X, Y) < 1e-5
    d, a = ng.diff_norm(x, y)
    assert abs(d / a - oracle.rel_l1(X, Y)) < 1e-6
    # KL(X || X) == 0 exactly term by term
    assert ng.kl_divergence(x, x) == 0.0

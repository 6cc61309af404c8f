"""The algebra behind the Cholesky form of the marginalisation prior (include/sadvio_ba.h, docs/KERNEL_HISTORY.md "Round 4"), stated in NumPy so
that the claims the device code relies on are executable without a GPU:
  * J = L^T, r0 = -L^-1 bk (bk carried through the factorisation as its right-hand side) is a MarginalizationFactor with the same
    J^T J, J^T r0 and hence the same Gauss-Newton system as the reference's eigen form J = Lambda^1/2 U^T, r0 = -Lambda^-1/2 U^T bk
    (marginalization.cpp:318-342, 516-530);
  * the augmented matrix [[Ak, bk], [bk^T, -1]] of the pivoted route yields the same r0 as an extra column;
  * an UNPIVOTED factorisation is not rank revealing: the pivot of the last index of a dependent set is lambda / v_i^2 for the null
    vector v, so its size says nothing about the rank — why the noise-floor mode always pivots;
  * threshold pivoting (a pivot is at least theta times the largest remaining diagonal) bounds the entries of the factor by
    sqrt(1 / theta) of the greedy order's bound."""
import numpy as np


def spd(n, rng, cond=1e6):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.logspace(0, np.log10(cond), n)
    return (Q * lam) @ Q.T, Q, lam


def test_cholesky_form_is_the_same_factor_as_the_eigen_form():
    rng = np.random.default_rng(1)
    Ak, Q, lam = spd(40, rng)
    bk = rng.standard_normal(40)
    L = np.linalg.cholesky(Ak)
    Jc, r0c = L.T, -np.linalg.solve(L, bk)
    Je, r0e = np.sqrt(lam)[:, None] * Q.T, -(Q.T @ bk) / np.sqrt(lam)
    for J, r0 in ((Jc, r0c), (Je, r0e)):
        assert np.allclose(J.T @ J, Ak, rtol=0, atol=1e-9 * lam.max())
        assert np.allclose(J.T @ r0, -bk, atol=1e-9)
    assert np.isclose(r0c @ r0c, r0e @ r0e, rtol=1e-9)      # |r0|^2 = bk^T Ak^-1 bk either way
    # the factor's residual at a state change dx: same cost
    dx = 1e-2 * rng.standard_normal(40)
    assert np.isclose(np.sum((r0c + Jc @ dx) ** 2), np.sum((r0e + Je @ dx) ** 2), rtol=1e-10)


def test_augmented_matrix_carries_r0_as_an_extra_column():
    rng = np.random.default_rng(2)
    Ak, _, _ = spd(25, rng, cond=1e4)
    bk = rng.standard_normal(25)
    S = np.block([[Ak, bk[:, None]], [bk[None, :], -np.ones((1, 1))]])
    # right-looking Cholesky on the first 25 indices only (the extra index has a negative diagonal: never a pivot)
    G = np.zeros((25, 26))
    W = S.copy()
    for k in range(25):
        G[k, k:] = W[k, k:] / np.sqrt(W[k, k])
        W[k + 1:, k + 1:] -= np.outer(G[k, k + 1:], G[k, k + 1:])
    L = np.linalg.cholesky(Ak)
    assert np.allclose(G[:, :25], L.T, atol=1e-10)
    assert np.allclose(-G[:, 25], -np.linalg.solve(L, bk), atol=1e-10)


def test_unpivoted_cholesky_is_not_rank_revealing():
    """A (numerically) singular PSD matrix whose null vector has a small component on the LAST index: the unpivoted factorisation's last
    pivot is lambda_min / v_last^2 — here eight orders of magnitude above lambda_min — while the diagonally pivoted one ends on a
    pivot of the order of lambda_min."""
    rng = np.random.default_rng(3)
    n = 30
    v = rng.standard_normal(n); v[-1] = 1e-4; v /= np.linalg.norm(v)
    Q, _ = np.linalg.qr(np.column_stack([v, rng.standard_normal((n, n - 1))]))
    lam = np.concatenate([[1e-10], np.logspace(0, 2, n - 1)])
    A = (Q * lam) @ Q.T
    L = np.linalg.cholesky(A)
    last_unpivoted = L[-1, -1] ** 2
    Ainv_nn = np.sum(Q[-1] ** 2 / lam)                                            # (A^-1)_nn = v_n^2 / lambda_min + the other eigen-pairs' share
    assert np.isclose(last_unpivoted, 1.0 / Ainv_nn, rtol=1e-3)                   # the last pivot is the Schur complement 1 / (A^-1)_nn ...
    assert 0.5 * lam[0] / Q[-1, 0] ** 2 <= last_unpivoted <= lam[0] / Q[-1, 0] ** 2   # ... i.e. lambda / v_i^2 up to the well-conditioned rest
    assert last_unpivoted > 1e6 * lam[0]
    # diagonal pivoting
    W = A.copy(); idx = list(range(n)); piv = []
    for k in range(n):
        j = k + int(np.argmax(np.diag(W)[k:]))
        W[[k, j]] = W[[j, k]]; W[:, [k, j]] = W[:, [j, k]]
        piv.append(W[k, k])
        c = W[k + 1:, k] / W[k, k]
        W[k + 1:, k + 1:] -= np.outer(c, W[k + 1:, k])
    assert piv[-1] < 1e2 * lam[0]


def test_threshold_pivoting_bounds_the_factor():
    """With pivots d_k >= theta max_j d_j the entries of column k obey |L_ik| <= sqrt(d_i d_k) / sqrt(d_k) <= sqrt(max d / theta) ...
    i.e. L_ik^2 <= d_i <= max_j d_j <= d_k / theta: every entry of a column is within sqrt(1 / theta) of its diagonal entry."""
    rng = np.random.default_rng(4)
    A, _, _ = spd(60, rng, cond=1e8)
    theta = 0.1
    n = 60
    W = A.copy(); done = np.zeros(n, bool); L = np.zeros((n, n)); order = []
    for k in range(n):
        d = np.where(done, -np.inf, np.diag(W))
        ok = np.flatnonzero(d >= theta * d.max())
        j = int(rng.choice(ok))                     # ANY index that passes the threshold, not the arg max
        order.append(j)
        col = W[:, j] / np.sqrt(W[j, j]); col[done] = 0.0
        L[:, k] = col
        W -= np.outer(col, col)
        done[j] = True
    P = np.array(order)
    assert np.allclose(L @ L.T, A, atol=1e-9 * np.abs(A).max())
    for k in range(n):
        assert np.abs(L[:, k]).max() <= np.sqrt(1.0 / theta) * L[P[k], k] * (1 + 1e-9)


def _pivoted_cholesky(A):
    """Diagonally pivoted Cholesky A[p][:, p] = G^T G (G upper triangular in pivot order); returns G (rows by ORIGINAL column) and the order."""
    n = A.shape[0]
    S = A.copy()
    G = np.zeros((n, n))
    left = list(range(n))
    order = []
    for k in range(n):
        p = max(left, key=lambda i: S[i, i])
        order.append(p); left.remove(p)
        d = np.sqrt(S[p, p])
        G[k, p] = d
        for i in left:
            G[k, i] = S[p, i] / d
        for i in left:
            for j in left:
                S[i, j] -= G[k, i] * G[k, j]
    return G, order


def test_small_eigenvalue_from_the_trailing_pivot():
    """Round 6 (ba_capi.hip: refine_rank_by_eigenvalue): the reference cuts the prior's rank by EIGENVALUE (lambda > 1e-12,
    marginalization.cpp:318-342); a rank-revealing Cholesky sees pivots. With G in pivot order and x = G^-1 e_last, 1 / |x|^2 — the
    Rayleigh quotient of the near-null vector, d_last / (1 + |w|^2) — is the smallest eigenvalue to O(d / gap), with RELATIVE accuracy
    on a graded matrix whose double-precision eigen-decomposition only returns noise of size eps |A|: checked against mpmath at 50
    digits on a matrix of the measured kind (lambda_max 1e8, lambda_min 1e-14: below the cut under a pivot that is not)."""
    import mpmath as mp
    rng = np.random.default_rng(5)
    n = 24
    B = rng.standard_normal((n, n)); B = B @ B.T / n + np.eye(n)          # well conditioned
    D = np.diag(np.logspace(4, -1, n)); D[-1, -1] = 3e-8                  # graded scaling: the last direction carries ~ 1e-15 .. 1e-14
    A = D @ B @ D
    A = 0.5 * (A + A.T)
    mp.mp.dps = 50
    ev = mp.eigsy(mp.matrix(A.tolist()), eigvals_only=True)
    lam_min = float(min(ev))
    assert 1e-16 < lam_min < 1e-12 and float(max(ev)) > 1e7
    G, order = _pivoted_cholesky(A)
    Gp = G[:, order]                                                       # upper triangular
    assert np.allclose(np.tril(Gp, -1), 0.0)
    d_last = Gp[-1, -1] ** 2
    x = np.linalg.solve(Gp, np.eye(n)[:, -1])
    est = 1.0 / (x @ x)
    assert lam_min <= d_last * (1 + 1e-12)                                 # lambda_min <= the last pivot: a pivot <= cut certifies the drop
    assert abs(est - lam_min) <= 1e-6 * lam_min, (est, lam_min)            # the estimate: relative accuracy
    lapack = np.linalg.eigvalsh(A)[0]
    assert abs(lapack - lam_min) > 1e3 * abs(est - lam_min)                # a double-precision eigen-decomposition of A: noise at this scale


def test_trace_of_the_inverse_bounds_the_smallest_eigenvalue_where_pivots_do_not():
    """Round 6, second half (sadvio_ba_marginalize: Amm^+; sadvio_ba_sparsify: Sigma_k). A block with ONE eigenvalue below the reference's
    cut (1e-12, marginalization.hpp:58) whose eigenvector is spread over all coordinates: (i) every pivot of its Cholesky factorisation
    is far above the pivot floor - pivots bound eigenvalues from above only; (ii) the reference's pseudo-inverse zeroes that direction,
    an inverse divides by it: the Schur complement Arr - Arm Amm^+ Amr then differs by (Arm v)(Arm v)^T / lambda - at step 13 of the
    config-3-size dense sequence that was 9 187 of information along the kept frame's rotation; (iii) trace(Amm^-1), which the Cholesky
    route has from the triangular inverse it forms anyway, proves lambda_min >= 1 / trace, so a trace below 1e12 certifies the inverse
    and a larger one sends the call to the eigen-decomposition."""
    rng = np.random.default_rng(7)
    m = 48
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    lam = np.concatenate([[5e-13], np.logspace(-2, 1, m - 1)])      # (|Amm| = 10: the float64 image of the matrix keeps the small eigenvalue to a few per cent)
    Amm = (Q * lam) @ Q.T
    Amm = 0.5 * (Amm + Amm.T)
    L = np.linalg.cholesky(Amm + 0.0)                       # it factorises: no pivot is even close to zero
    piv = np.diag(L) ** 2
    floor = 1e-12 / m
    assert piv.min() > 1e2 * floor and piv.min() > 1e-12    # (i) the pivot test passes with a wide margin (the smallest pivot is lambda / v_i^2 >> lambda) ...
    assert lam.min() < 1e-12                                # ... although one eigenvalue lies below the cut
    Z = np.linalg.inv(L)                                    # Amm^-1 = Z^T Z
    tr = float((Z * Z).sum())
    assert tr >= 1e12 and 1.0 / tr <= lam.min() * 1.05            # (iii) the bound is rigorous (up to the rounding of the matrix itself) and it fires
    # (ii) what the two "inverses" do to a Schur complement
    Arm = rng.standard_normal((6, m))
    pinv = (Q[:, 1:] / lam[1:]) @ Q[:, 1:].T                # the reference: eigenvalues <= 1e-12 dropped
    inv = Z.T @ Z
    diff = Arm @ (inv - pinv) @ Arm.T
    u = Arm @ Q[:, 0]
    expect = np.outer(u, u) / lam[0]
    assert np.allclose(diff, expect, rtol=1e-3, atol=1e-6 * np.abs(expect).max())
    assert np.abs(diff).max() > 1e9                         # not a rounding-level difference
    # a block whose smallest eigenvalue is comfortably above the cut is certified by its trace
    lam2 = np.logspace(-6, 1, m)
    A2 = (Q * lam2) @ Q.T
    Z2 = np.linalg.inv(np.linalg.cholesky(0.5 * (A2 + A2.T)))
    assert (Z2 * Z2).sum() < 1e12

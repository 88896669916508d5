"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot finish N^2 there): a target subset
against the oracle with ALL sources, linearity in the density, translation invariance, split-invariance."""
import numpy as np
import pytest

import sctl_amd
from conftest import ctx_for, rel_l2

pytestmark = pytest.mark.gpu


def _cloud(seed, Nt, Ns, info, dt=np.float64):
    rng = np.random.default_rng(seed)
    return (rng.random(Nt * 3).astype(dt), rng.random(Ns * 3).astype(dt), (rng.random(Ns * info["nd"]) - 0.5).astype(dt),
            (rng.random(Ns * info["k0"]) - 0.5).astype(dt))


@pytest.mark.parametrize("name,N,dt,tol", [
    ("Laplace3D-FxU", 1 << 20, np.float64, 1e-12),      # BASELINE config 2 size, headline kernel
    ("Laplace3D-DxU", 1 << 20, np.float64, 1e-12),      # double layer on the tile-centred path at full size
    ("Laplace3D-FDxUdU", 1 << 20, np.float64, 1e-12),   # config 2 (SL+DL potential+gradient) at its full size
    ("Stokes3D-FxU", 1 << 18, np.float64, 1e-12),       # config 3
    ("Helmholtz3D-FxU", 1 << 20, np.float64, 1e-12),    # config 5 (complex wavenumber) at its full size
    ("Laplace3D-FxU", 1 << 21, np.float32, 1e-4),       # config 4 precision (tolerance vs the f64 oracle)
    ("Laplace3D-FxU", 1 << 23, np.float32, 1e-4),       # config 4 at its full size (2^23 x 2^23; ~11 s on one GPU)
])
def test_full_size_target_subset_against_oracle(O, name, N, dt, tol):
    import torch
    info = sctl_amd.kernel_info(name)
    xt, xs, xn, f = _cloud(21, N, N, info, dt)
    d = [torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]
    u = sctl_amd.eval_device(name, *d, ctx=ctx_for(name)).cpu().numpy().reshape(N, info["k1"])
    assert np.all(np.isfinite(u))
    sel = np.random.default_rng(5).choice(N, 512, replace=False)
    xt64 = xt.reshape(N, 3)[sel].astype(np.float64).ravel().copy()
    ref = O.eval(name, xt64, xs.astype(np.float64), xn.astype(np.float64), f.astype(np.float64), ctx=ctx_for(name)).reshape(512, info["k1"])
    assert rel_l2(u[sel], ref) <= tol, rel_l2(u[sel], ref)


def test_linearity_and_translation_invariance_at_scale():
    import torch
    name = "Stokes3D-FxU"
    info = sctl_amd.kernel_info(name)
    N = 1 << 17
    xt, xs, xn, f1 = _cloud(31, N, N, info)
    f2 = np.random.default_rng(32).random(f1.size) - 0.5
    dxt, dxs, df1, df2 = (torch.from_numpy(a).cuda() for a in (xt, xs, f1, f2))
    u1 = sctl_amd.eval_device(name, dxt, dxs, None, df1)
    u2 = sctl_amd.eval_device(name, dxt, dxs, None, df2)
    u12 = sctl_amd.eval_device(name, dxt, dxs, None, 2.0 * df1 - 3.0 * df2)
    assert rel_l2(u12.cpu().numpy(), (2.0 * u1 - 3.0 * u2).cpu().numpy()) < 1e-12
    # translation by a power of two keeps every difference x_t - x_s exact
    shift = 4.0
    us = sctl_amd.eval_device(name, dxt + shift, dxs + shift, None, df1)
    assert rel_l2(us.cpu().numpy(), u1.cpu().numpy()) < 1e-13
    # accumulating into the result of a first call == evaluating twice
    u_acc = sctl_amd.eval_device(name, dxt, dxs, None, df2, v_trg=u1.clone())
    assert rel_l2(u_acc.cpu().numpy(), (u1 + u2).cpu().numpy()) < 1e-14


@pytest.mark.parametrize("N", [1 << 16, 300000])
@pytest.mark.parametrize("name,dt,tol", [("Laplace3D-FxU", np.float64, 1e-12), ("Stokes3D-DxU", np.float64, 1e-12), ("Laplace3D-FxdU", np.float32, 1e-4)])
def test_shuffled_self_interaction_repairs_masked_tiles(O, N, name, dt, tol):
    """targets == sources in SHUFFLED order: every lane meets its coincident source in a different LDS tile.  The kernel
    evaluates tiles without the r = 0 mask, detects the poisoned tile sums and re-runs those tiles masked (N = 300000:
    ~5 % of tiles per wave); at N = 2^16 more than 1/8 of a wave's tiles need repair and it switches to the masked loop."""
    import torch
    info = sctl_amd.kernel_info(name)
    rng = np.random.default_rng(77)
    xs = rng.random(N * 3).astype(dt)
    perm = rng.permutation(N)
    xt = xs.reshape(N, 3)[perm].ravel().copy()
    xn = (rng.random(N * info["nd"]) - 0.5).astype(dt)
    f = (rng.random(N * info["k0"]) - 0.5).astype(dt)
    u = sctl_amd.eval_device(name, *[torch.from_numpy(a).cuda() for a in (xt, xs, xn, f)]).cpu().numpy().reshape(N, info["k1"])
    assert np.all(np.isfinite(u))
    sel = rng.choice(N, 256, replace=False)
    ref = O.eval(name, xt.reshape(N, 3)[sel].astype(np.float64).ravel().copy(), xs.astype(np.float64), xn.astype(np.float64),
                 f.astype(np.float64)).reshape(256, info["k1"])
    assert rel_l2(u[sel], ref) <= tol, rel_l2(u[sel], ref)

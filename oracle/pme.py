"""oracle/pme.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

numpy (float64) restatement of the Reference platform's smooth particle-mesh Ewald,
platforms/reference/src/SimTKReference/ReferencePME.cpp (paths relative to the OpenMM tree):

  bspline_moduli          pme_calculate_bsplines_moduli   :98-193
  invert_box_vectors      invert_box_vectors              :196-204
  grid_index_and_fraction pme_update_grid_index_and_fraction :206-266
  bsplines                pme_update_bsplines             :274-327
  spread_charge           pme_grid_spread_charge          :330-405
  reciprocal_convolution  pme_reciprocal_convolution      :409-514
  interpolate_force       pme_grid_interpolate_force      :617-713
  pme_exec                pme_exec                        :760-803  (3-D FFT = numpy.fft, unnormalised
                                                           forward and backward like fftpack_exec_3d)
Pinned against the real reference in tests/test_oracle_vs_reference.py.
"""
import numpy as np

ONE_4PI_EPS0 = 138.93545764438198  # 1/(4 pi EPSILON0), SimTKOpenMMRealType.h:74-89 (CODATA 2018)
ORDER = 5  # ReferenceLJCoulombIxn.cpp:243


def bspline_moduli(n, order=ORDER):
    data = np.zeros(order)
    data[0] = 1.0
    for k in range(3, order):
        div = 1.0 / (k - 1.0)
        data[k - 1] = 0.0
        for l in range(1, k - 1):
            data[k - l - 1] = div * (l * data[k - l - 2] + (k - l) * data[k - l - 1])
        data[0] = div * data[0]
    div = 1.0 / (order - 1)
    data[order - 1] = 0.0
    for l in range(1, order - 1):
        data[order - l - 1] = div * (l * data[order - l - 2] + (order - l) * data[order - l - 1])
    data[0] = div * data[0]
    bs = np.zeros(n)
    bs[1:order + 1] = data
    j = np.arange(n)
    mod = np.empty(n)
    for i in range(n):
        arg = 2.0 * np.pi * i * j / n
        sc = np.sum(bs * np.cos(arg))
        ss = np.sum(bs * np.sin(arg))
        mod[i] = sc * sc + ss * ss
    out = mod.copy()
    for i in range(n):
        if mod[i] < 1.0e-7:
            out[i] = 0.5 * (out[(i - 1 + n) % n] + out[(i + 1) % n])
            mod[i] = out[i]     # the reference updates in place while sweeping
    return out


def invert_box_vectors(box):
    box = np.asarray(box, dtype=np.float64)
    det = box[0, 0] * box[1, 1] * box[2, 2]
    r = np.zeros((3, 3))
    r[0] = [box[1, 1] * box[2, 2], 0, 0]
    r[1] = [-box[1, 0] * box[2, 2], box[0, 0] * box[2, 2], 0]
    r[2] = [box[1, 0] * box[2, 1] - box[1, 1] * box[2, 0], -box[0, 0] * box[2, 1], box[0, 0] * box[1, 1]]
    return r / det


def grid_index_and_fraction(pos, recip, ngrid):
    t = np.asarray(pos, dtype=np.float64) @ recip      # t[i,d] = sum_k pos[i,k]*recip[k,d]
    t = (t - np.floor(t)) * np.asarray(ngrid)[None, :]
    ti = t.astype(np.int64)
    return ti % np.asarray(ngrid)[None, :], t - ti


def bsplines(frac, order=ORDER):
    """frac [N,3] -> theta, dtheta [N,3,order]."""
    n = frac.shape[0]
    th = np.zeros((n, 3, order))
    dth = np.zeros((n, 3, order))
    dr = frac
    th[:, :, order - 1] = 0.0
    th[:, :, 1] = dr
    th[:, :, 0] = 1.0 - dr
    for k in range(3, order):
        div = 1.0 / (k - 1.0)
        th[:, :, k - 1] = div * dr * th[:, :, k - 2]
        for l in range(1, k - 1):
            th[:, :, k - l - 1] = div * ((dr + l) * th[:, :, k - l - 2] + (k - l - dr) * th[:, :, k - l - 1])
        th[:, :, 0] = div * (1.0 - dr) * th[:, :, 0]
    dth[:, :, 0] = -th[:, :, 0]
    for k in range(1, order):
        dth[:, :, k] = th[:, :, k - 1] - th[:, :, k]
    div = 1.0 / (order - 1)
    th[:, :, order - 1] = div * dr * th[:, :, order - 2]
    for l in range(1, order - 1):
        th[:, :, order - l - 1] = div * ((dr + l) * th[:, :, order - l - 2] + (order - l - dr) * th[:, :, order - l - 1])
    th[:, :, 0] = div * (1.0 - dr) * th[:, :, 0]
    return th, dth


def _stencil_indices(index, ngrid, order=ORDER):
    off = np.arange(order)
    gx = (index[:, 0, None] + off[None, :]) % ngrid[0]
    gy = (index[:, 1, None] + off[None, :]) % ngrid[1]
    gz = (index[:, 2, None] + off[None, :]) % ngrid[2]
    return gx, gy, gz


def spread_charge(index, theta, charge, ngrid, order=ORDER):
    grid = np.zeros(tuple(ngrid))
    gx, gy, gz = _stencil_indices(index, ngrid, order)
    q = np.asarray(charge, dtype=np.float64)
    w = q[:, None, None, None] * theta[:, 0, :, None, None] * theta[:, 1, None, :, None] * theta[:, 2, None, None, :]
    np.add.at(grid, (gx[:, :, None, None] + 0 * gy[:, None, :, None] + 0 * gz[:, None, None, :],
                     0 * gx[:, :, None, None] + gy[:, None, :, None] + 0 * gz[:, None, None, :],
                     0 * gx[:, :, None, None] + 0 * gy[:, None, :, None] + gz[:, None, None, :]), w)
    return grid


def influence_function(ngrid, box, recip, alpha, moduli=None):
    """eterm on the full grid (0 at k=0).  ReferencePME.cpp:436-497."""
    nx, ny, nz = ngrid
    if moduli is None:
        moduli = [bspline_moduli(n) for n in ngrid]
    kx, ky, kz = np.arange(nx), np.arange(ny), np.arange(nz)
    mx = np.where(kx < (nx + 1) // 2, kx, kx - nx).astype(np.float64)
    my = np.where(ky < (ny + 1) // 2, ky, ky - ny).astype(np.float64)
    mz = np.where(kz < (nz + 1) // 2, kz, kz - nz).astype(np.float64)
    mhx = mx[:, None, None] * recip[0, 0]
    mhy = mx[:, None, None] * recip[1, 0] + my[None, :, None] * recip[1, 1]
    mhz = mx[:, None, None] * recip[2, 0] + my[None, :, None] * recip[2, 1] + mz[None, None, :] * recip[2, 2]
    m2 = mhx ** 2 + mhy ** 2 + mhz ** 2
    volume = box[0][0] * box[1][1] * box[2][2]
    denom = m2 * (np.pi * volume * moduli[0][:, None, None]) * moduli[1][None, :, None] * moduli[2][None, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        eterm = ONE_4PI_EPS0 * np.exp(-(np.pi ** 2 / alpha ** 2) * m2) / denom
    eterm[0, 0, 0] = 0.0
    return eterm


def interpolate_force(index, theta, dtheta, charge, grid, ngrid, recip, order=ORDER):
    gx, gy, gz = _stencil_indices(index, ngrid, order)
    g = grid[gx[:, :, None, None], gy[:, None, :, None], gz[:, None, None, :]]
    fx = np.einsum("nabc,na,nb,nc->n", g, dtheta[:, 0], theta[:, 1], theta[:, 2])
    fy = np.einsum("nabc,na,nb,nc->n", g, theta[:, 0], dtheta[:, 1], theta[:, 2])
    fz = np.einsum("nabc,na,nb,nc->n", g, theta[:, 0], theta[:, 1], dtheta[:, 2])
    q = np.asarray(charge, dtype=np.float64)
    nx, ny, nz = ngrid
    f = np.zeros((len(q), 3))
    f[:, 0] = -q * (fx * nx * recip[0, 0])
    f[:, 1] = -q * (fx * nx * recip[1, 0] + fy * ny * recip[1, 1])
    f[:, 2] = -q * (fx * nx * recip[2, 0] + fy * ny * recip[2, 1] + fz * nz * recip[2, 2])
    return f


def pme_exec(pos, charge, box, alpha, ngrid):
    """Reciprocal-space forces [N,3] and energy (no self term).  ReferencePME.cpp:760-803."""
    box = np.asarray(box, dtype=np.float64)
    recip = invert_box_vectors(box)
    index, frac = grid_index_and_fraction(pos, recip, ngrid)
    theta, dtheta = bsplines(frac)
    grid = spread_charge(index, theta, charge, ngrid)
    fgrid = np.fft.fftn(grid)                      # unnormalised forward, sign -1 (fftpack forward)
    eterm = influence_function(ngrid, box, recip, alpha)
    energy = 0.5 * float(np.sum(eterm * np.abs(fgrid) ** 2))
    conv = np.fft.ifftn(fgrid * eterm) * np.prod(ngrid)   # unnormalised backward
    forces = interpolate_force(index, theta, dtheta, charge, conv.real, ngrid, recip)
    return forces, energy

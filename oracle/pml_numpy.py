"""CPU oracle for the build-defined absorbing layer (`boundary="pml"`) -- TEST INFRASTRUCTURE.

PARITY UNPINNED: the reference has no time-domain PML (its only PML is the frequency-domain
stretched-coordinate one of python-src/fdfd.py:14-38), so there is nothing to pin this file
to.  It restates the build's own definition independently of the
product code; tests compare the device path against it bit for bit and check the physics
(reflection well below the reference's Mur frame).

Definition.  Berenger's split-field PML for the TE-mode (Ez, Hx, Hy) system, L cells deep on
all four sides, polynomial grading of order m as in fdfd.py:16-30 (sigma ~ (d/L)^m, default
L = 40, m = 3), outer edge cells of Ez never updated (PEC).  Ez = Ezx + Ezy; the engine stores
Ez (total) and Ezx.  With s = sigma*dt/(2*eps) the dimensionless loss,
    a(s) = (1 - s)/(1 + s),  b(s) = 1/(1 + s),
    s(d) = s_max (d/L)^m,    s_max = (m + 1) ln(1/R0) S / (4 L),   S = Courant number of cell [0,0]
and d the depth into the layer in cells (E rows/columns at integer positions, Hx rows and Hy
columns at half-integer positions), one step is, in this operation order,
    Hx[i,j]  = ahr[i]*Hx[i,j] - (bhr[i]*ch[i,j]) * (Ez[i+1,j] - Ez[i,j])          i <= R-2, j <= C-2
    Hy[i,j]  = ahc[j]*Hy[i,j] + (bhc[j]*ch[i,j]) * (Ez[i,j+1] - Ez[i,j])
  inside the layer (row i or column j at depth > 0), 1 <= i <= R-2, 1 <= j <= C-2:
    ey       = Ez[i,j] - Ezx[i,j]
    Ezx[i,j] = aec[j]*Ezx[i,j] + (bec[j]*ce[i,j]) * (Hy[i,j] - Hy[i,j-1])
    ey       = aer[i]*ey       - (ber[i]*ce[i,j]) * (Hx[i,j] - Hx[i-1,j])
    Ez[i,j]  = Ezx[i,j] + ey
  outside the layer the reference's own update (main.py:21-27), Ezx untouched (it stays 0):
    Ez[i,j] += ((Hy[i,j] - Hy[i,j-1]) - (Hx[i,j] - Hx[i-1,j])) * ce[i,j]
with ch = dt/(mu*dx), ce = dt/(eps*dx) as in the reference (main.py:27,70,74); the point source
is added to Ez (total) afterwards, as fdtd.py:34 does.  Outside the layer a = b = 1, so the H
updates reduce to the reference's exactly (x*1 is exact).
"""
import numpy as np


def depth(n, L, half=False):
    """Depth into the layer (cells) of positions 0..n-1 (or i+1/2 when half)."""
    x = np.arange(n, dtype=np.float64) + (0.5 if half else 0.0)
    return np.maximum(0.0, np.maximum(L - x, x - (n - 1 - L)))


def profiles(rows, cols, courant00, L=40, m=3, R0=1e-6, dtype=np.float64):
    """The eight 1-D factor arrays (float64 math rounded once to dtype) and the layer masks."""
    smax = (m + 1) * np.log(1.0 / R0) * courant00 / (4.0 * L)
    out = {"L": L}
    for name, n in (("r", rows), ("c", cols)):
        se = smax * (depth(n, L) / L) ** m
        sh = smax * (depth(n, L, half=True) / L) ** m
        out["ae" + name] = ((1 - se) / (1 + se)).astype(dtype)
        out["be" + name] = (1 / (1 + se)).astype(dtype)
        out["ah" + name] = ((1 - sh) / (1 + sh)).astype(dtype)
        out["bh" + name] = (1 / (1 + sh)).astype(dtype)
        out["in_" + name] = depth(n, L) > 0
    return out


def step(Ez, Ezx, Hx, Hy, eps, mu, dt, dx, P):
    """One H -> E step in place, arithmetic in the arrays' dtype (scalars weak, as NumPy does)."""
    ch = dt / (mu[:-1, :-1] * dx)
    core = Ez[:-1, :-1]
    Hx[:-1, :] = P["ahr"][:-1, None] * Hx[:-1, :] - (P["bhr"][:-1, None] * ch) * (Ez[1:, :-1] - core)
    Hy[:, :-1] = P["ahc"][None, :-1] * Hy[:, :-1] + (P["bhc"][None, :-1] * ch) * (Ez[:-1, 1:] - core)
    ce = dt / (eps[1:-1, 1:-1] * dx)
    dhy = Hy[1:, 1:-1] - Hy[1:, :-2]
    dhx = Hx[1:-1, 1:] - Hx[:-2, 1:]
    plain = Ez[1:-1, 1:-1] + (dhy - dhx) * ce
    ey = Ez[1:-1, 1:-1] - Ezx[1:-1, 1:-1]
    ex = P["aec"][None, 1:-1] * Ezx[1:-1, 1:-1] + (P["bec"][None, 1:-1] * ce) * dhy
    ey = P["aer"][1:-1, None] * ey - (P["ber"][1:-1, None] * ce) * dhx
    layer = P["in_r"][1:-1, None] | P["in_c"][None, 1:-1]
    Ezx[1:-1, 1:-1] = np.where(layer, ex, Ezx[1:-1, 1:-1])
    Ez[1:-1, 1:-1] = np.where(layer, ex + ey, plain)
    return Ez, Ezx, Hx, Hy


def leapfrog(Ez, Ezx, Hx, Hy, eps, mu, dt, dx, nsteps, src_row, src_col, amps, P):
    for n in range(nsteps):
        step(Ez, Ezx, Hx, Hy, eps, mu, dt, dx, P)
        if amps is not None:
            Ez[src_row, src_col] = Ez.dtype.type(np.float64(Ez[src_row, src_col]) + np.float64(amps[n]))
    return Ez, Ezx, Hx, Hy

"""Stand-in for `torchdiffeq.odeint`, fixed-grid midpoint only (SURVEY.md
Appendix D): grid = t; per interval f0 = f(t0, y); y_mid = y + f0*dt/2;
y <- y + dt * f(t0 + dt/2, y_mid).  atol/rtol unused by fixed-grid solvers.
PARITY UNPINNED by any reference test."""
import torch

def odeint(func, y0, t, *, method='midpoint', atol=None, rtol=None, **kw):
    assert method == 'midpoint', 'shim implements the fixed-grid midpoint solver only'
    ys = [y0]
    y = y0
    for i in range(len(t) - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        f0 = func(t0, y)
        y_mid = y + f0 * (dt * 0.5)
        y = y + dt * func(t0 + dt * 0.5, y_mid)
        ys.append(y)
    return torch.stack(ys)

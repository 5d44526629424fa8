import numpy as np, math
from decimal import Decimal, getcontext
getcontext().prec = 60
from numpy.polynomial import chebyshev as Ch, polynomial as Pl

# P(z) = (atanh(sqrt z)/sqrt z - 1)/z = 1/3 + z/5 + z^2/7 + ...  on z in [0, 1/9]
def P_exact(z):
    z = Decimal(z)
    if z == 0: return Decimal(1)/Decimal(3)
    s = z.sqrt()
    at = ((1+s)/(1-s)).ln()/2
    return (at/s - 1)/z
zmax = 1.0/9.0*1.0001
for deg in (9,10,11,12):
    # Chebyshev nodes
    n = deg+1
    k = np.arange(n)
    x = np.cos(np.pi*(2*k+1)/(2*n))           # [-1,1]
    z = (x+1)*zmax/2
    f = np.array([float(P_exact(float(zz))) for zz in z])
    c = Ch.chebfit(x, f, deg)
    # convert to monomial in z
    px = Ch.cheb2poly(c)                      # poly in x
    # x = 2z/zmax - 1
    pz = np.zeros(1)
    base = np.array([-1.0, 2.0/zmax])
    acc = np.array([1.0])
    res = np.zeros(deg+1)
    for i,ci in enumerate(px):
        res[:len(acc)] += ci*acc
        acc = np.convolve(acc, base)
    # test
    zt = np.linspace(0, 1/9.0, 2001)
    approx = np.zeros_like(zt)
    for ci in res[::-1]: approx = approx*zt + ci
    exact = np.array([float(P_exact(float(zz))) for zz in zt])
    # effect on log: err_log/log = z*dP/(1+zP) roughly
    rel = np.abs(approx-exact)*zt/(1+zt*exact)
    print(deg, 'max rel err contribution to log:', rel.max())
    if deg==11: np.save('scratch/P_coef.npy', res); print([float.hex(float(v)) for v in res]); print(res)

import numpy as np
from decimal import Decimal, getcontext
getcontext().prec = 60
from numpy.polynomial import chebyshev as Ch
def P_exact(z):
    z = Decimal(z)
    if z == 0: return Decimal(1)/Decimal(3)
    s = z.sqrt(); at = ((1+s)/(1-s)).ln()/2
    return (at/s - 1)/z
zmax = 1.0/9.0*1.0001
def fit(deg):
    n=deg+1; k=np.arange(n); x=np.cos(np.pi*(2*k+1)/(2*n)); z=(x+1)*zmax/2
    # solve in Decimal for accuracy: Vandermonde in z (small system) using float128-free approach: use numpy longdouble
    f=np.array([P_exact(float(zz)) for zz in z],dtype=object)
    # Newton divided differences in Decimal -> monomial coefficients
    zz=[Decimal(float(v)) for v in z]
    coef=list(f)
    for j in range(1,n):
        for i in range(n-1,j-1,-1):
            coef[i]=(coef[i]-coef[i-1])/(zz[i]-zz[i-j])
    # convert Newton form to monomial
    mono=[Decimal(0)]*n
    mono[0]=coef[n-1]
    m=1
    for i in range(n-2,-1,-1):
        # mono = mono*(z - zz[i]) + coef[i]
        new=[Decimal(0)]*(m+1)
        for t in range(m):
            new[t+1]+=mono[t]; new[t]-=mono[t]*zz[i]
        new[0]+=coef[i]
        for t in range(m+1): mono[t]=new[t]
        m+=1
    return [float(v) for v in mono]
for deg in (6,7,8,9,10):
    c=fit(deg)
    worst=Decimal(0)
    for zt in np.linspace(0,1/9.0,801):
        zd=Decimal(float(zt)); acc=Decimal(0)
        for ci in c[::-1]: acc=acc*zd+Decimal(ci)
        ex=P_exact(float(zt))
        rel=abs(acc-ex)*zd/(1+zd*ex)
        worst=max(worst,rel)
    print(deg,'approx rel err in log:',float(worst))
    if deg in (8,9): print('  ',', '.join(repr(v) for v in c))

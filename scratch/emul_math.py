import numpy as np, math
LOG2E=1.4426950408889634
LN2_HI=6.93147180369123816490e-01
LN2_LO=1.90821492927058770002e-10
EC=[1.0/math.factorial(n) for n in range(2,14)]   # c2..c13
PC=[0.3333333333333333, 0.20000000000007298, 0.1428571428355253, 0.11111111358900891, 0.09090894708663223, 0.07692785296456121, 0.06657067775440581, 0.06000577591428889, 0.04400158825434387, 0.08082469084735669]
def rcp_approx(x):
    return (1.0/x).astype(np.float32).astype(np.float64)   # pessimistic hardware rcp (24 bits)
def nr(x, t, it=2):
    for _ in range(it):
        e = 1.0 - t*x
        x = x + x*e
    return x
def lean(eta):
    a = np.minimum(np.abs(eta), 750.0)
    x = -a
    kd = np.rint(x*LOG2E)
    r = x - kd*LN2_HI
    r = r - kd*LN2_LO
    q = np.full_like(r, EC[-1])
    for c in EC[-2::-1]: q = q*r + c
    p = (r*r)*q + r
    p = p + 1.0
    e = np.ldexp(p, kd.astype(np.int64))
    t = 1.0 + e
    inv = nr(rcp_approx(t), t)
    u = 2.0 + e
    ru = nr(rcp_approx(u), u)
    s = e*ru
    res = e - s*u
    s = s + res*ru
    z = s*s
    P = np.full_like(z, PC[-1])
    for c in PC[-2::-1]: P = P*z + c
    s2 = s + s
    l1p = (s2*z)*P + s2
    return e, inv, l1p
rng = np.random.default_rng(0)
eta = np.concatenate([rng.normal(size=200000)*3, rng.uniform(-40,40,size=200000), np.array([0.0,1e-300,-1e-300,700,-700,745,750,1000,-1e6, 36.7, 1e-8])])
e, inv, l1p = lean(eta)
a=np.abs(eta)
e0=np.exp(-a); inv0=1.0/(1.0+e0); l0=np.log1p(e0)
def ulps(x,x0):
    with np.errstate(divide='ignore',invalid='ignore'):
        d=np.abs(x-x0)/np.spacing(np.abs(x0))
    d[(x0==0)&(x==0)]=0
    return d
print('exp   max ulp', np.nanmax(ulps(e,e0)[a<700]), ' denormal region max abs err', np.max(np.abs(e-e0)[a>=700]))
print('inv   max ulp', np.nanmax(ulps(inv,inv0)))
m=l0>0
print('log1p max ulp', np.nanmax(ulps(l1p,l0)[m & (a<700)]), 'max abs err', np.max(np.abs(l1p-l0)))

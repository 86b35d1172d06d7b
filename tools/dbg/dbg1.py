import sys, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
pkg=g.load_package(); O=g.load_oracle()
ring=pkg.ring
for nq,np_ in [(2,2),(3,3),(3,18),(4,4),(16,16)]:
    N=1<<10
    Q,P=list(pkg.params.Qi60()[-nq:]),list(pkg.params.Pi60()[-np_:])
    cQ,cP=ring.NewContextWithParams(N,Q),ring.NewContextWithParams(N,P)
    be=ring.NewFastBasisExtender(cQ,cP)
    obe=O.BasisExtender(O.Context(N,Q),O.Context(N,P))
    x=pkg.sampling.uniform_poly(Q,N,2,seed=nq)
    px,pp=cQ.NewPoly(2).set(x),cP.NewPoly(2)
    be.ModUpSplitQP(nq-1,px,pp)
    got=pp.get()
    for b in range(2):
        want=obe.modup_split_qp(nq-1,x[b])
        bad=[(j,int((got[b][j]!=want[j]).sum())) for j in range(np_) if not np.array_equal(got[b][j],want[j])]
        print(nq,np_,b,'bad rows:',bad[:20])
        if bad:
            j=bad[0][0]; idx=np.nonzero(got[b][j]!=want[j])[0][:5]
            print('  idx',idx, got[b][j][idx], want[j][idx])

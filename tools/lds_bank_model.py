import sys
from collections import defaultdict
G32=[list(range(0,32)),list(range(32,64))]
G16c=[list(range(i,i+16)) for i in range(0,64,16)]
G8c=[list(range(i,i+8)) for i in range(0,64,8)]
G128=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128=G128+[[x+32 for x in g] for g in G128]
def cost(addrs, nd, groups, mod):
    # addrs: dict lane -> dword address (or None); nd dwords per lane
    tot=0; ideal=0
    for g in groups:
        banks=defaultdict(set)
        for l in g:
            a=addrs.get(l)
            if a is None: continue
            for d in range(nd): banks[(a+d)%mod].add(a+d)
        c=max([len(v) for v in banks.values()] or [0])
        tot+=max(c,1); ideal+=1
    return tot,ideal
def rd32(ad): return cost(ad,1,G32,32)
def rd64(ad): return cost(ad,2,G32,64)
def rd128(ad): return cost(ad,4,G128,64)
def wr64(ad): return cost(ad,2,G16c,32)
def wr128(ad): return cost(ad,4,G8c,32)
def sim(A,B,hop,K=None):
    KB=A*B; LT=max(A,B); T=64//LT; TRS=A*(B+1)
    res=defaultdict(lambda:[0,0])
    def acc(name,r): res[name][0]+=r[0]; res[name][1]+=r[1]
    lanes=range(64)
    g=lambda x:x//LT; l=lambda x:x%LT
    # build reads (floats)
    for n1 in range(A):
        ad={x:(2*g(x)*hop + l(x) + B*n1) if g(x)<T else (l(x)+B*n1) for x in lanes}
        acc('build fa',rd32(ad)); acc('build fb',rd32({k:v+hop for k,v in ad.items()}))
        acc('window',rd32({x:B*n1+l(x) for x in lanes}))
    for k1 in range(1,A):
        acc('tw',rd64({x:2*(k1*B+l(x)) for x in lanes if l(x)<B}))
    for k1 in range(A):
        acc('transpose wr',wr64({x:2*(g(x)*TRS+k1*(B+1)+l(x)) for x in lanes if g(x)<T and l(x)<B}))
    for n2 in range(B):
        acc('passB rd',rd64({x:2*(g(x)*TRS+l(x)*(B+1)+n2) for x in lanes if g(x)<T and l(x)<A}))
    for k2 in range(B):
        acc('U wr',wr64({x:2*(g(x)*KB+l(x)+A*k2) for x in lanes if g(x)<T and l(x)<A}))
    NP=KB//2; NI=(NP+63)//64
    for gg in range(T):
        for i in range(NI):
            act=[x for x in lanes if x+64*i<NP]
            acc('unt uu',rd128({x:2*(gg*KB+2*(x+64*i)) for x in act}))
            acc('unt p0',rd64({x:2*(gg*KB+((KB-2*(x+64*i)) if x+64*i else 0)) for x in act}))
            acc('unt p1',rd64({x:2*(gg*KB+KB-1-2*(x+64*i)) for x in act}))
    tot=[0,0]
    for k,v in res.items():
        print(f"  {k:14s} cycles {v[0]:5d} ideal {v[1]:5d}"); tot[0]+=v[0]; tot[1]+=v[1]
    print(f"  TOTAL {tot[0]} ideal {tot[1]}  conflict share {(tot[0]-tot[1])/tot[0]:.2f}")
for A,B in ((16,20),(20,20),(24,20),(32,20),(32,30)):
    print(A,B); sim(A,B,A*B//4)

import numpy as np, sys, itertools
R128=[[*range(0,4),*range(12,16),*range(20,28)],[*range(4,12),*range(16,20),*range(28,32)]]
R128=R128+[[l+32 for l in g] for g in R128]
W128=[list(range(8*g,8*g+8)) for g in range(8)]
R64=[list(range(0,32)),list(range(32,64))]
W64=[list(range(16*g,16*g+16)) for g in range(4)]
def cyc(addr_fn, groups, nbanks, dw):
    tot=0
    for g in groups:
        cnt={}
        for l in g:
            a=addr_fn(l)
            if a is None: continue
            for d in range(dw):
                b=((a//4)+d)%nbanks
                cnt.setdefault(b,set()).add(a//4+d)
        tot+=max([len(v) for v in cnt.values()]+[0])
    return tot
def rd128(fn): return cyc(fn,R128,64,4),4
def wr128(fn): return max(cyc(fn,W128,32,4),13),13   # instruction costs 13 anyway
def rd64(fn): return cyc(fn,R64,64,2),2
def wr64(fn): return max(cyc(fn,W64,32,2),6),6
def SW(n): return n ^ ((0xFE10 >> ((n>>1)&12)) & 15)
def ZP(n): return n+(n>>3)
def analyse(LL,TP,sched,S,verbose=False):
    tc=ti=0
    def acc(name,r):
        nonlocal tc,ti
        c,i=r; tc+=c; ti+=i
        if verbose and c!=i: print(f"  {name:30s} {c} vs {i}")
    EPT=LL//TP
    for (R,P) in sched:
        NB=LL//R;MAXB=EPT//R
        for b in range(MAXB):
            for q in range(R):
                acc(f"st({R},{P}) rd b{b} q{q}", rd128(lambda t: 16*S(t%TP+b*TP+q*NB)))
        for b in range(MAXB):
            for q in range(R):
                def f(t):
                    t=t%TP
                    i=t+b*TP;k=i&(P-1);j=(i-k)*R+k
                    return 16*S(j+q*P)
                acc(f"st({R},{P}) wr b{b} q{q}", wr128(f))
    KPT=EPT//2
    for i_ in range(KPT):
        acc(f"post rd Wk {i_}", rd128(lambda t: 16*S(KPT*(t%TP)+i_)))
        acc(f"post rd Wm {i_}", rd128(lambda t: 16*S(0 if KPT*(t%TP)+i_==0 else LL-(KPT*(t%TP)+i_))))
        acc(f"post wr 2k {i_}", wr128(lambda t: 16*S(2*(KPT*(t%TP)+i_))))
        acc(f"post wr 2k+1 {i_}", wr128(lambda t: 16*S(2*(KPT*(t%TP)+i_)+1)))
    core=(tc,ti)
    # x pass fold + readout
    for i_ in range(EPT//2):
        acc(f"xfold a {i_}", wr128(lambda t: 16*S(1+t%TP+i_*TP)))
        acc(f"xfold b {i_}", wr128(lambda t: 16*S(LL-1-t%TP-i_*TP)))
    for i_ in range(EPT):
        acc(f"xread {i_}", rd128(lambda t: 16*S(min(1+t%TP+i_*TP,LL-1))))
    return core,(tc,ti)
for name,S in [("padded",ZP),("swz",SW),("plain",lambda n:n)]:
    for LL,TP,sched in [(768,64,[(4,1),(4,4),(4,16),(12,64)]),(384,32,[(4,1),(4,4),(2,16),(12,32)]),(1024,64,[(8,1),(8,8),(4,64),(4,256)])]:
        core,tot=analyse(LL,TP,sched,S,verbose=(name=="swz" and LL==768))
        print(name,LL,"core",core,"x total",tot)
print("---- 1024 options")
for sched in [[(4,1),(4,4),(4,16),(4,64),(4,256)],[(4,1),(4,4),(4,16),(16,64)],[(16,1),(16,16),(4,256)],[(4,1),(16,4),(16,64)],[(8,1),(8,8),(16,64)]]:
    for name,S in [("padded",ZP),("swz",SW)]:
        try:
            core,tot=analyse(1024,64,sched,S)
            print(sched,name,core,tot)
        except AssertionError as e: print(sched,"n/a")

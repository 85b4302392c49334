"""CPU model of the one-wave tile kernel k_ccl_rows (moving_object_detector_amd/csrc/cluster.hip): the same steps in plain Python —
run labels, down / up sweeps with run levelling, the window pass over the other positions, publish, link requests — followed by
a stand-in for k_ccl_link / k_ccl_merge, against brute-force connected components of the reference's graph
(clusterer_nodelet.cpp:56-83,186-219: up-left window, `!(|dz| > th)` gate, NaN links).  Checks the partition and every
component's first_edge_key.  TEST INFRASTRUCTURE (tests/test_ccl_rows_model.py); it found the kernel's one logic error of round 3
(an up sweep skipped after the window pass) before a second GPU run was spent on it."""
import numpy as np, random, sys
TH,HL,PH,PW=16,4,20,68
HALO=0x8000; NONE=0xFFFF
def gate(a,b,th):
    d=abs(np.float32(a)-np.float32(b))
    return not (d>th)   # NaN links
def tile(mask,z,W,H,x0,y0,n,th):
    # grid
    M=[[False]*PW for _ in range(PH)]; Z=[[0.0]*PW for _ in range(PH)]
    for r in range(PH):
        gy=y0-HL+r
        if gy<0 or gy>=H or r<HL-n: continue
        for j in range(PW):
            gx=x0-HL+j
            if gx<0 or gx>=W: continue
            if j<HL and j<HL-n: continue
            M[r][j]=bool(mask[gy,gx]); Z[r][j]=z[gy,gx]
    if not any(M[r][j] for r in range(HL,PH) for j in range(HL,PW)): return None
    LAB=[[NONE]*PW for _ in range(PH)]
    L1=[[False]*PW for _ in range(PH)]; V=[[False]*PW for _ in range(PH)]; U=[[False]*PW for _ in range(PH)]
    for r in range(PH):
        for j in range(HL,PW):
            if M[r][j] and j>HL and M[r][j-1] and gate(Z[r][j],Z[r][j-1],th): L1[r][j]=True
            if r>0 and M[r][j] and M[r-1][j] and gate(Z[r][j],Z[r-1][j],th): V[r][j]=True
            U[r][j]=L1[r][j] or V[r][j]
        s=None
        for j in range(HL,PW):
            if M[r][j]:
                if not L1[r][j]: s=j
                LAB[r][j]=(r*PW+s)|(HALO if r<HL else 0)
        for j in range(HL):
            if M[r][j]: LAB[r][j]=HALO|(r*PW+j)
    def level(r):
        j=HL
        while j<PW:
            e=j
            while e+1<PW and L1[r][e+1]: e+=1
            m=min(LAB[r][j:e+1])
            for i in range(j,e+1):
                LAB[r][i]=m
            j=e+1
    def sweep(down,force):
        changed=False; carry=None
        rows=range(PH) if down else range(PH-1,-1,-1)
        for i,r in enumerate(rows):
            if not any(M[r][HL:]): continue
            cur=list(LAB[r][HL:]); nv=list(cur)
            if i>0:
                vr=r if down else r+1
                for x in range(64):
                    if V[vr][HL+x] and carry is not None: nv[x]=min(nv[x],carry[x])
            if force or nv!=cur:
                LAB[r][HL:]=nv
                level(r)
                nv=list(LAB[r][HL:])
                if nv!=cur: changed=True
                else: LAB[r][HL:]=cur
            carry=nv
        return changed
    def window():
        changed=False
        for r in range(HL,PH):
            if not any(M[r][HL:]): continue
            cur=list(LAB[r][HL:]); cur0=list(cur)
            for dv in range(0,n+1):
                qr=r-dv
                if not any(M[qr]): continue
                for k in range(1 if dv==0 else 0,n+1):
                    C=[M[r][HL+x] and M[qr][HL+x-k] for x in range(64)]
                    if not any(C): continue
                    lq=[LAB[qr][HL+x-k] for x in range(64)]
                    differ=[C[x] and lq[x]!=cur[x] for x in range(64)]
                    if not any(differ) and not any(C[x] and not U[r][HL+x] for x in range(64)): continue
                    E=[C[x] and gate(Z[r][HL+x],Z[qr][HL+x-k],th) for x in range(64)]
                    for x in range(64):
                        if E[x]: U[r][HL+x]=True
                    D=[E[x] and differ[x] for x in range(64)]
                    if any(D):
                        for x in range(64):
                            if D[x]:
                                m=min(cur[x],lq[x])
                                if lq[x]!=m: LAB[qr][HL+x-k]=m
                                cur[x]=m
                        changed=True
            for x in range(64):
                if cur[x]<cur0[x]: LAB[r][HL+x]=cur[x]
        return changed
    force=False
    while True:
        first=True
        while True:
            d=sweep(True,force); force=False
            if not d and not first: break
            first=False
            if not sweep(False,False): break
        if not window(): break
        force=True
    parent={}; roots={}; reqs=[]
    for r in range(HL,PH):
        for j in range(HL,PW):
            if M[r][j]:
                cur=LAB[r][j]; assert cur<HALO
                rr,cc=divmod(cur,PW); assert rr>=HL and cc>=HL
                parent[(y0+r-HL,x0+j-HL)]=(y0+rr-HL,x0+cc-HL)
    for (p,rt) in parent.items():
        rec=roots.setdefault(rt,[0,None])
        rec[0]+=1
    for r in range(HL,PH):
        for j in range(HL,PW):
            if M[r][j] and U[r][j]:
                rt=parent[(y0+r-HL,x0+j-HL)]
                if roots[rt][1] is None: roots[rt][1]=(y0+r-HL)*W+x0+j-HL
    def lab2pix(lab):
        rr,cc=divmod(lab,PW); return (y0+rr-HL,x0+cc-HL)
    for r in range(HL-n,HL):
        for j in range(HL,PW):
            if M[r][j] and not (L1[r][j] or V[r][j]) and not (LAB[r][j]&HALO):
                reqs.append(((y0-HL+r,x0+j-HL),lab2pix(LAB[r][j])))
    for r in range(PH):
        for j in range(HL):
            if M[r][j] and not (LAB[r][j]&HALO):
                lab=LAB[r][j]
                cu=r>0 and M[r-1][j] and LAB[r-1][j]==lab and gate(Z[r][j],Z[r-1][j],th)
                cl=j>0 and M[r][j-1] and LAB[r][j-1]==lab and gate(Z[r][j],Z[r][j-1],th)
                if not (cu or cl): reqs.append(((y0-HL+r,x0-HL+j),lab2pix(lab)))
    return parent,roots,reqs
def brute(mask,z,W,H,n,th):
    par={}
    def find(a):
        while par[a]!=a:
            par[a]=par[par[a]]; a=par[a]
        return a
    has=set()
    for y in range(H):
        for x in range(W):
            if mask[y,x]: par[(y,x)]=(y,x)
    for y in range(H):
        for x in range(W):
            if not mask[y,x]: continue
            for dv in range(0,n+1):
                for k in range(0,n+1):
                    if dv==0 and k==0: continue
                    qy,qx=y-dv,x-k
                    if qy<0 or qx<0 or not mask[qy,qx]: continue
                    if gate(z[y,x],z[qy,qx],th):
                        has.add((y,x))
                        a,b=find((y,x)),find((qy,qx))
                        if a!=b: par[max(a,b)]=min(a,b)
    comp={p:find(p) for p in par}
    return comp,has
def run(seed,W,H,n,dens,zlev):
    rng=np.random.default_rng(seed)
    mask=rng.random((H,W))<dens
    z=(rng.integers(0,zlev,size=(H,W))*0.2).astype(np.float32)
    if seed%3==0: z[rng.random((H,W))<0.05]=np.nan
    th=np.float32(0.15)
    comp,has=brute(mask,z,W,H,n,th)
    par={}
    def find(a):
        while par[a]!=a: a=par[a]
        return a
    allroots={}; allreq=[]
    for ty in range((H+15)//16):
        for tx in range((W+63)//64):
            t=tile(mask,z,W,H,tx*64,ty*16,n,th)
            if t is None: continue
            p,roots,reqs=t
            for k,v in p.items(): par[k]=v
            for rt in roots: par[rt]=rt
            allroots.update(roots); allreq+=reqs
    for k in list(par):
        if k not in par: pass
    for (h,rt) in allreq:
        assert h in par,(h,"halo pixel not published")
        a,b=find(h),find(rt)
        if a!=b: par[max(a,b)]=min(a,b)
    mine={p:find(p) for p in par}
    assert set(mine)==set(comp)
    # partitions equal?
    m={}
    for p in comp:
        a,b=comp[p],mine[p]
        if m.setdefault(a,b)!=b: return "partition differs at %s"%(p,)
    if len(set(m.values()))!=len(m): return "merged components"
    # keys
    key={}
    for rt,(sz,k) in allroots.items():
        f=find(rt)
        if k is not None: key[f]=min(key.get(f,1<<60),k)
    ref={}
    for p in has:
        c=comp[p]; ref[c]=min(ref.get(c,1<<60),p[0]*W+p[1])
    for c,k in ref.items():
        if key.get(m[c])!=k: return "key differs"
    if len(key)!=len(ref): return "extra keys"
    return None

if __name__ == "__main__":
    bad = cases = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        for (W, H) in ((130, 40), (70, 33)):
            for n in (4, 2, 1):
                for dens, zlev in ((0.1, 2), (0.5, 3), (0.9, 2), (0.3, 8)):
                    cases += 1
                    r = run(seed * 7 + n, W, H, n, dens, zlev)
                    if r:
                        bad += 1
                        print("FAIL", seed, W, H, n, dens, zlev, r)
    print("cases", cases, "bad", bad)
    sys.exit(1 if bad else 0)

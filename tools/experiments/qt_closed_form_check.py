import random
PL=12
def digit(k,d): return (k >> (2*(PL-1-d))) & 3      # k = 28-bit prefix: root<<24 | digits
def iterative(keys,N):
    # keys: sorted list of 28-bit prefixes (root 4 bits + 24)
    n=len(keys)
    roots=[]
    lo=0
    for r in range(16):
        a=lo
        while a<n and (keys[a]>>24)<=r: a+=1
        if a>lo: roots.append((lo,a,0))
        lo=a
    cur=roots; size0=len(cur); state=0; prev=[]; pc=0; D=0
    while state==0:
        va=[]; cuts=[]
        for (lo,hi,dep) in cur:
            if hi-lo==1: va.append(None); cuts.append(None)
            else:
                if dep>=PL: c=(hi,hi,hi)
                else:
                    cs=[]
                    for q in (1,2,3):
                        x=lo
                        while x<hi and digit(keys[x]&0xFFFFFF,dep)<q: x+=1
                        cs.append(x)
                    c=tuple(cs)
                cuts.append(c)
        # children creation order
        children=[]; singles=[]
        for i,(lo,hi,dep) in enumerate(cur):
            if hi-lo==1: singles.append((lo,hi,dep))
            else:
                c=cuts[i]; edge=[lo,c[0],c[1],c[2],hi]
                for q in range(4):
                    a,b=edge[q],edge[q+1]
                    if b>a: children.append((a,b,dep+1))
        sTot=len(children)
        nxt=[None]*(sTot+len(singles))
        prev=[]
        for j,ch in enumerate(children):
            pos=sTot-1-j
            nxt[pos]=ch
            if ch[1]-ch[0]>1: prev.append((ch[1]-ch[0],j,pos))
        for t,sg in enumerate(singles): nxt[sTot+t]=sg
        size=len(nxt); pc=len(prev); D+=1
        if size>=N or size==size0: state=2
        elif size+3*pc>N: state=1
        size0=size; cur=nxt
    return cur,sorted(prev),size0,pc,state,D
def common(a,b):
    x=a^b
    if x>>24: return 0
    if x==0: return 1+PL
    return 1+((x.bit_length() and (24-x.bit_length()))>>1)
def closed(keys,N):
    n=len(keys); ND=PL+1
    c=[0]*(n+1)
    for i in range(1,n): c[i]=common(keys[i-1],keys[i])
    C=[sum(1 for i in range(n) if c[i]<=d) for d in range(ND)]
    E=[sum(1 for i in range(n) if c[i]<=d and c[i+1]>d) for d in range(ND)]
    D=0;st=0;Cprev=C[0];Eprev=E[0]
    for p in range(1,ND):
        Cd,Ed=C[p],E[p]
        if Cd>=N or Cd==Cprev: D=p;st=2
        elif Cd+3*Ed>N: D=p;st=1
        if D: break
        Cprev,Eprev=Cd,Ed
    if not D: return None
    M=C[D]; sTot=M-(Cprev-Eprev)
    los=[i for i in range(n) if c[i]<=D]
    nodes=[]
    for m,lo in enumerate(los):
        hi=los[m+1] if m+1<M else n
        b=D
        if hi-lo==1: b=max(c[lo],c[lo+1])
        kp=keys[lo]
        flip=0xFCCCCCC if b&1 else 0x0333333
        keep=~((1<<(2*(PL-b)))-1)&0xFFFFFFF
        ok=((D-b)<<28)|((kp^flip)&keep)
        nodes.append((ok,(lo,hi,b)))
    order=sorted(range(M),key=lambda m:nodes[m][0])
    A=[None]*M; prev=[]
    for rank,m in enumerate(order):
        nd=nodes[m][1]; A[rank]=nd
        if nd[1]-nd[0]>1: prev.append((nd[1]-nd[0],sTot-1-rank,rank))
    return A,sorted(prev),M,E[D],st,D
random.seed(5)
bad=0
for trial in range(3000):
    nIni=random.choice([1,1,2,4])
    n=random.choice([1,2,3,5,17,100,400,935,1700])
    N=random.choice([5,20,60,105,217,434])
    ks=set()
    cl=random.choice([0,1,2])
    while len(ks)<n:
        r=random.randrange(nIni)
        if cl==0: d=random.getrandbits(24)
        elif cl==1: d=(random.getrandbits(6)<<18)|random.getrandbits(18) if random.random()<0.7 else random.getrandbits(24)
        else: d=(0b101101<<18)|random.getrandbits(18)
        ks.add((r<<24)|d)
    keys=sorted(ks)
    it=iterative(keys,N)
    cf=closed(keys,N)
    if cf is None:
        print('closed gave up',trial); continue
    # compare lists: nodes (lo,hi) and depth for multi-key; prev; size,pc,state
    A1=[(a,b,(d if b-a>1 else None)) for a,b,d in it[0]]
    A2=[(a,b,(d if b-a>1 else None)) for a,b,d in cf[0]]
    if A1!=A2 or it[1]!=cf[1] or it[2:]!=cf[2:]:
        bad+=1
        if bad<4:
            print('MISMATCH',trial,n,N,nIni,it[2:],cf[2:])
            for x,y in list(zip(A1,A2))[:12]: print(x,y)
print('bad',bad)

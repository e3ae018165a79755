import itertools, collections
def mult(slots, nslots):
    c=collections.Counter(s % nslots for s in slots); return max(c.values())
def write_cost(addr_of_lane):  # ds_write_b64: 4 groups x 16 lanes, 16 slots of 8 B; >= 6 cycles (data transfer)
    arr=sum(mult([addr_of_lane(l) for l in range(g*16,g*16+16)],16) for g in range(4))
    return max(6,arr), arr
def read_cost(addr_of_lane):   # ds_read_b64: 2 groups x 32 lanes, 32 slots
    return sum(mult([addr_of_lane(l) for l in range(g*32,g*32+32)],32) for g in range(2))
def exchange(NC,P,R,NS,PM,SH):
    pad=lambda i: i+PM*(i>>SH)
    U=P//R
    w=wa=0
    for u in range(U):
        for t in range(R):
            def a(l,u=u,t=t):
                b=l+64*u
                return pad((b//NS)*(NS*R)+(b%NS)+t*NS)
            c,arr=write_cost(a); w+=c; wa+=arr
    r=0
    for q in range(P):
        r+=read_cost(lambda l,q=q: pad(l+64*q))
    size=pad(NC-1)+1
    return w,wa,r,size
for name,NC,P,stages in (("2048",1024,16,[(16,1),(16,16)]),("1024",512,8,[(8,1),(8,8)]),("512",256,4,[(4,1),(4,4),(4,16)]),("256",128,2,[(2,1),(2,2),(2,4),(2,8),(2,16),(2,32)])):
    print("n_fft",name)
    for (R,NS) in stages:
        cur_pm = 1 if (NS==1 or NS>=16) else (NS if R==2 else 16//R)
        cur=exchange(NC,P,R,NS,cur_pm,4)
        best=None
        for SH in (3,4,5,6):
            for PM in range(0,17):
                e=exchange(NC,P,R,NS,PM,SH)
                tot=e[0]+e[2]
                if e[3] > NC*1.13: continue
                if best is None or tot<best[0]: best=(tot,PM,SH,e)
        print(f"  R {R} NS {NS}: current PM {cur_pm} SH 4 -> write {cur[0]} (array {cur[1]}) read {cur[2]} total {cur[0]+cur[2]} size {cur[3]} | best PM {best[1]} SH {best[2]} write {best[3][0]} (array {best[3][1]}) read {best[3][2]} total {best[0]} size {best[3][3]}")
print("untangle exchange (write rows P/2.., read 64-lane reversed)")
for name,NC,P in (("2048",1024,16),("1024",512,8),("512",256,4),("256",128,2)):
    for lane0_fix in (False,True):
        res=[]
        for SH in (3,4,5,6):
            for PM in range(0,5):
                pad=lambda i: i+PM*(i>>SH)
                w=sum(write_cost(lambda l,q=q: pad(l+64*q))[0] for q in range(P//2,P))
                r=sum(read_cost(lambda l,q=q: pad((64-l if (l or not lane0_fix) else 63)+64*(P-1-q))) for q in range(P//2))
                res.append((w+r,PM,SH,w,r))
        res.sort()
        cur=[x for x in res if x[1]==1 and x[2]==4][0]
        print(f"  n_fft {name} lane0_fix {lane0_fix}: current (PM1,SH4) write {cur[3]} read {cur[4]} | best PM {res[0][1]} SH {res[0][2]} write {res[0][3]} read {res[0][4]}")

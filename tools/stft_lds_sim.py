"""Bank-conflict passes of stft400_kernel's LDS accesses, simulated per phase (no GPU): ds_read2_b64 / ds_write(2)_b64 are served in groups
of 16 lanes over 32 dword banks, ds_read_b64 in groups of 32 lanes over 64.  Prints extra passes per workgroup for 20 and 18 frames with
table rows of 25 and of 40 float2 (SMH_STFT_ROW); the 25-row total (721) matches SQ_LDS_BANK_CONFLICT / 5120 workgroups (703)."""
import numpy as np
M=200; MP=201; nthr=256
def passes(addrs_f2, group, banks_f2):
    # addrs_f2: float2 index per lane (len 64, -1 = inactive); returns extra passes summed over groups
    extra=0
    for g in range(0,64,group):
        a=[x for x in addrs_f2[g:g+group] if x>=0]
        if not a: continue
        bybank={}
        for x in set(a): bybank.setdefault(x%banks_f2,set()).add(x)
        extra+=max(len(v) for v in bybank.values())-1
    return extra
def sim(nf, row=25, tbl_wrapfix=False):
    tot={'p1_tbl':0,'p1_w':0,'p2_r':0,'p2_w':0,'p3_r':0}
    # phase 1
    items=25*nf
    for w0 in range(0,((items+nthr-1)//nthr)*nthr,64):
        lanes=np.arange(w0,w0+64)
        act=lanes<items
        f=lanes//25; n2=lanes%25
        if tbl_wrapfix:
            f0=(lanes & ~15)//25
            j=n2+25*(f-f0)
        else: j=n2
        for n1 in range(8):   # win2 (8) and tw (7) reads, read2 => 16 lanes / 16 float2 banks
            a=[(n1*row+j[i]) if act[i] else -1 for i in range(64)]
            tot['p1_tbl']+=passes(a,16,16)*(2 if n1>0 else 1)
        for k1 in range(8):
            a=[(f[i]*MP+n2[i]+25*k1) if act[i] else -1 for i in range(64)]
            tot['p1_w']+=passes(a,16,16)
    # phase 2
    for w0 in range(0,nthr,64):
        lanes=np.arange(w0,w0+64); k1=lanes//nf; f=lanes%nf; act=k1<8
        for n2 in range(25):
            a=[(f[i]*MP+k1[i]*25+n2) if act[i] else -1 for i in range(64)]
            tot['p2_r']+=passes(a,16,16)
            a=[(f[i]*MP+k1[i]+8*n2) if act[i] else -1 for i in range(64)]
            tot['p2_w']+=passes(a,16,16)
    # phase 3
    items=101*nf
    for w0 in range(0,((items+nthr-1)//nthr)*nthr,64):
        lanes=np.arange(w0,w0+64); k=lanes//nf; f=lanes%nf; act=lanes<items
        a=[(f[i]*MP+k[i]) if act[i] else -1 for i in range(64)]
        tot['p3_r']+=passes(a,32,32)
        a=[(f[i]*MP+(0 if k[i]==0 else M-k[i])) if act[i] else -1 for i in range(64)]
        tot['p3_r']+=passes(a,32,32)
    return tot
for nf in (20,18):
    print(nf, sim(nf), sim(nf,row=40,tbl_wrapfix=True))

"""numpy model of the pair step of potrf_block_pair (tulip.jl_amd/csrc/kernels.hip): publishing the columns j, j + 1 (and the rows j, j + 1 of the
inverse) as they are BEFORE step j and redoing step j locally gives, entry for entry and bit for bit, what two single column steps give
(test infrastructure of round 5; the GPU check is tools/potrf_wave_bench.hip)."""
import numpy as np
rng=np.random.default_rng(0)
def seq(A0, nb):
    A=A0.copy(); W=np.eye(64)
    for j in range(nb):
        colj=A[:,j].copy(); rowj=W[j,:].copy()
        d=colj[j]; inv2=1.0/d; isq=1.0/np.sqrt(d); sq=np.sqrt(d)
        r=np.arange(64)
        arj=np.where((r<nb)&(r>j), colj*inv2, 0.0)
        for col in range(j+1,64):
            A[:,col]=A[:,col]-arj*colj[col]       # cv[col] = A[col][j]
        for c in range(0,j+1):
            W[:,c]=W[:,c]-arj*rowj[c]
        A[:,j]=np.where(r==j, sq, np.where(r>j, colj*isq, colj))
    return A,W
def pair(A0, nb):
    A=A0.copy(); W=np.eye(64)
    r=np.arange(64)
    j=0
    while j<nb:
        if j+1>=nb:   # odd tail: single step
            colj=A[:,j].copy(); rowj=W[j,:].copy()
            d=colj[j]; inv2=1.0/d; isq=1.0/np.sqrt(d); sq=np.sqrt(d)
            arj=np.where((r<nb)&(r>j), colj*inv2, 0.0)
            for col in range(j+1,64): A[:,col]=A[:,col]-arj*colj[col]
            for c in range(0,j+1): W[:,c]=W[:,c]-arj*rowj[c]
            A[:,j]=np.where(r==j, sq, np.where(r>j, colj*isq, colj))
            j+=1; continue
        c0=A[:,j].copy(); c1=A[:,j+1].copy(); r0=W[j,:].copy(); r1=W[j+1,:].copy()   # broadcast (pre-step) columns / rows
        d0=c0[j]; inv20=1.0/d0; isq0=1.0/np.sqrt(d0); sq0=np.sqrt(d0)
        a0=np.where((r<nb)&(r>j), c0*inv20, 0.0)           # arj0 for every row (each thread computes its own and those of its columns)
        c1p=c1-a0*c0[j+1]                                   # A'[.,j+1]
        d1=c1p[j+1]; inv21=1.0/d1; isq1=1.0/np.sqrt(d1); sq1=np.sqrt(d1)
        a1=np.where((r<nb)&(r>j+1), c1p*inv21, 0.0)
        r1p=r1-a0[j+1]*r0                                   # W'[j+1, .]  (only columns <= j matter; r0 is zero beyond j)
        for col in range(j+2,64):
            A[:,col]=(A[:,col]-a0*c0[col])-a1*c1p[col]
        for c in range(0,j+2):
            if c<=j: W[:,c]=(W[:,c]-a0*r0[c])-a1*r1p[c]
            else:    W[:,c]=(W[:,c]-a0*r0[c])-a1*r1p[c]     # c == j+1: r0[j+1] = 0
        A[:,j]=np.where(r==j, sq0, np.where(r>j, c0*isq0, c0))
        A[:,j+1]=np.where(r==j+1, sq1, np.where(r>j+1, c1p*isq1, c1p))
        j+=2
    return A,W
for nb in (64,63,17,2,1):
    M=rng.standard_normal((64,80)); S=M@M.T+64*np.eye(64)
    S[:,nb:]=0; S[nb:,:]=0
    a,w=seq(S,nb); b,v=pair(S,nb)
    L=np.tril(a)[:nb,:nb]; L2=np.tril(b)[:nb,:nb]
    print(nb, "L equal bitwise:", np.array_equal(L,L2), "W equal:", np.array_equal(np.tril(w)[:nb,:nb],np.tril(v)[:nb,:nb]),
          "chol err", np.abs(L@L.T-S[:nb,:nb]).max(), "inv err", np.abs((np.tril(w)[:nb,:nb]/np.diag(L)[:,None])@L-np.eye(nb)).max())

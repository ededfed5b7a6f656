import numpy as np
def hilbert_index(q, bits):
    """Skilling's AxesToTranspose, vectorised. q: (n,3) ints in [0,2^bits). Returns Hilbert index (n,) uint64."""
    X = q.astype(np.uint32).T.copy()  # (3,n)
    n = 3
    M = np.uint32(1) << np.uint32(bits-1)
    Q = M
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(n):
            cond = (X[i] & Q) != 0
            # invert
            X[0] = np.where(cond, X[0] ^ P, X[0])
            # exchange
            t = (X[0] ^ X[i]) & P
            X[0] = np.where(cond, X[0], X[0] ^ t)
            X[i] = np.where(cond, X[i], X[i] ^ t)
        Q = np.uint32(Q >> 1)
    # Gray encode
    for i in range(1, n):
        X[i] ^= X[i-1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[n-1] & Q) != 0, t ^ np.uint32(Q-1), t)
        Q = np.uint32(Q >> 1)
    for i in range(n):
        X[i] ^= t
    # interleave transpose: bit b of X[i] -> position (b*3 + (n-1-i))
    h = np.zeros(X.shape[1], dtype=np.uint64)
    for b in range(bits):
        for i in range(n):
            h |= ((X[i].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(b*3 + (n-1-i))
    return h
def sort_hilbert(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = np.minimum(((p-lo)/(hi-lo+1e-30)*(1<<bits)).astype(np.int64), (1<<bits)-1)
    o = np.argsort(hilbert_index(q, bits), kind='stable')
    return p[o]
def sort_str(p, leaf=64):
    """Sort-Tile-Recursive: x slabs, y strips, z runs."""
    n = len(p); nl = n//leaf
    import math
    s = round(nl ** (1/3))
    sx = s; sy = s; 
    o = np.argsort(p[:,0], kind='stable'); p = p[o]
    out=[]
    per_slab = math.ceil(n/sx/leaf)*leaf
    for a in range(0,n,per_slab):
        sl = p[a:a+per_slab]; sl = sl[np.argsort(sl[:,1],kind='stable')]
        per_strip = math.ceil(len(sl)/sy/leaf)*leaf
        for b in range(0,len(sl),per_strip):
            st = sl[b:b+per_strip]; out.append(st[np.argsort(st[:,2],kind='stable')])
    return np.concatenate(out)
if __name__=='__main__':
    # sanity: consecutive hilbert cells adjacent
    b=3
    g = np.array([(x,y,z) for x in range(8) for y in range(8) for z in range(8)])
    h = hilbert_index(g,b); o=np.argsort(h); d=np.abs(np.diff(g[o],axis=0)).sum(1)
    print('unique',len(np.unique(h)),'max step',d.max())

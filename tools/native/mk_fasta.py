import numpy as np, sys
w=int(sys.argv[1])
r=np.random.default_rng(1)
seq=r.choice(np.frombuffer(b"ACGT",dtype=np.uint8),size=(1<<30)//w*w)
with open("/tmp/big%d.fasta"%w,"wb") as f:
    f.write(b">chr1 test\n")
    a=seq.reshape(-1,w)
    out=np.empty((a.shape[0],w+1),dtype=np.uint8); out[:,:w]=a; out[:,w]=10
    f.write(out.tobytes())

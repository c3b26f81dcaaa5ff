import sys, numpy as np, torch
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/oracle")
import pyoracle as po
from ria_amd.engine import RxEngine
g=np.load("/root/repo/tests/golden/robust_ldpc.npz")
for rate,rn in ((po.R1_3,"R1_3"),(po.R2_3,"R2_3"),(po.R5_6,"R5_6")):
    e=RxEngine("QAM16",rn)
    r=g[f"res_{rate}"]
    out,ok,it,tries=e.ldpc_decode_robust(torch.from_numpy(g[f"llr_{rate}"]).cuda())
    out=out.cpu().numpy(); nb=out.shape[1]
    exp=r[:,3:3+nb].astype(np.uint8)
    bad=np.nonzero((out!=exp).any(axis=1))[0]
    print(rn,"info_bits",e.geo.info_bits,"nb",nb,"rows",len(r),"bad rows",bad, "ok of bad", r[bad,0], "tries", r[bad,1])
    for b in bad[:3]:
        d=np.nonzero(out[b]!=exp[b])[0]
        print("  row",b,"diff byte idx",d[:10], out[b][d[:5]], exp[b][d[:5]])

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from edge264_amd import backend, packet as P, synth
from oracle.pyoracle import Oracle
w,h=6,5
dev=backend.Device(0); orc=Oracle()
s=synth.StreamSynth(w,h,0,weighted=1)
nb=P.frame_bytes(w,h)
rng=np.random.default_rng(1000)
dpb=[rng.integers(0,256,nb+16,dtype=np.uint8) for _ in range(6)]+[None]*26
st=backend.Stream(dev,w,h)
for i in range(6):
    st.alloc(i); st.upload(i,dpb[i][:nb])
for i,t in enumerate("IP"):
    pkt=s.next_frame(t); pk=P.Packet(pkt); d=int(pk.hdr["dst_slot"])
    dp=dev.upload_packet(pkt)
    orc.decode_frame(pkt,dpb,1); dev.submit_batch([st],[dp],1)
    got=st.download(d)
    if t=="P":
        g=got[:w*16*h*16].reshape(h*16,w*16); e=dpb[d][:w*16*h*16].reshape(h*16,w*16)
        bad=np.argwhere(g!=e)
        print("n bad", len(bad))
        mbs=sorted(set((int(y)//16,int(x)//16) for y,x in bad))
        print("bad mbs", mbs[:20])
        my,mx=mbs[0]
        print("hip\n", g[my*16:my*16+8, mx*16:mx*16+16]); print("oracle\n", e[my*16:my*16+8, mx*16:mx*16+16])
        a=my*w+mx
        print("mb", pk.mbs[a]); print("motion refPic", pk.motion[a]["refPic"], "refIdx", pk.motion[a]["refIdx"], "mvs", pk.motion[a]["mvs"][:32].reshape(16,2).tolist())
        sl=pk.slices[int(pk.mbs[a]["slice"])]
        print("idc", sl["weighted_bipred_idc"], "lwd", sl["luma_log2_weight_denom"], "cwd", sl["chroma_log2_weight_denom"])
        print("ew", sl["explicit_weights"][:, :4].tolist(), "eo", sl["explicit_offsets"][:, :4].tolist())
        # per-MB summary: which MBs bad vs their refIdx
        for (yy,xx) in mbs[:12]:
            aa=yy*w+xx
            print((yy,xx), "kind", pk.mbs[aa]["kind"], "refIdx", pk.motion[aa]["refIdx"][:4].tolist(), "coded", hex(int(pk.mbs[aa]["coded"])))
        good=[(yy,xx) for yy in range(h) for xx in range(w) if (yy,xx) not in mbs and pk.mbs[yy*w+xx]["kind"]==5]
        for (yy,xx) in good[:8]:
            aa=yy*w+xx
            print("good", (yy,xx), "refIdx", pk.motion[aa]["refIdx"][:4].tolist())
        # unweighted prediction of the same packet (idc forced to 0) for a macroblock without residual
        import copy
        pk2 = bytearray(pkt)
        so = int(pk.hdr["slices_off"])
        pk2[so + 1] = 0
        dpb2=[None if b is None else b.copy() for b in dpb]
        # rebuild the references as they were before this frame: only dst slot changed
        orc.decode_frame(bytes(pk2), dpb2, 1)
        p=dpb2[d][:w*16*h*16].reshape(h*16,w*16)
        for (yy,xx) in [(0,3),(0,4)]:
            print("MB",(yy,xx),"mv",pk.motion[yy*w+xx]["mvs"][:2].tolist())
            for r in range(16):
                print(r, "p", p[yy*16+r, xx*16:xx*16+16].tolist(), "| exp", e[yy*16+r, xx*16:xx*16+16].tolist(), "| hip", g[yy*16+r, xx*16:xx*16+16].tolist())

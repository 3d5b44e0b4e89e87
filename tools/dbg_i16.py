import sys; sys.path.insert(0,'.')
import numpy as np
from cutter_vad_amd import weights_io
from cutter_vad_amd.engine import Engine
from tests.signals import make_streams
blob=open(weights_io.packaged_blob_path(5),'rb').read()
e=Engine(blob,max_streams=64)
f32=make_streams(3,2,seed=11)
i16=np.clip(np.round(f32*32767.0),-32768,32767).astype(np.int16)
as32=(i16.astype(np.float32)/np.float32(32767.0)).astype(np.float32)
a=e.open_streams(3); b=e.open_streams(3)
for t in range(2):
    pa=e.step(a,as32[:,t]); pb=e.step(b,i16[:,t])
    print(t,pa,pb)
# isolate: a frame that is zero except one sample
for pos in (0,1,5,64,65,128,129,192,200,255,256,300,511):
    z=np.zeros((1,512),np.int16); z[0,pos]=20000
    e.reset(a[:1]); e.reset(b[:1])
    pa=e.step(a[:1],(z.astype(np.float32)/np.float32(32767.0))); pb=e.step(b[:1],z)
    print(pos,pa,pb, 'DIFF' if abs(pa[0]-pb[0])>1e-5 else '')

import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pfb_imaging_amd import _lib
from pfb_imaging_amd.operators.gridder import PartitionResidual
from pfb_imaging_amd.utils import synth
npix=8192; nrow=500000
c=synth.make_case(nrow,8,npix,zscale=1e-3,seed=0,with_vis=False)
part={"UVW":c["uvw"],"FREQ":c["freq"],"WEIGHT":c["wgt"][None],"MASK":c["mask"],"BEAM":np.ones((1,npix,npix)),"attrs":{"l0":0.0,"m0":0.0}}
pr=PartitionResidual([part],npix,npix,c["cell"])
dirty=np.zeros((1,npix,npix)); model=np.random.default_rng(0).standard_normal((1,npix,npix))
pr.residual(dirty,model)
T=time.perf_counter
for rep in range(2):
    t0=T(); pr._m_dev.upload(model[0]); t1=T(); pr._a_dev.upload(dirty[0]); t2=T()
    g,wgt,_=pr.items[0]; g.set_weights(wgt[0]); t3=T()
    g.residual_dev(pr._m_dev,pr._a_dev,pr._a_dev,beam_dev=pr._beams_dev[0][0]); t4=T()
    out=_lib.result_empty((1,npix,npix),np.float64); t5=T(); pr._a_dev.download(out[0]); t6=T()
    cube=np.zeros((4,1,npix,npix)); t7=T(); cube[0]=out; t8=T()
    print("up model %.1f up dirty %.1f set_w %.1f apply %.1f alloc %.1f down %.1f zeros %.1f assign %.1f"%tuple(1e3*x for x in (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5,t7-t6,t8-t7)))
t0=T(); r=pr.residual(dirty,model); print("residual() total %.1f"%(1e3*(T()-t0)))

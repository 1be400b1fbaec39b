import sys, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from circl_amd import _native as nat
from oracle import orc
L=nat.lib()
param=65; PK,SK,SIG=orc.DSA_SIZES[param]
for n in (1<<14, 1<<16, 1<<17):
    rng=np.random.default_rng(1); pool=1<<10
    pk,sk=orc.mldsa_keygen(param, rng.integers(0,256,(pool,32),dtype=np.uint8))
    d_sk=torch.from_numpy(np.tile(sk,(n//pool,1))).cuda()
    d_msg=torch.from_numpy(rng.integers(0,256,32*n+16,dtype=np.uint8)).cuda()
    d_off=torch.arange(0,32*(n+1),32,dtype=torch.int64).cuda()
    d_rnd=torch.zeros((n,32),dtype=torch.uint8,device='cuda')
    sig=torch.empty((n,SIG),dtype=torch.uint8,device='cuda')
    wsb=L.circl_hip_mldsa_sign_workspace_size(param,n); ws=torch.empty(wsb,dtype=torch.uint8,device='cuda')
    st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        rc=L.circl_hip_mldsa_sign_dev(param,d_sk.data_ptr(),d_msg.data_ptr(),d_off.data_ptr(),None,None,d_rnd.data_ptr(),0,sig.data_ptr(),n,ws.data_ptr(),wsb,st); assert rc==0
    run(); torch.cuda.synchronize()
    import time
    t=time.perf_counter(); run(); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(f"n={n}: {dt*1e3:.2f} ms -> {n/dt:.3e}/s  ws {wsb/1e9:.2f} GB")

import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from vilmedic_amd.generation import _attn
from test_hip_kernels_gpu import _attn_ref, rnd
dev=torch.device('cuda:0')
M,H,T,D=5,2,24,128
for Lk in (15,16,17,18,24):
    q=rnd(M,D,seed=1).to(dev); cache=rnd(M*T,2*D,seed=2).to(dev)
    index=(torch.arange(M,device=dev,dtype=torch.int32)[:,None]*T+torch.arange(T,device=dev,dtype=torch.int32)[None,:]).contiguous()
    o=_attn(q,D,cache,2*D,cache[:,D:],2*D,M,H,1,Lk,64,kv_index=index,kv_index_ld=T)
    kk=cache.view(M,T,2*D)[:,:Lk,:D].float(); vv=cache.view(M,T,2*D)[:,:Lk,D:].float()
    ref=_attn_ref(q.float().view(M,1,D),kk,vv,H,None,False).view(M,D)
    print(Lk,(o.float()-ref).abs().max().item())
# permuted index
perm=torch.randperm(M*T,device=dev).to(torch.int32)[:M*T].view(M,T).contiguous()
Lk=20
o=_attn(q,D,cache,2*D,cache[:,D:],2*D,M,H,1,Lk,64,kv_index=perm,kv_index_ld=T)
g=cache[perm.long()[:,:Lk]]
ref=_attn_ref(q.float().view(M,1,D),g[...,:D].float(),g[...,D:].float(),H,None,False).view(M,D)
print('perm',(o.float()-ref).abs().max().item())

"""fp32 720p warp (upsample_grid_sample_fwd_kernel): the paired 8-byte tap loads at 4-byte alignment (product) against aligned 12-byte loads + selects
(PWS_OPT_EXPERIMENT 6, timing probe), three fields, 8 frames; microseconds per launch."""
import contextlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
from pwstablenet_amd import functional as PF, hipabi as A, synth
from pwstablenet_amd.lib.networks_cascading import define_G
B=8; dev=torch.device("cuda"); torch.manual_seed(0)
rot=[torch.rand((B,3,720,1280),device=dev)*255 for _ in range(4)]
th=torch.tensor([1,0,0,0,1,0],device=dev,dtype=torch.float32).repeat(B,1)
ramp=torch.linspace(0,6.28,256,device=dev)
fields={"smooth":PF.affine_grid(th+0.02*torch.randn_like(th),(B,3,256,256))+(4.0/256)*(torch.sin(3*ramp).view(1,256,1,1)*torch.cos(2*ramp).view(1,1,256,1)),
        "translation":PF.affine_grid(th+torch.tensor([0,0,0.01,0,0,-0.02],device=dev),(B,3,256,256))}
with contextlib.redirect_stdout(sys.stderr):
    net=define_G(31,2,64,"normal",0.02)
net.load_state_dict({"module."+k:torch.from_numpy(v) for k,v in synth.make_weights("W1",seed=123,ngf=64)}); net=net.cuda()
with torch.no_grad():
    fields["generator"]=net(torch.from_numpy(synth.make_window(B,31,256,seed=11)).cuda(),False).clone()
L=A.lib()
for name,field in fields.items():
    out={}
    for exp in (0,6,0,6):
        L.pws_set_option(A.OPT_EXPERIMENT, exp)
        with torch.no_grad():
            for i in range(8): o=PF.upsample_grid_sample(rot[i%4],field)
            torch.cuda.synchronize(); L.pws_prof_enable(1)
            for i in range(24): PF.upsample_grid_sample(rot[i%4],field)
            L.pws_prof_enable(0)
        r=sorted(x[4] for x in A.prof_collect() if x[0]=="upsample_grid_sample_fwd_kernel")
        out.setdefault(exp,[]).append(1e3*r[len(r)//2]); last=o
        if exp==0: ref=o.clone()
        else: same=bool(torch.equal(ref,o))
    by=B*720*1280*24.0+8.0*B*256*256
    print("%-12s product %s us   aligned %s us   (%.3f / %.3f of 8 TB/s)  equal %s"%(name,out[0],out[6],by/min(out[0])/1e3/8000,by/min(out[6])/1e3/8000,same))
L.pws_set_option(A.OPT_EXPERIMENT, 0)

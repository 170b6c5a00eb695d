"""A/B switches of the uint8 720p warp on the smooth field, 8 frames (PWS_OPT_EXPERIMENT 49: per-lane field windows instead of the wave's
shared row pair; 4: float -> byte by (int) + clamp + shift instead of the SDWA convert), microseconds per launch (median, min)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pwstablenet_amd import functional as PF, hipabi as A
B=8; dev=torch.device("cuda"); torch.manual_seed(0)
rot=[torch.randint(0,256,(B,720,1280,3),device=dev,dtype=torch.uint8) for _ in range(4)]
th=torch.tensor([1,0,0,0,1,0],device=dev,dtype=torch.float32).repeat(B,1)
ramp=torch.linspace(0,6.28,256,device=dev)
field=PF.affine_grid(th+0.02*torch.randn_like(th),(B,3,256,256))+(4.0/256)*(torch.sin(3*ramp).view(1,256,1,1)*torch.cos(2*ramp).view(1,1,256,1))
L=A.lib()
for exp in (0,49,4,0):
    L.pws_set_option(A.OPT_EXPERIMENT, exp)
    with torch.no_grad():
        for i in range(8): PF.upsample_grid_sample_u8(rot[i%4],field,swap_rb=True)
        torch.cuda.synchronize(); L.pws_prof_enable(1)
        for i in range(24): PF.upsample_grid_sample_u8(rot[i%4],field,swap_rb=True)
        L.pws_prof_enable(0)
    r=sorted(x[4] for x in A.prof_collect() if x[0]=="upsample_grid_sample_u8_kernel")
    print("experiment %d: %.1f us  min %.1f"%(exp, 1e3*r[len(r)//2], 1e3*r[0]))
L.pws_set_option(A.OPT_EXPERIMENT, 0)

import os, sys, time
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/any-stereo_amd"]
import torch
from anystereo import ops
from anystereo.harness.synthetic import det_uniform, fill_module_deterministic
from anystereo.models.base import default_args
from anystereo.nn.geometry import Combined_Geo_Encoding_Volume
from anystereo.nn.update import BasicMultiUpdateBlock
dev="cuda:0"
h,w=136,240
args=default_args("continuous_IGEVStereo")
ub=BasicMultiUpdateBlock(args, hidden_dims=args.hidden_dims, geo_channels=8).eval(); fill_module_deterministic(ub, base_seed=5); ub=ub.to(dev)
f1,f2=det_uniform((1,96,h,w),1).to(dev),det_uniform((1,96,h,w),2).to(dev)
fn=Combined_Geo_Encoding_Volume(f1,f2,det_uniform((1,8,48,h,w),3).to(dev),num_levels=2,radius=4)
net0=torch.tanh(det_uniform((1,128,h,w),4,-2,2)).to(dev)
disp=det_uniform((1,1,h,w),5,0.0,60.0).to(dev)
with torch.no_grad():
    taps=ub.disp_head.taps(net0)
    pack=ub.encoder.__dict__.setdefault("_plc1", ops.LookupConvPack()).get(ub.encoder.convc1.weight, ub.encoder.convc1.bias)
    out=ub.encoder.new_output(disp)
    def front(): return fn.loop_front(taps, ub.disp_head.conv2.bias, disp, pack, ub.encoder.convd1.weight, ub.encoder.convd1.bias, out, 127)
    def staged():
        d=ub.disp_head.finish(taps, addend=disp)
        cor=ops.BS8.empty(1,64,h,w,dev); fn.lookup_convc1(d, pack, out_bs=cor)
        d1=ops.BS8.empty(1,64,h,w,dev); ops.conv7x7_c1_relu(d, ub.encoder.convd1.weight, ub.encoder.convd1.bias, out=d1, copy_out=out, copy_coff=127)
    for name,fnc in (("front",front),("staged",staged)):
        for _ in range(3): fnc()
        torch.cuda.synchronize()
        g=torch.cuda.CUDAGraph()
        s_=torch.cuda.Stream()
        with torch.cuda.stream(s_):
            with torch.cuda.graph(g):
                for _ in range(20): fnc()
        g.replay(); torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        print(name, round(a.elapsed_time(b)/20*1e3,1), "us")

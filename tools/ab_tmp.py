import sys, time, torch
sys.path.insert(0, '/root/repo')
from wetts_amd import SynthesizerTrn, config, synth
net = SynthesizerTrn(256, 513, 32, n_speakers=1, **config.MODEL_CONFIGS["v1"])
net.load_state_dict(synth.make_state_dict(net.cfg, 0)).to("cuda")
g = torch.Generator().manual_seed(0)
x = torch.randint(0, 256, (16, 128), generator=g).cuda(); xl = torch.full((16,), 128).cuda(); sid = torch.zeros(16, dtype=torch.long).cuda()
kw = dict(noise_scale=0.667, length_scale=0.92, noise_scale_w=0.8)
def one(ov, seed=5):
    torch.manual_seed(seed); net.set_overlap(ov)
    o, a, ym, (z, zp, mp, lp) = net.infer(x, xl, sid=sid, **kw)
    torch.cuda.synchronize()
    return dict(o=o.clone(), z=z.clone(), zp=zp.clone(), logw=net._last["logw"].clone(), Ty=ym.shape[-1])
r = [one(False), one(False), one(True), one(True), one(False)]
for i in range(1, 5):
    a, b = r[0], r[i]
    print(i, "Ty", a["Ty"], b["Ty"], {k: (float((a[k] - b[k]).abs().max()) if a[k].shape == b[k].shape else "shape") for k in ("logw", "zp", "z", "o")})

"""development: the stage-2 pose encoder's gradients on the GPU against float64, on the model's own encoder and input."""
import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianavatar_amd.avatar_model import AvatarModel, collate_frames, default_params
torch.manual_seed(0)
mp, npar, op = default_params(batch_size=2, num_points=3000, image_width=128, image_height=128, num_frames=16,
                              train_stage=2, query_posmap_size=64)
m = AvatarModel(mp, npar, op, train=True)
batch = collate_frames([m.train_dataset[i] for i in (0, 1)], "cuda")
x = batch["inp_pos_map"]
enc = m.pose_encoder.train()
for trial in range(2):
    w = torch.randn(2, 64, 128, 128, device="cuda")
    e64 = copy.deepcopy(enc).cpu().double()
    acts64, actsg = {}, {}
    def hook(store):
        def f(mod, inp, out):
            if torch.is_tensor(out):
                out.retain_grad(); store[id(mod)] = out
        return f
    hs = [mod.register_forward_hook(hook(actsg)) for mod in enc.modules()]
    names = {id(mod): n for n, mod in enc.named_modules()}
    for p in enc.parameters(): p.grad = None
    (enc(x) * w).sum().backward()
    for h in hs: h.remove()
    names64 = {}
    hs = []
    for (n, mod) in e64.named_modules():
        names64[id(mod)] = n
        hs.append(mod.register_forward_hook(hook(acts64)))
    (e64(x.cpu().double()) * w.cpu().double()).sum().backward()
    for h in hs: h.remove()
    a64 = {names64[k]: v for k, v in acts64.items()}
    ag = {names[k]: v for k, v in actsg.items()}
    print("trial", trial)
    for n in a64:
        if n in ag and a64[n].grad is not None and ag[n].grad is not None:
            f = float((ag[n].detach().cpu().double() - a64[n].detach()).abs().max() / a64[n].detach().abs().max())
            b = float((ag[n].grad.cpu().double() - a64[n].grad).abs().max() / a64[n].grad.abs().max())
            print(f"  act {n:16s} fwd rel {f:.2e}  grad rel {b:.2e}  shape {tuple(a64[n].shape)} min|x| {float(a64[n].detach().abs().min()):.1e}")
    for (n, p64), (_, pg) in zip(e64.named_parameters(), enc.named_parameters()):
        t = float(p64.grad.abs().max())
        print(f"  {n:24s} gpu rel {float((pg.grad.cpu().double()-p64.grad).abs().max())/t:.2e}")
    if trial == 0:      # second trial: after one "training-mode forward under no_grad" (as the assembled test does)
        with torch.no_grad():
            enc(x)

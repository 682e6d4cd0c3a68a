"""GPU diagnostic: first-order check of the cfg-4 training gradient (loss(theta - eps g/|g|) vs loss(theta) - eps |g|)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "any-stereo_amd")]
import torch
from anystereo.harness.metrics import sequence_loss_multiscale
from anystereo.harness.synthetic import fill_module_deterministic
from anystereo.harness.train import synthetic_train_batch
from anystereo.models import __models__, default_args
DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 16
args = default_args("continuous_IGEVStereo")
model = __models__["continuous_IGEVStereo"](args)
fill_module_deterministic(model, base_seed=1)
model = model.to(DEV).train()
model.freeze_bn()
img1, img2, coord, gt, scale = synthetic_train_batch(4, 160, 320, seed=3, device=DEV)

def loss_of(grad):
    with torch.set_grad_enabled(grad):
        _, preds = model(img1, img2, iters=iters, hr_coord=coord.clone(), scale=scale)
        return sequence_loss_multiscale(preds, gt, ((gt < 512) & (gt > 0)).float(), max_disp=args.max_disp)[0]

l0 = loss_of(True)
l0.backward()
print("loss grad-mode", l0.item(), " no_grad", loss_of(False).item(), " grad-mode again", loss_of(True).item())
params = [p for p in model.parameters() if p.grad is not None]
gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)).item()
print("gnorm", gnorm)
base = [p.detach().clone() for p in params]
for eps in (1e-5, 1e-4, 1e-3, 2e-3):
    with torch.no_grad():
        for p, b in zip(params, base):
            p.copy_(b - p.grad * (eps / gnorm))
    la, lb = loss_of(True).item(), loss_of(False).item()
    print(f"eps {eps:g}: predicted decrease {eps * gnorm:.5f}  actual (grad-mode fwd) {l0.item() - la:.5f}  (no_grad fwd) {l0.item() - lb:.5f}")
